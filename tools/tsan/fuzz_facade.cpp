// Sanitizer harness (see run.sh): mutated program streams through the whole host stack — Demux, MPEG facade (DecodeVideo,
// DecodeAudio, Seek, SeekFrame, Duration, Rewind, Decode), Video and Audio parsers — over the lane-emulator backends.
// usage: fuzz_facade file.mpg first_seed n_seeds        (prints the seed before each run: a crash names its input)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "mpeg.hpp"
#include "iso11172_synth_window.h"
extern "C" void *host_emu_video_backend(int flavour);
extern "C" void *host_emu_audio_backend(int fma, const float *window512);
using namespace mpeg;

static uint64_t g_state;
static uint64_t rnd()
{
    g_state ^= g_state << 13;
    g_state ^= g_state >> 7;
    g_state ^= g_state << 17;
    return g_state;
}
static uint64_t below(uint64_t n) { return rnd() % n; }

int main(int argc, char **argv)
{
    if (argc < 4)
        return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f)
        return 2;
    std::vector<uint8_t> clean(8 << 20);
    clean.resize(fread(clean.data(), 1, clean.size(), f));
    fclose(f);
    static float window[512];
    for (int i = 0; i < 512; i++)
        window[i] = (float)mpg_synth_window_x2[i] * 0.5f;
    const int first = atoi(argv[2]), count = atoi(argv[3]);
    for (int seed = first; seed < first + count; seed++) {
        g_state = 0x9E3779B97F4A7C15ull * (uint64_t)(seed + 1);
        std::vector<uint8_t> d = clean;
        const int n_mut = 3 + (int)below(58);
        for (int k = 0; k < n_mut; k++) {
            const size_t p = (size_t)below(d.size() - 8);
            switch (below(4)) {
            case 0: d[p] ^= (uint8_t)(1u << below(8)); break;
            case 1: d[p] = (uint8_t)below(256); break;
            case 2: for (int i = 0; i < 4; i++) d[p + i] = (uint8_t)below(256); break;
            default: {
                static const uint8_t codes[] = {0xBA, 0xBB, 0xE0, 0xC0, 0xB3, 0x00, 0xB9, 0xB8, 0x01};
                d[p] = 0, d[p + 1] = 0, d[p + 2] = 1, d[p + 3] = codes[below(sizeof(codes))];
            }
            }
        }
        if (below(4) == 0)
            d.resize(1000 + (size_t)below(d.size() - 1000));
        printf("seed %d: %d mutations, %zu bytes\n", seed, n_mut, d.size());
        fflush(stdout);
        MPEG::Backends be;
        be.video = [] { return std::unique_ptr<VideoBackend>(static_cast<VideoBackend *>(host_emu_video_backend(0))); };
        be.audio = [](int fma) { return std::unique_ptr<AudioBackend>(static_cast<AudioBackend *>(host_emu_audio_backend(fma, window))); };
        try {
            MPEG m(d.data(), d.size(), be);
            int nv = 0, na = 0;
            for (int t = 0; t < 400 && m.DecodeVideo(); t++)
                nv++;
            for (int t = 0; t < 400 && m.DecodeAudio(); t++)
                na++;
            m.Seek((double)below(9000) / 1000.0, below(2) != 0);
            for (int t = 0; t < 10 && m.DecodeVideo(); t++)
                nv++;
            Frame *fr = m.SeekFrame((double)below(9000) / 1000.0, true);
            const double dur = m.Duration();
            m.Rewind();
            for (int t = 0; t < 30 && !m.HasEnded(); t++)
                m.Decode(1.0 / 30);
            printf("  %d frames, %d sample blocks, seek frame %s, duration %.3f\n", nv, na, fr ? "yes" : "no", dur);
        } catch (const std::exception &e) {
            printf("  refused: %s\n", e.what());
        }
    }
    return 0;
}
