// Sanitizer harness (see run.sh): mutated ELEMENTARY streams through the Video / Audio parsers over the lane-emulator
// backends, with half of the mutations aimed at the headers (sequence headers, picture headers, slice starts / audio frame
// headers): sizes, matrices, picture types, f_codes, bitrate / sample-rate / mode fields.
// usage: fuzz_streams video|audio file first_seed n_seeds
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
    if (argc < 5)
        return 2;
    const bool video = !strcmp(argv[1], "video");
    FILE *f = fopen(argv[2], "rb");
    if (!f)
        return 2;
    std::vector<uint8_t> clean(8 << 20);
    clean.resize(fread(clean.data(), 1, clean.size(), f));
    fclose(f);
    static float window[512];
    for (int i = 0; i < 512; i++)
        window[i] = (float)mpg_synth_window_x2[i] * 0.5f;
    std::vector<size_t> headers; // where headers start
    for (size_t i = 0; i + 4 < clean.size(); i++) {
        if (video ? (clean[i] == 0 && clean[i + 1] == 0 && clean[i + 2] == 1) : (clean[i] == 0xFF && (clean[i + 1] & 0xFE) == 0xFC))
            headers.push_back(i);
    }
    const int first = atoi(argv[3]), count = atoi(argv[4]);
    for (int seed = first; seed < first + count; seed++) {
        g_state = 0x9E3779B97F4A7C15ull * (uint64_t)(seed + 1);
        std::vector<uint8_t> d = clean;
        const int n_mut = 2 + (int)below(40);
        for (int k = 0; k < n_mut; k++) {
            size_t p = (size_t)below(d.size() - 16);
            if (below(2) && !headers.empty())
                p = headers[below(headers.size() < 40 ? headers.size() : 40 + below(headers.size() - 40 + 1) % headers.size())] + 3 + (size_t)below(9);
            if (below(6) == 0)
                p = 4 + (size_t)below(8); // the first header's fields: picture size, rates (video) / the first frame header (audio)
            if (p + 8 >= d.size())
                continue;
            switch (below(3)) {
            case 0: d[p] ^= (uint8_t)(1u << below(8)); break;
            case 1: d[p] = (uint8_t)below(256); break;
            default: for (int i = 0; i < 4; i++) d[p + i] = (uint8_t)below(256); break;
            }
        }
        if (below(5) == 0)
            d.resize(200 + (size_t)below(d.size() - 200));
        printf("seed %d: %d mutations, %zu bytes\n", seed, n_mut, d.size());
        fflush(stdout);
        try {
            std::unique_ptr<Buffer> buf = Buffer::FromMemory(d.data(), d.size());
            int n = 0;
            if (video) {
                Video v(buf.get(), std::unique_ptr<VideoBackend>(static_cast<VideoBackend *>(host_emu_video_backend((int)below(2)))));
                while (n < 300 && v.Decode())
                    n++;
                v.Rewind();
                for (int t = 0; t < 5 && v.Decode(); t++)
                    n++;
                printf("  %d frames, %d x %d\n", n, v.Width(), v.Height());
            } else {
                Audio a(buf.get(), std::unique_ptr<AudioBackend>(static_cast<AudioBackend *>(host_emu_audio_backend((int)below(2), window))));
                while (n < 400 && a.Decode())
                    n++;
                a.Rewind();
                for (int t = 0; t < 5 && a.Decode(); t++)
                    n++;
                printf("  %d sample blocks\n", n);
            }
        } catch (const std::exception &e) {
            printf("  refused: %s\n", e.what());
        }
    }
    return 0;
}
