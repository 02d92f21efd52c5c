// Sanitizer harness (see run.sh): the whole host stack on a program stream — Demux, MPEG facade (Decode with
// callbacks, Seek, SeekFrame), Video and Audio parsers — over the lane-emulator backends.
#include <stdio.h>
#include <vector>
#include "mpeg.hpp"
#include "iso11172_synth_window.h"
extern "C" void *host_emu_video_backend(int flavour);
extern "C" void *host_emu_audio_backend(int fma, const float *window512);
using namespace mpeg;
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    std::vector<uint8_t> d(8 << 20);
    d.resize(fread(d.data(), 1, d.size(), f));
    static float window[512];
    for (int i = 0; i < 512; i++)
        window[i] = (float)mpg_synth_window_x2[i] * 0.5f;
    MPEG::Backends be;
    be.video = [] { return std::unique_ptr<VideoBackend>(static_cast<VideoBackend *>(host_emu_video_backend(0))); };
    be.audio = [](int fma) { return std::unique_ptr<AudioBackend>(static_cast<AudioBackend *>(host_emu_audio_backend(fma, window))); };
    MPEG m(d.data(), d.size(), be);
    int nv = 0, na = 0;
    m.SetVideoCallback([&](MPEG *, Frame *) { nv++; });
    m.SetAudioCallback([&](MPEG *, Samples *) { na++; });
    for (int t = 0; t < 400 && !m.HasEnded(); t++)
        m.Decode(1.0 / 30);
    printf("callbacks: %d frames, %d sample blocks, duration %.3f\n", nv, na, m.Duration());
    m.Seek(4.0, false);
    m.Seek(2.5, true);
    Frame *fr = m.SeekFrame(7.0, true);
    printf("seek frame at %.3f\n", fr ? fr->Time : -1.0);
    m.Rewind();
    int n = 0;
    while (m.DecodeVideo())
        n++;
    printf("decode video after rewind: %d\n", n);
    return 0;
}
