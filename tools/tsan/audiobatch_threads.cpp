#include <stdio.h>
#include <string.h>
#include <vector>
#include "mpeg.hpp"
extern "C" void *host_emu_audio_batch_store(void);
extern "C" void host_emu_configure(int flavour, const float *window512);
using namespace mpeg;
int main(int argc, char **argv)
{ // mpeg::AudioBatch with a parse pool: six copies of an MP2 stream, every tick parsed on four threads into the batch's slots
    FILE *f = fopen(argv[1], "rb");
    std::vector<uint8_t> d(8 << 20);
    d.resize(fread(d.data(), 1, d.size(), f));
    float window[512];
    for (int i = 0; i < 512; i++)
        window[i] = (float)((i * 37 % 101) - 50) / 64.0f; // (any table: the run is about the threads, not the samples)
    host_emu_configure(0, window);
    const int n = 6;
    AudioBatch b(std::unique_ptr<AudioBatchStore>(static_cast<AudioBatchStore *>(host_emu_audio_batch_store())), n, AudioF32N, 0);
    b.SetThreads(4);
    std::vector<std::unique_ptr<Buffer>> bufs;
    for (int i = 0; i < n; i++) {
        bufs.push_back(Buffer::FromMemory(d.data(), d.size()));
        b.AddStream(bufs.back().get());
    }
    std::vector<Samples *> samples;
    size_t total = 0;
    for (int t = 0; t < 120; t++)
        total += b.DecodeAll(samples);
    printf("sample blocks %zu device calls %llu\n", total, (unsigned long long)b.DeviceCalls());
}
