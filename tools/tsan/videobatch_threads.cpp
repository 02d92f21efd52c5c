#include <stdio.h>
#include <vector>
#include "mpeg.hpp"
extern "C" void *host_emu_batch_store(void);
using namespace mpeg;
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    std::vector<uint8_t> d(8 << 20);
    d.resize(fread(d.data(), 1, d.size(), f));
    const int n = 6;
    VideoBatch b(std::unique_ptr<BatchStore>(static_cast<BatchStore *>(host_emu_batch_store())), n);
    b.SetThreads(4);
    std::vector<std::unique_ptr<Buffer>> bufs;
    for (int i = 0; i < n; i++) {
        bufs.push_back(Buffer::FromMemory(d.data(), d.size()));
        b.AddStream(bufs.back().get());
    }
    std::vector<Frame *> frames;
    size_t total = 0;
    for (int t = 0; t < 40; t++)
        total += b.DecodeAll(frames, true);
    printf("frames %zu submits %llu\n", total, (unsigned long long)b.DeviceSubmits());
}
