// Sanitizer harness (see run.sh): a lone Video over the damaged golden stream with the look-ahead and the HOST MIRROR on — the lane
// emulator's rc_mirror_mb writes every frame's linear copy (emu_wide_chunk), Decode hands the copies out — through rewinds; the
// planes' FNV-1a-64 must be the reference's (mpeg_test.go:227) whichever way the frames come back.
#include <stdio.h>
#include <vector>
#include "mpeg.hpp"
extern "C" void *host_emu_video_backend(int flavour);
using namespace mpeg;
static uint64_t run(const std::vector<uint8_t> &d, bool mirror, bool lookahead, bool preroll)
{
    std::unique_ptr<Buffer> buf = Buffer::FromMemory(d.data(), d.size());
    Video v(buf.get(), std::unique_ptr<VideoBackend>(static_cast<VideoBackend *>(host_emu_video_backend(0))));
    v.SetHostMirror(mirror);
    v.SetLookahead(lookahead);
    if (preroll) { // a few frames, a rewind in the middle of a GOP with a picture parsed ahead, then the whole stream
        for (int i = 0; i < 7; i++)
            v.Decode();
        v.Rewind();
    }
    uint64_t h = 0xcbf29ce484222325ull;
    int n = 0;
    while (Frame *f = v.Decode()) {
        for (const Plane *p : {&f->Y, &f->Cb, &f->Cr})
            for (size_t i = 0; i < p->Len; i++)
                h = (h ^ p->Data[i]) * 0x100000001b3ull;
        n++;
    }
    printf("mirror %d lookahead %d preroll %d: %d frames, hash %016llx\n", (int)mirror, (int)lookahead, (int)preroll, n, (unsigned long long)h);
    return h;
}
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    std::vector<uint8_t> d(8 << 20);
    d.resize(fread(d.data(), 1, d.size(), f));
    // a fresh decoder: the reference's hash, whichever way the frames come back; after a rewind in mid-stream (the damaged stream's
    // first pictures then find other leftovers in the frame store): the same hash in all four modes
    int bad = 0;
    const uint64_t rewound = run(d, false, false, true);
    for (int mode = 0; mode < 4; mode++) {
        bad += run(d, mode & 1, mode & 2, false) != 0xea6d7fcb1340ba3full;
        bad += run(d, mode & 1, mode & 2, true) != rewound;
    }
    printf("%s\n", bad ? "MISMATCH" : "all modes agree");
    return bad;
}
