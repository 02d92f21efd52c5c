/* Exhaustive check of the short division used by the audio kernel (mpeg_amd/csrc/audio_lane.h,
 * scale_short): for EVERY float32 x that passes scale_short_ok, Markstein's sequence equals the
 * IEEE quotient x / -1090519040 bit for bit.  Build: gcc -O2 -ffp-contract=off [-mfma] -pthread.
 * usage: div_const <threads>; exit status 0 = proved, prints the number of inputs checked. */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const float kScale = -1090519040.0f;
static int n_threads;
static uint64_t bad[256], checked[256], skipped[256];

static int short_ok(float x) { return !(fabsf(x) < 0x1p-95f) || x == 0.0f; }

static void *run(void *arg)
{
    const int id = (int)(intptr_t)arg;
    const float y = 1.0f / kScale;
    const uint64_t lo = (uint64_t)id * (1ull << 32) / n_threads, hi = (uint64_t)(id + 1) * (1ull << 32) / n_threads;
    uint64_t b = 0, c = 0, s = 0;
    for (uint64_t u = lo; u < hi; u++) {
        const uint32_t w = (uint32_t)u;
        float x;
        memcpy(&x, &w, 4);
        if (x != x || isinf(x) || !short_ok(x)) { /* NaN / inf cannot occur (see audio_lane.h); the band takes the long path */
            s++;
            continue;
        }
        volatile float want = x / kScale;
        const float q = x * y;
        const float r = fmaf(-q, kScale, x);
        const float got = fmaf(r, y, q);
        const float w2 = want;
        b += memcmp(&w2, &got, 4) != 0;
        c++;
    }
    bad[id] = b;
    checked[id] = c;
    skipped[id] = s;
    return 0;
}

int main(int argc, char **argv)
{
    n_threads = argc > 1 ? atoi(argv[1]) : 8;
    if (n_threads < 1 || n_threads > 256)
        n_threads = 8;
    pthread_t t[256];
    for (int i = 0; i < n_threads; i++)
        pthread_create(&t[i], 0, run, (void *)(intptr_t)i);
    uint64_t b = 0, c = 0, s = 0;
    for (int i = 0; i < n_threads; i++) {
        pthread_join(t[i], 0);
        b += bad[i];
        c += checked[i];
        s += skipped[i];
    }
    printf("checked %llu skipped %llu mismatches %llu\n", (unsigned long long)c, (unsigned long long)s, (unsigned long long)b);
    return b != 0;
}
