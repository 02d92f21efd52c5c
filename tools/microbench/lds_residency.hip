// How many 256-thread workgroups does a CU really keep resident for a given LDS size (and ~100 VGPRs)?
// hipOccupancyMaxActiveBlocksPerMultiprocessor answers from a formula (lds_occupancy.hip); this one measures: N x 256
// workgroups (N per CU) each spin for a fixed number of clocks; if all are resident at once the launch takes one spin,
// otherwise two.  Prints the launch time for N = 4, 5, 6 and several LDS sizes.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int kVgprPad> __global__ __launch_bounds__(256) void spin(float *o, long long clocks)
{
    extern __shared__ float s[];
    float r[kVgprPad];
#pragma unroll
    for (int i = 0; i < kVgprPad; i++)
        r[i] = o[i] * (float)threadIdx.x; // live registers across the spin
    s[threadIdx.x] = 1;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < clocks)
        __builtin_amdgcn_s_sleep(8);
    float acc = s[255 - threadIdx.x];
#pragma unroll
    for (int i = 0; i < kVgprPad; i++)
        acc += r[i];
    if (acc == 12345.0f)
        o[threadIdx.x] = acc;
}
template <int kVgprPad> static void run(const char *what, float *d)
{
    for (int b : {24576, 28672, 30720, 31744, 32000, 32256, 32632, 32768}) {
        printf("%s, LDS %6d B:", what, b);
        for (int per_cu : {4, 5, 6}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            spin<kVgprPad><<<256 * per_cu, 256, b>>>(d, 100000); // warm
            hipEventRecord(e0);
            spin<kVgprPad><<<256 * per_cu, 256, b>>>(d, 5000000); // 50 ms at 100 MHz wall clock
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("  %d per CU: %.1f ms", per_cu, ms);
        }
        printf("\n");
    }
}
int main()
{
    float *d;
    hipMalloc(&d, 1 << 16);
    hipMemset(d, 0, 1 << 16);
    run<8>("few registers", d);
    run<72>("~100 registers", d);
    return 0;
}
