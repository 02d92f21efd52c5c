// What does the 8 x 8 transposition across the 8 lanes of a block (rc_transpose8: lane j holds column j, leaves with row j) cost
// on gfx950, by the way its two in-quad exchange steps are written?  tools/microbench/valu_rate.hip shows a stream of VOP2 selects
// (v_cndmask_b32 / v_cndmask_b32_dpp, which read VCC implicitly) issuing at one per ~23 clocks per SIMD, where the VOP3 form of the
// same select issues at one per ~4 — and a single VOP2 select among other instructions costing nothing extra.  The kernel's quad step
// is 2 x 4 v_cndmask_b32_dpp back to back.  Variants (each checked against the transposition it must produce, then timed with 8 waves
// per SIMD, `fill` plain additions per register between two transpositions as the IDCT passes would supply):
//   A  the kernel's form: 8 v_cndmask_b32_dpp per quad step (VCC set / inverted by scalar moves)
//   B  8 v_mov_b32_dpp + 8 v_cndmask_b32_e64 with the lane mask in a scalar pair
//   C  8 v_mov_b32_dpp + 8 v_bfi_b32 with the lane mask in a vector register
//   D  A with one independent v_add_u32 between every two selects


// usage: transpose_cost
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define QUAD_STEP_A(PERM)                                                                                                                       \
    asm volatile("s_nop 1\n\ts_mov_b64 vcc, %[set]\n\t"                                                                                        \
                 "v_cndmask_b32_dpp %[nb0], %[a0], %[b0], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                          \
                 "v_cndmask_b32_dpp %[nb1], %[a1], %[b1], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                          \
                 "v_cndmask_b32_dpp %[nb2], %[a2], %[b2], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                          \
                 "v_cndmask_b32_dpp %[nb3], %[a3], %[b3], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                          \
                 "s_not_b64 vcc, vcc\n\t"                                                                                                      \
                 "v_cndmask_b32_dpp %[na0], %[b0], %[a0], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                          \
                 "v_cndmask_b32_dpp %[na1], %[b1], %[a1], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                          \
                 "v_cndmask_b32_dpp %[na2], %[b2], %[a2], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                          \
                 "v_cndmask_b32_dpp %[na3], %[b3], %[a3], vcc " PERM " row_mask:0xf bank_mask:0xf"                                               \
                 : [na0] "=&v"(na[0]), [na1] "=&v"(na[1]), [na2] "=&v"(na[2]), [na3] "=&v"(na[3]), [nb0] "=&v"(nb[0]), [nb1] "=&v"(nb[1]),      \
                   [nb2] "=&v"(nb[2]), [nb3] "=&v"(nb[3])                                                                                       \
                 : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]),              \
                   [b3] "v"(b[3]), [set] "s"(set_lanes)                                                                                         \
                 : "vcc", "scc")
// D: the same selects, an independent addition after each (x is a scratch accumulator)
#define QUAD_STEP_D(PERM)                                                                                                                       \
    asm volatile("s_nop 1\n\ts_mov_b64 vcc, %[set]\n\t"                                                                                        \
                 "v_cndmask_b32_dpp %[nb0], %[a0], %[b0], vcc " PERM " row_mask:0xf bank_mask:0xf\n\tv_add_u32 %[x], %[x], %[a0]\n\t"           \
                 "v_cndmask_b32_dpp %[nb1], %[a1], %[b1], vcc " PERM " row_mask:0xf bank_mask:0xf\n\tv_add_u32 %[x], %[x], %[a1]\n\t"           \
                 "v_cndmask_b32_dpp %[nb2], %[a2], %[b2], vcc " PERM " row_mask:0xf bank_mask:0xf\n\tv_add_u32 %[x], %[x], %[a2]\n\t"           \
                 "v_cndmask_b32_dpp %[nb3], %[a3], %[b3], vcc " PERM " row_mask:0xf bank_mask:0xf\n\tv_add_u32 %[x], %[x], %[a3]\n\t"           \
                 "s_not_b64 vcc, vcc\n\t"                                                                                                      \
                 "v_cndmask_b32_dpp %[na0], %[b0], %[a0], vcc " PERM " row_mask:0xf bank_mask:0xf\n\tv_add_u32 %[x], %[x], %[b0]\n\t"           \
                 "v_cndmask_b32_dpp %[na1], %[b1], %[a1], vcc " PERM " row_mask:0xf bank_mask:0xf\n\tv_add_u32 %[x], %[x], %[b1]\n\t"           \
                 "v_cndmask_b32_dpp %[na2], %[b2], %[a2], vcc " PERM " row_mask:0xf bank_mask:0xf\n\tv_add_u32 %[x], %[x], %[b2]\n\t"           \
                 "v_cndmask_b32_dpp %[na3], %[b3], %[a3], vcc " PERM " row_mask:0xf bank_mask:0xf\n\tv_add_u32 %[x], %[x], %[b3]"                \
                 : [na0] "=&v"(na[0]), [na1] "=&v"(na[1]), [na2] "=&v"(na[2]), [na3] "=&v"(na[3]), [nb0] "=&v"(nb[0]), [nb1] "=&v"(nb[1]),      \
                   [nb2] "=&v"(nb[2]), [nb3] "=&v"(nb[3]), [x] "+v"(x)                                                                          \
                 : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]),              \
                   [b3] "v"(b[3]), [set] "s"(set_lanes)                                                                                         \
                 : "vcc", "scc")
// B: DPP moves, then VOP3 selects on a scalar pair (set lanes take the partner's a into a; clear lanes the partner's ... see below)
#define QUAD_STEP_B(PERM)                                                                                                                       \
    asm volatile("s_nop 1\n\t"                                                                                                                  \
                 "v_mov_b32_dpp %[ta0], %[a0] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_mov_b32_dpp %[ta1], %[a1] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_mov_b32_dpp %[ta2], %[a2] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_mov_b32_dpp %[ta3], %[a3] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_mov_b32_dpp %[tb0], %[b0] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_mov_b32_dpp %[tb1], %[b1] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_mov_b32_dpp %[tb2], %[b2] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_mov_b32_dpp %[tb3], %[b3] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_cndmask_b32_e64 %[nb0], %[ta0], %[b0], %[set]\n\t"                                                                          \
                 "v_cndmask_b32_e64 %[nb1], %[ta1], %[b1], %[set]\n\t"                                                                          \
                 "v_cndmask_b32_e64 %[nb2], %[ta2], %[b2], %[set]\n\t"                                                                          \
                 "v_cndmask_b32_e64 %[nb3], %[ta3], %[b3], %[set]\n\t"                                                                          \
                 "v_cndmask_b32_e64 %[na0], %[a0], %[tb0], %[set]\n\t"                                                                          \
                 "v_cndmask_b32_e64 %[na1], %[a1], %[tb1], %[set]\n\t"                                                                          \
                 "v_cndmask_b32_e64 %[na2], %[a2], %[tb2], %[set]\n\t"                                                                          \
                 "v_cndmask_b32_e64 %[na3], %[a3], %[tb3], %[set]"                                                                              \
                 : [na0] "=&v"(na[0]), [na1] "=&v"(na[1]), [na2] "=&v"(na[2]), [na3] "=&v"(na[3]), [nb0] "=&v"(nb[0]), [nb1] "=&v"(nb[1]),      \
                   [nb2] "=&v"(nb[2]), [nb3] "=&v"(nb[3]), [ta0] "=&v"(ta[0]), [ta1] "=&v"(ta[1]), [ta2] "=&v"(ta[2]), [ta3] "=&v"(ta[3]),      \
                   [tb0] "=&v"(tb[0]), [tb1] "=&v"(tb[1]), [tb2] "=&v"(tb[2]), [tb3] "=&v"(tb[3])                                               \
                 : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]),              \
                   [b3] "v"(b[3]), [set] "s"(set_lanes))
// C: the selects as bit-field inserts on a vector mask (all ones in the set lanes)
#define QUAD_STEP_C(PERM)                                                                                                                       \
    asm volatile("s_nop 1\n\t"                                                                                                                  \
                 "v_mov_b32_dpp %[ta0], %[a0] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_mov_b32_dpp %[ta1], %[a1] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_mov_b32_dpp %[ta2], %[a2] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_mov_b32_dpp %[ta3], %[a3] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_mov_b32_dpp %[tb0], %[b0] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_mov_b32_dpp %[tb1], %[b1] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_mov_b32_dpp %[tb2], %[b2] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_mov_b32_dpp %[tb3], %[b3] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                          \
                 "v_bfi_b32 %[nb0], %[m], %[b0], %[ta0]\n\t"                                                                                   \
                 "v_bfi_b32 %[nb1], %[m], %[b1], %[ta1]\n\t"                                                                                   \
                 "v_bfi_b32 %[nb2], %[m], %[b2], %[ta2]\n\t"                                                                                   \
                 "v_bfi_b32 %[nb3], %[m], %[b3], %[ta3]\n\t"                                                                                   \
                 "v_bfi_b32 %[na0], %[m], %[tb0], %[a0]\n\t"                                                                                   \
                 "v_bfi_b32 %[na1], %[m], %[tb1], %[a1]\n\t"                                                                                   \
                 "v_bfi_b32 %[na2], %[m], %[tb2], %[a2]\n\t"                                                                                   \
                 "v_bfi_b32 %[na3], %[m], %[tb3], %[a3]"                                                                                       \
                 : [na0] "=&v"(na[0]), [na1] "=&v"(na[1]), [na2] "=&v"(na[2]), [na3] "=&v"(na[3]), [nb0] "=&v"(nb[0]), [nb1] "=&v"(nb[1]),      \
                   [nb2] "=&v"(nb[2]), [nb3] "=&v"(nb[3]), [ta0] "=&v"(ta[0]), [ta1] "=&v"(ta[1]), [ta2] "=&v"(ta[2]), [ta3] "=&v"(ta[3]),      \
                   [tb0] "=&v"(tb[0]), [tb1] "=&v"(tb[1]), [tb2] "=&v"(tb[2]), [tb3] "=&v"(tb[3])                                               \
                 : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]),              \
                   [b3] "v"(b[3]), [m] "v"(vmask))

template <int kVariant> __device__ __forceinline__ void transpose8(int32_t (&v)[8], uint32_t lane, uint32_t &x)
{
    (void)x;
    {
        const int32_t a[4] = {v[0], v[2], v[4], v[6]}, b[4] = {v[1], v[3], v[5], v[7]};
        int32_t na[4], nb[4], ta[4], tb[4];
        (void)ta, (void)tb;
        const uint64_t set_lanes = 0xAAAAAAAAAAAAAAAAull;
        const uint32_t vmask = 0u - (lane & 1u);
        (void)vmask;
        if (kVariant == 0) QUAD_STEP_A("quad_perm:[1,0,3,2]");
        if (kVariant == 1) QUAD_STEP_B("quad_perm:[1,0,3,2]");
        if (kVariant == 2) QUAD_STEP_C("quad_perm:[1,0,3,2]");
        if (kVariant == 3) QUAD_STEP_D("quad_perm:[1,0,3,2]");
        v[0] = na[0], v[2] = na[1], v[4] = na[2], v[6] = na[3];
        v[1] = nb[0], v[3] = nb[1], v[5] = nb[2], v[7] = nb[3];
    }
    {
        const int32_t a[4] = {v[0], v[1], v[4], v[5]}, b[4] = {v[2], v[3], v[6], v[7]};
        int32_t na[4], nb[4], ta[4], tb[4];
        (void)ta, (void)tb;
        const uint64_t set_lanes = 0xCCCCCCCCCCCCCCCCull;
        const uint32_t vmask = 0u - ((lane >> 1) & 1u);
        (void)vmask;
        if (kVariant == 0) QUAD_STEP_A("quad_perm:[2,3,0,1]");
        if (kVariant == 1) QUAD_STEP_B("quad_perm:[2,3,0,1]");
        if (kVariant == 2) QUAD_STEP_C("quad_perm:[2,3,0,1]");
        if (kVariant == 3) QUAD_STEP_D("quad_perm:[2,3,0,1]");
        v[0] = na[0], v[1] = na[1], v[4] = na[2], v[5] = na[3];
        v[2] = nb[0], v[3] = nb[1], v[6] = nb[2], v[7] = nb[3];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int32_t a = v[r], b = v[r + 4];
        v[r + 4] = __builtin_amdgcn_update_dpp(b, a, 0x104, 0xf, 0x5, false);
        v[r] = __builtin_amdgcn_update_dpp(a, b, 0x114, 0xf, 0xa, false);
    }
}

template <int kVariant> __global__ __launch_bounds__(64) void check(int32_t *out)
{
    const uint32_t lane = threadIdx.x;
    int32_t v[8];
    uint32_t x = 0;
    for (int r = 0; r < 8; r++)
        v[r] = (int32_t)((lane >> 3) * 1000 + (lane & 7) * 10 + r); // block g, column j, row r
    transpose8<kVariant>(v, lane, x);
    for (int c = 0; c < 8; c++)
        out[lane * 8 + c] = v[c]; // lane (g, j) must hold row j: v[c] = value of (column c, row j)
}

template <int kVariant, int kFill> __global__ __launch_bounds__(256) void timed(int32_t *out, int iters, int32_t seed)
{
    const uint32_t lane = threadIdx.x & 63;
    int32_t v[8];
    uint32_t x = (uint32_t)seed;
    for (int r = 0; r < 8; r++)
        v[r] = seed + (int32_t)threadIdx.x * 8 + r;
    for (int it = 0; it < iters; it++) {
        if (kVariant >= 0)
            transpose8 < kVariant<0 ? 0 : kVariant>(v, lane, x);
#pragma unroll
        for (int f = 0; f < kFill; f++)
#pragma unroll
            for (int r = 0; r < 8; r++)
                asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[r]) : "v"(v[(r + 1) & 7]));
    }
    int32_t s = (int32_t)x;
    for (int r = 0; r < 8; r++)
        s ^= v[r];
    if (s == 0x1234567)
        out[threadIdx.x] = s;
}

template <int kVariant, int kFill> static double run(int32_t *out, int cus)
{
    const int iters = 4000, blocks = cus * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((timed<kVariant, kFill>), dim3(blocks), dim3(256), 0, 0, out, 50, 1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((timed<kVariant, kFill>), dim3(blocks), dim3(256), 0, 0, out, iters, 1);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / iters / 8; // ns per (iteration of one wave) per SIMD slot: 8 waves share a SIMD
}

template <int kVariant> static int verify(const char *name, int32_t *d)
{
    int32_t h[512];
    hipLaunchKernelGGL(check<kVariant>, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++)
        for (int c = 0; c < 8; c++)
            bad += h[lane * 8 + c] != (lane >> 3) * 1000 + c * 10 + (lane & 7);
    printf("%-58s %s\n", name, bad ? "WRONG" : "transposes");
    return bad;
}

int main()
{
    int32_t *d;
    (void)hipMalloc(&d, 1 << 16);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    int bad = verify<0>("A 8 v_cndmask_b32_dpp per quad step (the kernel's form)", d);
    bad += verify<1>("B 8 v_mov_b32_dpp + 8 v_cndmask_b32_e64 (scalar-pair mask)", d);
    bad += verify<2>("C 8 v_mov_b32_dpp + 8 v_bfi_b32 (vector mask)", d);
    bad += verify<3>("D as A, an independent v_add_u32 after every select", d);
    printf("ns per wave-iteration per SIMD (8 waves per SIMD); fill = plain additions per register between two transpositions\n");
    printf("%-10s %10s %10s %10s %10s %10s\n", "fill", "none", "A", "B", "C", "D");
#define ROW(F) printf("%-10d %10.1f %10.1f %10.1f %10.1f %10.1f\n", F, run<-1, F>(d, cus), run<0, F>(d, cus), run<1, F>(d, cus), run<2, F>(d, cus), run<3, F>(d, cus))
    ROW(0);
    ROW(4);
    ROW(12);
    return bad != 0;
}
