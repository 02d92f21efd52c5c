// Issue rate of wave64 VALU instructions on gfx950, per instruction kind: a loop of 64 independent
// instructions of one kind, enough waves to fill every SIMD.  Prints wave-instructions per clock per CU
// (4.0 = every SIMD issues one per clock; 1.0 = one per 4 clocks, the classic 16-lane SIMD figure).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int KIND> __global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t a[8], b = seed ^ threadIdx.x, c = seed * 3u + threadIdx.x;
    uint32_t sb = seed * 5u, sc = seed * 7u + 1u; // wave-uniform: scalar registers
    uint64_t mask = 0x5555555555555555ull * seed, mask2 = 0;
    for (int i = 0; i < 8; i++) a[i] = seed + i * 7 + threadIdx.x;
    if (KIND == 3 || KIND == 35 || KIND == 50 || KIND >= 56) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_mov_b64 s[20:21], vcc" : : "v"(b), "v"(c) : "vcc", "s20", "s21");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 1) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 2) asm volatile("v_lerp_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
                if (KIND == 4) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 5) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 6) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));
                if (KIND == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 8) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 9) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[i]));
                if (KIND == 10) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(uint64_t *)&a[i & 6]) : "v"(*(uint64_t *)&a[(i + 2) & 6]), "v"(*(uint64_t *)&a[(i + 4) & 6]));
                if (KIND == 12) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sb) : "s"(sc) : "scc");
                if (KIND == 13) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(sb) : "s"(sc));
                if (KIND == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
                if (KIND == 15) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(mask));
                if (KIND == 16) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 17) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 18) asm volatile("v_sat_pk_u8_i16 %0, %0" : "+v"(a[i]));
                if (KIND == 19) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 20) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 21) asm volatile("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(mask2) : "v"(a[i]), "v"(b));
                if (KIND == 22) asm volatile("v_mul_i32_i24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(a[i]) : "v"(b));
                if (KIND == 23) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a[i]));
                if (KIND == 24) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(*(uint64_t *)&a[i & 6]) : "v"(*(uint64_t *)&a[(i + 2) & 6]));
                if (KIND == 25) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 26) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(sc));
                if (KIND == 27) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"(sc));
                if (KIND == 28) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 29) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 30) asm volatile("v_lshrrev_b32 %0, 8, %0" : "+v"(a[i]));
                if (KIND == 31) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                // round 5: what decides between the two rates — the encoding, a scalar source, a literal, DPP?
                if (KIND == 32) asm volatile("v_and_b32 %0, 0x00ff00ff, %0" : "+v"(a[i]));
                if (KIND == 33) asm volatile("v_add_u32 %0, 5, %0" : "+v"(a[i]));
                if (KIND == 34) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if (KIND == 35) asm volatile("v_cndmask_b32_dpp %0, %0, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if (KIND == 36) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if (KIND == 37) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sc), "v"(b));
                if (KIND == 38) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "s"(sc));
                if (KIND == 39) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[i]) : "s"(sc));
                if (KIND == 40) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a[i]));
                if (KIND == 41) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 42) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 43) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 44) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
                if (KIND == 45) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 46) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 47) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 48) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 49) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 50) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
                if (KIND == 51) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 52) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7])); // (a second changing source)
                if (KIND == 53) asm volatile("v_lshrrev_b64 %0, 8, %0" : "+v"(*(uint64_t *)&a[i & 6]));
                if (KIND == 56) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
                if (KIND == 57) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
                if (KIND == 58) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 59) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : "s20", "s21");
                // ... one select among seven additions: is the VOP2 select's cost additive in a mix?
                if (KIND >= 60 && KIND <= 64 && i != 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 60 && i == 3) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
                if (KIND == 61 && i == 3) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
                if (KIND == 62 && i == 3) asm volatile("v_cndmask_b32_dpp %0, %0, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if (KIND == 63 && i == 3) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if (KIND == 64 && i == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 54) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sc), "v"(b));
                if (KIND == 55) asm volatile("v_mad_i32_i24 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));
            }
        }
    }
    uint32_t s = b ^ sb ^ (uint32_t)mask2;
    for (int i = 0; i < 8; i++) s ^= a[i];
    if (s == 0x12345) out[threadIdx.x] = s;
}
template <int KIND> void run(const char *name, uint32_t *out, int cus, double ghz)
{
    const int iters = 2000, blocks = cus * 8; // 8 blocks x 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 64;
    printf("%-16s %8.3f ms  %6.3f wave-instr/clk/CU at %.2f GHz (%.2f per SIMD)\n", name, ms, winstr / (ms * 1e-3) / (ghz * 1e9) / cus, ghz,
           winstr / (ms * 1e-3) / (ghz * 1e9) / cus / 4);
}
int main()
{
    uint32_t *out; hipMalloc(&out, 4096);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double ghz = p.clockRate / 1e6;
    printf("%s: %d CUs, %.2f GHz\n", p.gcnArchName, cus, ghz);
    run<0>("v_add_u32", out, cus, ghz); run<1>("v_mul_i32_i24", out, cus, ghz); run<2>("v_lerp_u8", out, cus, ghz);
    run<3>("v_cndmask_b32", out, cus, ghz); run<4>("v_mul_lo_u32", out, cus, ghz); run<5>("v_pk_add_i16", out, cus, ghz);
    run<6>("v_alignbyte_b32", out, cus, ghz); run<7>("v_fma_f32", out, cus, ghz); run<8>("v_med3_i32", out, cus, ghz);
    run<9>("v_ashrrev_i32", out, cus, ghz); run<10>("v_mad_i32_i24", out, cus, ghz); run<11>("v_pk_fma_f32", out, cus, ghz);
    run<12>("s_add_u32", out, cus, ghz); run<13>("s_mul_i32", out, cus, ghz);
    run<14>("v_mov_b32 v,v", out, cus, ghz); run<26>("v_mov_b32 v,s", out, cus, ghz); run<15>("v_cndmask_e64 sgpr", out, cus, ghz);
    run<16>("v_and_b32", out, cus, ghz); run<31>("v_xor_b32", out, cus, ghz); run<25>("v_sub_u32", out, cus, ghz);
    run<27>("v_add_u32 s,v", out, cus, ghz); run<30>("v_lshrrev_b32", out, cus, ghz); run<17>("v_perm_b32", out, cus, ghz);
    run<18>("v_sat_pk_u8_i16", out, cus, ghz); run<28>("v_cvt_pk_i16_i32", out, cus, ghz); run<19>("v_add3_u32", out, cus, ghz);
    run<20>("v_lshl_add_u32", out, cus, ghz); run<21>("v_cmp_lt_i32_e64", out, cus, ghz); run<22>("v_mul_i24_sdwa", out, cus, ghz);
    run<23>("v_bfe_u32", out, cus, ghz); run<24>("v_lshl_add_u64", out, cus, ghz); run<29>("v_bitop3_b32", out, cus, ghz);
    run<32>("v_and_b32 literal", out, cus, ghz); run<33>("v_add_u32 inline", out, cus, ghz); run<40>("v_add_u32 literal", out, cus, ghz);
    run<38>("v_and_b32 s,v", out, cus, ghz); run<39>("v_lshlrev_b32 s,v", out, cus, ghz); run<49>("v_add_u32_e64", out, cus, ghz);
    run<34>("v_mov_b32_dpp", out, cus, ghz); run<35>("v_cndmask_dpp", out, cus, ghz); run<36>("v_add_u32_dpp", out, cus, ghz);
    run<50>("v_cndmask_e32 vcc", out, cus, ghz); run<37>("v_mad_u32_u16 s", out, cus, ghz); run<41>("v_max_i32", out, cus, ghz);
    run<42>("v_or_b32", out, cus, ghz); run<43>("v_mul_u32_u24", out, cus, ghz); run<44>("v_cmp_lt_u32_e32", out, cus, ghz);
    run<45>("v_lshl_or_b32", out, cus, ghz); run<46>("v_and_or_b32", out, cus, ghz); run<47>("v_pk_mad_i16", out, cus, ghz);
    run<48>("v_alignbit_b32", out, cus, ghz); run<51>("v_bfi_b32", out, cus, ghz); run<52>("v_sub_u32 2 srcs", out, cus, ghz);
    run<53>("v_lshrrev_b64", out, cus, ghz); run<56>("v_cndmask_e64 vcc", out, cus, ghz); run<57>("v_addc_co_u32 vcc", out, cus, ghz); run<58>("v_cndmask d!=s vcc", out, cus, ghz); run<59>("v_cndmask_e64 s[20:21]", out, cus, ghz);
    run<64>("8 v_add_u32", out, cus, ghz); run<60>("7 add + cndmask_e32", out, cus, ghz); run<61>("7 add + cndmask_e64", out, cus, ghz);
    run<62>("7 add + cndmask_dpp", out, cus, ghz); run<63>("7 add + mov_dpp", out, cus, ghz);
    run<54>("v_mad_i24 v,s,v", out, cus, ghz); run<55>("v_mad_i24 +inline", out, cus, ghz);
    return 0;
}
