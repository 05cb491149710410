// valu_rate_bench.hip - issue cost of the vector instructions the BVH node test could be built from,
// measured per SIMD with 1, 4 and 8 waves resident:  hipcc --offload-arch=gfx950 -O3 tools/valu_rate_bench.hip -o /tmp/valu && /tmp/valu
// Each kernel runs N dependent-free copies of one instruction per loop trip (inline asm, 16 independent
// accumulators) and reports cycles per instruction per SIMD = clocks / (instructions issued on that SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(512) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[16];
    uint64_t p[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = seed + i * 0x01010101u + threadIdx.x; p[i] = ((uint64_t)a[i] << 32) | (a[i] ^ 0x3c003c00u); }
    uint32_t s1 = seed | 0x3c003c00u, s2 = 0x03020100u ^ (seed & 0x03030303u);
    uint64_t q1 = ((uint64_t)0x3f800000u << 32) | 0x3f800000u;
    const uint32_t den = 0x00000037u + (seed & 1u), big = 0x5a800000u;  // a denormal f32 / f16 and a large multiplier
    const uint64_t sm = __builtin_amdgcn_read_exec() ^ (seed * 0x5555555555555555ull);
    for (int it = 0; it < iters; it++) {
#define FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s1), "v"(s2));
#define PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(q1));
#define PKFMA16(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s1), "v"(s2));
#define PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s1), "v"(s2));
#define CVTUB(i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]));
#define PKMAX16(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(s1));
#define PKMAX3(i) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s1), "v"(s2));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s1), "v"(s2));
#define CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s1));
#define LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s2), "v"(s1));
#define CMPSDWA(i) asm volatile("v_cmp_le_f16_sdwa vcc, %0, %1 src0_sel:WORD_0 src1_sel:WORD_1" : : "v"(a[i]), "v"(s1) : "vcc");
#define CMP32(i) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[i]), "v"(s1) : "vcc");
#define PKADD16(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(s1));
#define PKMUL32(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q1));
#define CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s1), "s"(sm));
#define CMPCND(i) asm volatile("v_cmp_le_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(s1), "v"(s2) : "vcc");
#define BFE(i) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
#define MUL32(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s1));
#define ADD32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s1));
#define ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s1), "v"(s2));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(s1));
#define MULSDWA(i) asm volatile("v_mul_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(big));
#define MULDEN(i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(den), "v"(big));
#define FMAMIX(i) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(big), "v"(s2));
#define FMAMIXDEN(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(den), "v"(big));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s1));
#define MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p[i]) : "v"(s1), "v"(s2) : "vcc");
#define MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(s1));
#define LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(a[i]) : "v"(s1));
#define LDEXP(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s2));
#define MIN3(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s1), "v"(s2));
#define MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s1));
#define LSHLSDWA(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(s2));
#define OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s1), "v"(s2));
#define FMA32B(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(s1), "v"(s2));
        if (OP == 0) { REP16(FMA32) }
        if (OP == 1) { REP16(PKFMA32) }
        if (OP == 2) { REP16(PKFMA16) }
        if (OP == 3) { REP16(PERM) }
        if (OP == 4) { REP16(CVTUB) }
        if (OP == 5) { REP16(PKMAX16) }
        if (OP == 6) { REP16(PKMAX3) }
        if (OP == 7) { REP16(MAX3) }
        if (OP == 8) { REP16(CNDMASK) }
        if (OP == 9) { REP16(LSHLOR) }
        if (OP == 10) { REP16(CMPSDWA) }
        if (OP == 11) { REP16(CMP32) }
        if (OP == 12) { REP16(PKADD16) }
        if (OP == 13) { REP16(PKMUL32) }
        if (OP == 14) { REP16(CND64) }
        if (OP == 15) { REP16(CMPCND) }
        if (OP == 16) { REP16(BFE) }
        if (OP == 17) { REP16(MUL32) }
        if (OP == 18) { REP16(ADD32) }
        if (OP == 19) { REP16(ANDOR) }
        if (OP == 20) { REP16(MOV) }
        if (OP == 21) { REP16(FMA32B) }
        if (OP == 22) { REP16(MULSDWA) }
        if (OP == 23) { REP16(MULDEN) }
        if (OP == 24) { REP16(FMAMIX) }
        if (OP == 25) { REP16(FMAMIXDEN) }
        if (OP == 26) { REP16(MULLO) }
        if (OP == 27) { REP16(MAD64) }
        if (OP == 28) { REP16(MUL24) }
        if (OP == 29) { REP16(LSHLADD) }
        if (OP == 30) { REP16(LDEXP) }
        if (OP == 31) { REP16(MIN3) }
        if (OP == 32) { REP16(MAXF) }
        if (OP == 33) { REP16(LSHLSDWA) }
        if (OP == 34) { REP16(OR3) }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r ^= a[i] ^ (uint32_t)p[i] ^ (uint32_t)(p[i] >> 32);
    if (r == 0x12345678u) out[0] = r;
}

template <int OP>
double run(int waves_per_simd, uint32_t* d_out, double clock_hz, int n_cus) {
    const int iters = 60000;
    const int threads = 64 * 4 * (waves_per_simd > 2 ? 2 : waves_per_simd);       // waves per block spread over the 4 SIMDs
    const int blocks_per_cu = waves_per_simd > 2 ? waves_per_simd / 2 : 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(n_cus * blocks_per_cu), dim3(threads), 0, 0, d_out, 100, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(n_cus * blocks_per_cu), dim3(threads), 0, 0, d_out, iters, 1u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 16.0 * waves_per_simd;
    return ms * 1e-3 * clock_hz / insts_per_simd;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double clk = prop.clockRate * 1e3;
    uint32_t* d;
    hipMalloc(&d, 64);
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_fma_f16", "v_perm_b32", "v_cvt_f32_ubyte1", "v_pk_max_f16", "v_pk_maximum3_f16",
                           "v_max3_f32", "v_cndmask_b32", "v_lshl_or_b32", "v_cmp_le_f16_sdwa", "v_cmp_le_f32", "v_pk_add_f16", "v_pk_mul_f32", "v_cndmask_b32_e64 (sgpr mask)", "v_cmp+v_cndmask (2 insts)", "v_bfe_u32", "v_mul_f32", "v_add_f32", "v_and_or_b32", "v_mov_b32", "v_fma_f32 (acc in src2)", "v_mul_f32_sdwa BYTE_1", "v_mul_f32 denormal src", "v_fma_mix_f32 (f16 lo src0)", "v_fma_mix_f32 denormal f16", "v_mul_lo_u32", "v_mad_u64_u32", "v_mul_u32_u24", "v_lshl_add_u32", "v_ldexp_f32", "v_min3_f32", "v_max_f32", "v_lshlrev_b32_sdwa", "v_or3_b32"};
    std::printf("clock %.0f MHz, %d CUs; cycles per instruction per SIMD (nominal clock) at 1 / 4 / 8 waves per SIMD\n", clk / 1e6, prop.multiProcessorCount);
#define ROW(OP) std::printf("%-30s %6.2f %6.2f %6.2f\n", names[OP], run<OP>(1, d, clk, prop.multiProcessorCount), run<OP>(4, d, clk, prop.multiProcessorCount), run<OP>(8, d, clk, prop.multiProcessorCount));
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(16) ROW(17) ROW(18) ROW(19) ROW(20) ROW(21) ROW(22) ROW(23) ROW(24) ROW(25) ROW(26) ROW(27) ROW(28) ROW(29) ROW(30) ROW(31) ROW(32) ROW(33) ROW(34)
    return 0;
}
