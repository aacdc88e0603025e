// Micro-benchmark (measurement tool, not product code): issue cost of the VALU / transcendental / MFMA instructions
// the fused bf16 pipeline is made of, alone and interleaved, on gfx950.  Every wave runs ITER iterations of an
// unrolled block of independent instructions; cycles per wave-instruction per SIMD = elapsed * clock / count, and
// the v_fma_f32 row (2 cycles by construction on a SIMD-32) calibrates the clock.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/ubench/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = 0.001f * (threadIdx.x + i);
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * i); b[i] = (__bf16)(0.02f * i); }
    for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(r[i]));
#define PKM(i) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*(double *)&r[(i) & 6]));
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(r[i]));
#define EXH(i) asm volatile("v_exp_f16 %0, %0" : "+v"(r[i]));
#define RCH(i) asm volatile("v_rcp_f16 %0, %0" : "+v"(r[i]));
#define PKH(i) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(r[i]));
        if (MODE == 0) { REP8(FMA) REP8(FMA) }
        if (MODE == 1) { REP8(EXP) REP8(EXP) }
        if (MODE == 2) { REP8(RCP) REP8(RCP) }
        if (MODE == 3) { REP8(MUL) REP8(MUL) }
        if (MODE == 4) { REP8(PKM) REP8(PKM) }
        if (MODE == 5) { REP8(CVT) REP8(CVT) }
        if (MODE == 6) { REP8(EXH) REP8(EXH) }
        if (MODE == 7) { REP8(RCH) REP8(RCH) }
        if (MODE == 8) { REP8(PKH) REP8(PKH) }
        if (MODE == 9) {  // swish body: mul, exp, add, rcp, mul on 8 independent values (40 instr counted as 16 "units")
#define SW(i) asm volatile("v_mul_f32 %1, 0xbfb8aa3b, %0\n v_exp_f32 %1, %1\n v_add_f32 %1, 1.0, %1\n v_rcp_f32 %1, %1\n v_mul_f32 %0, %0, %1" : "+v"(r[i]), "=&v"(r[(i + 4) & 7]));
            SW(0) SW(1) SW(2) SW(3)
        }
        if (MODE == 10 || MODE == 11 || MODE == 12) {  // 8 bf16 MFMAs (2 accumulators) [+ 8 exp | + 16 fma]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
                if (MODE == 11) { EXP(0) EXP(1) }
                if (MODE == 12) { FMA(0) FMA(1) FMA(2) FMA(3) }
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
                if (MODE == 11) { EXP(2) EXP(3) }
                if (MODE == 12) { FMA(4) FMA(5) FMA(6) FMA(7) }
            }
        }
        if (MODE == 13) {  // 8 MFMAs + 4 full swish bodies (20 VALU incl. 8 transcendental)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
            }
            SW(0) SW(1) SW(2) SW(3)
        }
    }
    float s = acc0[0] + acc1[1];
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
double run(float *d, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *d; hipMalloc(&d, sizeof(float) * 256 * cus * 8);
    const int iters = 20000;
    const char *names[] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_mul_f32", "v_pk_mul_f32", "v_cvt_pk_bf16_f32", "v_exp_f16", "v_rcp_f16",
                           "v_pk_fma_f16", "swish x4 (20 instr)", "8 mfma16x16x32bf16", "8 mfma + 4 exp... (8 exp)", "8 mfma + 32 fma", "8 mfma + 4 swish"};
    const double per_iter[] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 20, 8, 8, 8, 8};
    for (int wps = 1; wps <= 2; ++wps) {  // waves per SIMD
        const int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
        double t[14];
        t[0] = run<0>(d, blocks, iters); t[1] = run<1>(d, blocks, iters); t[2] = run<2>(d, blocks, iters); t[3] = run<3>(d, blocks, iters);
        t[4] = run<4>(d, blocks, iters); t[5] = run<5>(d, blocks, iters); t[6] = run<6>(d, blocks, iters); t[7] = run<7>(d, blocks, iters);
        t[8] = run<8>(d, blocks, iters); t[9] = run<9>(d, blocks, iters); t[10] = run<10>(d, blocks, iters); t[11] = run<11>(d, blocks, iters);
        t[12] = run<12>(d, blocks, iters); t[13] = run<13>(d, blocks, iters);
        // clock from the fma row: 16 instr/iter * wps waves per SIMD * 2 cycles
        const double clk = 16.0 * iters * wps * 2.0 / t[0];
        printf("waves/SIMD=%d  calibrated clock %.2f GHz (assuming v_fma_f32 = 2 cycles/wave64)\n", wps, clk * 1e-9);
        for (int m = 0; m < 14; ++m) {
            const double cyc_per_iter_per_wave = t[m] * clk / iters / wps;
            printf("  %-28s %8.3f ms  %7.2f cycles per unit (%g units/iter)  %8.1f cycles/iter/wave\n", names[m], t[m] * 1e3,
                   cyc_per_iter_per_wave / per_iter[m], per_iter[m], cyc_per_iter_per_wave);
        }
    }
    return 0;
}
