// Micro-benchmark (measurement tool, not product code): shader cycles per wave64 instruction on gfx950, from
// s_memtime around an unrolled block of INDEPENDENT instructions (16 registers round-robin), one or two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_cycles tools/ubench/valu_cycles.hip && /tmp/valu_cycles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters) {
    float r[16];
    for (int i = 0; i < 16; ++i) r[i] = 1.0f + 0.001f * (threadIdx.x + i);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(r[i]));
#define ADD(i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(r[i]));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(r[i]));
#define AND(i) asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(r[i]));
#define SHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r[i]));
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(r[i]));
#define CND64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %0, s[10:11]" : "+v"(r[i]));
#define CNDI(i) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(r[i]));
#define CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %0" : : "v"(r[i]) : "vcc");
#define MED3(i) asm volatile("v_med3_f32 %0, %0, %0, %0" : "+v"(r[i]));
#define MINF(i) asm volatile("v_min_f32 %0, %0, %0" : "+v"(r[i]));
#define EXPH(i) asm volatile("v_exp_f16 %0, %0" : "+v"(r[i]));
#define PKF(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double *)&r[(i) & 14]));
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i]));
#define LOG(i) asm volatile("v_log_f32 %0, %0" : "+v"(r[i]));
#define FMA_EXP(i) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_exp_f32 %1, %1" : "+v"(r[i]), "+v"(r[(i + 8) & 15]));
        if (MODE == 0) { R16(FMA) R16(FMA) R16(FMA) R16(FMA) }
        if (MODE == 1) { R16(MUL) R16(MUL) R16(MUL) R16(MUL) }
        if (MODE == 2) { R16(ADD) R16(ADD) R16(ADD) R16(ADD) }
        if (MODE == 3) { R16(EXP) R16(EXP) R16(EXP) R16(EXP) }
        if (MODE == 4) { R16(RCP) R16(RCP) R16(RCP) R16(RCP) }
        if (MODE == 5) { R16(CVT) R16(CVT) R16(CVT) R16(CVT) }
        if (MODE == 6) { R16(AND) R16(AND) R16(AND) R16(AND) }
        if (MODE == 7) { R16(SHL) R16(SHL) R16(SHL) R16(SHL) }
        if (MODE == 8) { R16(CND) R16(CND) R16(CND) R16(CND) }
        if (MODE == 9) { R16(EXPH) R16(EXPH) R16(EXPH) R16(EXPH) }
        if (MODE == 10) { R16(PKF) R16(PKF) R16(PKF) R16(PKF) }
        if (MODE == 11) { R16(SQRT) R16(SQRT) R16(SQRT) R16(SQRT) }
        if (MODE == 12) { R16(LOG) R16(LOG) R16(LOG) R16(LOG) }
        if (MODE == 13) { R16(FMA_EXP) R16(FMA_EXP) }  // 32 fma + 32 exp interleaved (64 instr)
        if (MODE == 14) { R16(CND64) R16(CND64) R16(CND64) R16(CND64) }
        if (MODE == 15) { R16(CNDI) R16(CNDI) R16(CNDI) R16(CNDI) }
        if (MODE == 16) { R16(CMP) R16(CMP) R16(CMP) R16(CMP) }
        if (MODE == 17) { R16(MED3) R16(MED3) R16(MED3) R16(MED3) }
        if (MODE == 18) { R16(MINF) R16(MINF) R16(MINF) R16(MINF) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char *name, float *d, unsigned long long *dc, int cus, int wps, int iters) {
    const int blocks = cus * wps;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0, mn = 1e30;
    for (auto v : h) { sum += (double)v; mn = v < mn ? (double)v : mn; }
    const double instr = 64.0 * iters;
    // s_memtime counts at a fixed 100 MHz on this family?  report raw ticks per instruction AND relative to fma
    printf("  %-22s wps=%d  avg %8.3f  min %8.3f ticks/instr/wave (x wps = SIMD issue share)\n", name, wps, sum / h.size() / instr,
           mn / instr);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *d; (void)hipMalloc(&d, sizeof(float) * 256 * cus * 4);
    unsigned long long *dc; (void)hipMalloc(&dc, 8 * 4 * cus * 4);
    printf("clockRate %d kHz, wallclock rate %d kHz\n", p.clockRate, p.clockInstructionRate);
    for (int wps = 1; wps <= 2; ++wps) {
        const int it = 2000;
        run<0>("v_fma_f32", d, dc, cus, wps, it); run<1>("v_mul_f32", d, dc, cus, wps, it); run<2>("v_add_f32", d, dc, cus, wps, it);
        run<3>("v_exp_f32", d, dc, cus, wps, it); run<4>("v_rcp_f32", d, dc, cus, wps, it); run<5>("v_cvt_pk_bf16_f32", d, dc, cus, wps, it);
        run<6>("v_and_b32", d, dc, cus, wps, it); run<7>("v_lshlrev_b32", d, dc, cus, wps, it); run<8>("v_cndmask_b32", d, dc, cus, wps, it);
        run<9>("v_exp_f16", d, dc, cus, wps, it); run<10>("v_pk_fma_f32", d, dc, cus, wps, it); run<11>("v_sqrt_f32", d, dc, cus, wps, it);
        run<12>("v_log_f32", d, dc, cus, wps, it); run<13>("fma+exp interleaved", d, dc, cus, wps, it);
        run<14>("v_cndmask_b32_e64 sgpr", d, dc, cus, wps, it); run<15>("v_cndmask_b32 0,v,vcc", d, dc, cus, wps, it); run<16>("v_cmp_lt_f32 vcc", d, dc, cus, wps, it);
        run<17>("v_med3_f32", d, dc, cus, wps, it); run<18>("v_min_f32", d, dc, cus, wps, it);
    }
    // wall-clock cross-check of the tick unit: time the fma kernel with events
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<0>, dim3(cus), dim3(256), 0, 0, d, dc, 200000);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(4);
    (void)hipMemcpy(h.data(), dc, 32, hipMemcpyDeviceToHost);
    printf("fma kernel 200000 iters: %.3f ms wall, %llu ticks -> %.1f MHz tick rate; %.3f ns per wave-instruction\n", ms, h[0],
           (double)h[0] / (ms * 1e-3) * 1e-6, ms * 1e6 / (64.0 * 200000));
    return 0;
}
