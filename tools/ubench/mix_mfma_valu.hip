// Micro-benchmark (measurement tool, not product code): do a wave's matrix-core instructions and ANOTHER wave's VALU
// instructions on the same SIMD run side by side on gfx950?  A block of 8 waves puts waves w and w + 4 on one SIMD; each
// half of the block gets a role (0 idle, 1 MFMA: 4 independent v_mfma_f32_16x16x32_bf16 chains, 2 plain VALU: v_fma_f32 on 16
// registers, 3 transcendental: v_exp_f32 on 16 registers, 4 MFMA with a ds_read_b128 in front of each, 5 mixed epilogue-like
// stream 2 exp + 2 rcp + 4 plain; 6 / 7 the same with v_mfma_f32_32x32x16_bf16, two chains).  Ticks per instruction of each role, alone and next to the other.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mix tools/ubench/mix_mfma_valu.hip && /tmp/mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int ROLE>
__device__ __noinline__ float role_body(int iters, const uint4 *row) {
    float r[16];
    for (int i = 0; i < 16; ++i) r[i] = 1.0f + 0.001f * (threadIdx.x + i);
    const uint4 a = make_uint4(0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    uint4 b[4];
    for (int i = 0; i < 4; ++i) b[i] = row[64 * i];
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    f32x16 big[2];
    for (int c = 0; c < 2; ++c)
        for (int i = 0; i < 16; ++i) big[c][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (ROLE == 1 || ROLE == 4) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (ROLE == 4) b[c] = row[64 * ((u * 4 + c + it) & 15)];
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b[c]), acc[c], 0, 0, 0);
                }
        } else if (ROLE == 6 || ROLE == 7) {  // v_mfma_f32_32x32x16_bf16: twice the flops per instruction, two chains
#pragma unroll
            for (int u = 0; u < 32; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (ROLE == 7) b[c] = row[64 * ((u * 2 + c + it) & 15)];
                    big[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b[c]), big[c], 0, 0, 0);
                }
        } else if (ROLE == 2) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
            R16(FMA) R16(FMA) R16(FMA) R16(FMA)
        } else if (ROLE == 3) {
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
            R16(EXP) R16(EXP) R16(EXP) R16(EXP)
        } else if (ROLE == 5) {  // 64 instructions: 8 x (2 exp, 2 rcp, 2 add, 1 mul, 1 cvt)
#define EPI(i) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_add_f32 %0, 1.0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_mul_f32 %2, %0, %1\n v_cvt_pk_bf16_f32 %3, %2, %2\n v_add_f32 %1, 1.0, %1" : "+v"(r[(i)]), "+v"(r[(i) + 8]), "+v"(r[((i) + 1) & 7]), "+v"(r[8 + (((i) + 1) & 7)]));
            EPI(0) EPI(1) EPI(2) EPI(3) EPI(4) EPI(5) EPI(6) EPI(7)
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += r[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (ROLE == 6 || ROLE == 7) s += big[0][0] + big[1][5];
    return s;
}

__global__ void warm(float *out, int iters) {  // clocks up before anything is timed
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0000001f, 0.5f);
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

template <int LO, int HI>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters) {
    __shared__ uint4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 512) lds[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    __syncthreads();
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint4 *row = lds + (threadIdx.x & 63);
    const unsigned long long t0 = __builtin_readcyclecounter();
    float s = 0;
    if (w < 4) { if (LO) s = role_body<LO>(iters, row); }
    else { if (HI) s = role_body<HI>(iters, row); }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

static const char *names[8] = {"idle", "mfma x4 chains", "v_fma_f32", "v_exp_f32", "mfma + ds_read_b128", "epilogue mix", "mfma 32x32x16 x2", "mfma32 + ds_read_b128"};

template <int LO, int HI>
void run(float *d, unsigned long long *dc, int cus, int iters) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<LO, HI>), dim3(cus), dim3(512), 0, 0, d, dc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(cus * 8);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;
    for (int b = 0; b < cus; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += (double)h[b * 8 + w];
    const double n = 64.0 * iters * cus * 4;
    printf("  waves 0-3: %-20s %7.3f   | waves 4-7: %-20s %7.3f\n", names[LO], LO ? lo / n : 0.0, names[HI], HI ? hi / n : 0.0);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 4000;
    float *d; (void)hipMalloc(&d, sizeof(float) * 512 * cus);
    unsigned long long *dc; (void)hipMalloc(&dc, 8 * 8 * cus);
    hipLaunchKernelGGL(warm, dim3(cus * 8), dim3(256), 0, 0, d, 40000000 / 64);
    (void)hipDeviceSynchronize();
    printf("one block of 8 waves per CU (waves w and w + 4 share a SIMD); ticks of s_memtime per instruction and wave\n");
    run<1, 0>(d, dc, cus, iters); run<2, 0>(d, dc, cus, iters); run<3, 0>(d, dc, cus, iters); run<4, 0>(d, dc, cus, iters); run<5, 0>(d, dc, cus, iters);
    run<1, 1>(d, dc, cus, iters); run<2, 2>(d, dc, cus, iters); run<3, 3>(d, dc, cus, iters); run<5, 5>(d, dc, cus, iters); run<4, 4>(d, dc, cus, iters);
    run<1, 2>(d, dc, cus, iters); run<1, 3>(d, dc, cus, iters); run<1, 5>(d, dc, cus, iters);
    run<4, 2>(d, dc, cus, iters); run<4, 3>(d, dc, cus, iters); run<4, 5>(d, dc, cus, iters);
    run<6, 0>(d, dc, cus, iters); run<7, 0>(d, dc, cus, iters); run<6, 6>(d, dc, cus, iters); run<7, 7>(d, dc, cus, iters); run<7, 5>(d, dc, cus, iters);
    return 0;
}
