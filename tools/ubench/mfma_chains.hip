// Micro-benchmark (measurement tool, not product code): shader cycles per v_mfma_f32_16x16x32_bf16 on gfx950 as a function
// of (a) how many INDEPENDENT accumulator chains a wave interleaves, (b) whether an LDS read sits between the MFMAs (the
// shape of the fused front kernel's tile loops: one ds_read_b128 per MFMA), (c) one or two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chains tools/ubench/mfma_chains.hip && /tmp/mfma_chains
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, int LDS_READS>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters) {
    __shared__ uint4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    __syncthreads();
    const uint4 a = make_uint4(0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    uint4 b[8];
    for (int i = 0; i < 8; ++i) b[i] = lds[(threadIdx.x + 64 * i) & 1023];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint4 *row = lds + (threadIdx.x & 63);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {  // 16 rounds of CHAINS MFMAs
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (LDS_READS) b[c] = row[64 * ((u * CHAINS + c + it) & 15)];
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b[c]), acc[c], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// grouped schedule: per round and chain the G reads of the next group are issued together, then G MFMAs on the SAME
// accumulator back to back (nothing between dependent MFMAs), then the other chain's G
template <int CHAINS, int G>
__global__ __launch_bounds__(256) void kg(float *out, unsigned long long *cyc, int iters) {
    __shared__ uint4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    __syncthreads();
    const uint4 a = make_uint4(0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    f32x4 acc[CHAINS];
    for (int i = 0; i < CHAINS; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint4 *row = lds + (threadIdx.x & 63);
    uint4 cur[CHAINS][G], nxt[CHAINS][G];
    for (int c = 0; c < CHAINS; ++c)
        for (int g = 0; g < G; ++g) cur[c][g] = row[64 * ((c * G + g) & 15)];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / G; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c)
#pragma unroll
                for (int g = 0; g < G; ++g) nxt[c][g] = row[64 * ((u * CHAINS * G + c * G + g + it) & 15)];
#pragma unroll
            for (int c = 0; c < CHAINS; ++c)
#pragma unroll
                for (int g = 0; g < G; ++g)
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, cur[c][g]), acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, CHAINS * G, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, CHAINS * G, 0);
#pragma unroll
            for (int c = 0; c < CHAINS; ++c)
#pragma unroll
                for (int g = 0; g < G; ++g) cur[c][g] = nxt[c][g];
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < CHAINS; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CHAINS, int G>
void rung(float *d, unsigned long long *dc, int cus, int wps, int iters) {
    const int blocks = cus * wps;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((kg<CHAINS, G>), dim3(blocks), dim3(256), 0, 0, d, dc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double n = 16.0 * CHAINS * iters;
    printf("  grouped: chains %d, %d reads then %d MFMAs per chain, waves/SIMD %d : %7.2f cycles per MFMA per wave -> %7.2f per SIMD\n", CHAINS, G,
           G, wps, sum / h.size() / n, sum / h.size() / n / wps);
}

template <int CHAINS, int LDS_READS>
void run(float *d, unsigned long long *dc, int cus, int wps, int iters) {
    const int blocks = cus * wps;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<CHAINS, LDS_READS>), dim3(blocks), dim3(256), 0, 0, d, dc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double n = 16.0 * CHAINS * iters;
    printf("  chains %d  lds_read_per_mfma %d  waves/SIMD %d : %7.2f cycles per MFMA per wave -> %7.2f per SIMD\n", CHAINS, LDS_READS, wps,
           sum / h.size() / n, sum / h.size() / n / wps);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float *d; (void)hipMalloc(&d, sizeof(float) * 256 * cus * 4);
    unsigned long long *dc; (void)hipMalloc(&dc, 8 * 4 * cus * 4);
    for (int wps = 1; wps <= 2; ++wps) {
        const int it = 500;
        run<1, 0>(d, dc, cus, wps, it); run<2, 0>(d, dc, cus, wps, it); run<3, 0>(d, dc, cus, wps, it); run<4, 0>(d, dc, cus, wps, it);
        run<8, 0>(d, dc, cus, wps, it);
        run<1, 1>(d, dc, cus, wps, it); run<2, 1>(d, dc, cus, wps, it); run<4, 1>(d, dc, cus, wps, it); run<8, 1>(d, dc, cus, wps, it);
        rung<1, 4>(d, dc, cus, wps, it); rung<2, 4>(d, dc, cus, wps, it); rung<2, 2>(d, dc, cus, wps, it); rung<1, 8>(d, dc, cus, wps, it);
    }
    return 0;
}
