// agpr_isolation.hip — do a wave's accumulation registers (AGPRs) survive while OTHER processes run kernels on the same GPU?
// Round 4: every kernel of this library that the compiler gave AGPRs (sig3_front_kernel, lstm_bf16s_kernel, conv_bf16s_kernel)
// returned damaged results in a few per cent of the calls when another process used the GPU at the same time; kernels with
// agpr_count 0 never did (tools/stress_determinism.py, profiles/NOTES_r04.md).  Each thread parks a pattern in 16 AGPRs and 16
// VGPRs, idles, reads them back and counts the registers that changed.
//   bin/agpr_isolation <tag> <threads> <blocks_per_cu> <millis> [neighbour: 0 = park registers, 1 = plain VALU/LDS load only]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

#define AW(i, v) asm volatile("v_accvgpr_write_b32 a" #i ", %0" ::"v"(v) : "a" #i)
#define AR(i, v) asm volatile("v_accvgpr_read_b32 %0, a" #i : "=v"(v) : : "a" #i)

__global__ __launch_bounds__(256) void agpr_guard(unsigned tag, int rounds, unsigned long long *bad) {
    const unsigned base = tag * 0x9E3779B1u + blockIdx.x * 0x85EBCA77u + threadIdx.x * 0xC2B2AE3Du;
    unsigned long long bad_a = 0, bad_v = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned s = base + (unsigned)r * 0x27D4EB2Fu;
        AW(0, s + 0); AW(1, s + 1); AW(2, s + 2); AW(3, s + 3); AW(4, s + 4); AW(5, s + 5); AW(6, s + 6); AW(7, s + 7);
        AW(8, s + 8); AW(9, s + 9); AW(10, s + 10); AW(11, s + 11); AW(12, s + 12); AW(13, s + 13); AW(14, s + 14); AW(15, s + 15);
        unsigned keep[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            keep[k] = s ^ (0x1000u + k);
            asm volatile("" : "+v"(keep[k]));  // a live VGPR per entry
        }
        for (int i = 0; i < 64; ++i) __builtin_amdgcn_s_sleep(64);
        unsigned v;
#define CK(i) AR(i, v); asm volatile("s_nop 4"); if (v != s + i) ++bad_a;
        CK(0) CK(1) CK(2) CK(3) CK(4) CK(5) CK(6) CK(7) CK(8) CK(9) CK(10) CK(11) CK(12) CK(13) CK(14) CK(15)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            asm volatile("" : "+v"(keep[k]));
            if (keep[k] != (s ^ (0x1000u + k))) ++bad_v;
        }
    }
    if (bad_a) atomicAdd(bad, bad_a);
    if (bad_v) atomicAdd(bad + 1, bad_v);
}

__global__ __launch_bounds__(256) void busy(float *out, int rounds) {  // a neighbour without AGPRs: VALU + LDS traffic
    __shared__ float sm[4096];
    float x = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        sm[(threadIdx.x * 17 + r) & 4095] = x;
        __syncthreads();
        x = x * 1.0001f + sm[(threadIdx.x * 29 + r) & 4095];
    }
    if (x == 12345.f) out[0] = x;
}

int main(int argc, char **argv) {
    const unsigned tag = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
    const int threads = argc > 2 ? atoi(argv[2]) : 256, bpc = argc > 3 ? atoi(argv[3]) : 2, millis = argc > 4 ? atoi(argv[4]) : 3000;
    const int neighbour = argc > 5 ? atoi(argv[5]) : 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * bpc;
    unsigned long long *bad, h[2] = {0, 0};
    CHECK(hipMalloc(reinterpret_cast<void **>(&bad), 16));
    CHECK(hipMemset(bad, 0, 16));
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() < millis) {
        if (neighbour) hipLaunchKernelGGL(busy, dim3(grid), dim3(threads), 0, 0, reinterpret_cast<float *>(bad), 2000);
        else hipLaunchKernelGGL(agpr_guard, dim3(grid), dim3(threads), 0, 0, tag, 8, bad);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        ++launches;
    }
    CHECK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    if (neighbour) printf("tag %u: neighbour (no AGPRs), %ld launches\n", tag, launches);
    else printf("tag %u: %d blocks x %d threads, %ld launches: %llu AGPRs and %llu VGPRs changed under their wave\n", tag, grid, threads, launches, h[0], h[1]);
    return (h[0] || h[1]) ? 3 : 0;
}
