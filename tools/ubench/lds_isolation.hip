// lds_isolation.hip — do workgroups of DIFFERENT processes that share a CU keep their LDS (and registers) to themselves?
// Round 4: pipelines of different dtypes running in separate processes on one GPU disturbed each other's logits
// (tools/stress_determinism.py), alone or next to their own kind they are bit-stable.  Every block of this kernel fills
// its dynamic LDS and a few registers with a pattern derived from (tag, block, word), keeps checking them for a while, and
// counts the words that changed.  Run several instances side by side with different LDS sizes / tags:
//   bin/lds_isolation <tag> <lds_kib> <threads> <blocks_per_cu> <millis> [write_every]
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

__global__ void lds_guard(unsigned tag, int words, int rounds, unsigned long long *bad, unsigned long long *seen_foreign) {
    extern __shared__ unsigned smem[];
    const unsigned base = tag * 0x9E3779B1u + blockIdx.x * 0x85EBCA77u;
    unsigned keep[8];
    for (int k = 0; k < 8; ++k) keep[k] = base ^ (threadIdx.x * 31u + k);
    unsigned long long local_bad = 0, foreign = 0;
    for (int r = 0; r < rounds; ++r) {
        const unsigned salt = base + (unsigned)r * 0xC2B2AE3Du;
        for (int i = threadIdx.x; i < words; i += blockDim.x) smem[i] = salt ^ (unsigned)i;
        __syncthreads();
        for (int rep = 0; rep < 8; ++rep) {
            for (int i = threadIdx.x; i < words; i += blockDim.x) {
                const unsigned v = smem[i];
                if (v != (salt ^ (unsigned)i)) {
                    ++local_bad;
                    if ((v ^ (unsigned)i) != salt) ++foreign;
                }
            }
            __builtin_amdgcn_s_sleep(32);
        }
        for (int k = 0; k < 8; ++k)
            if (keep[k] != (base ^ (threadIdx.x * 31u + k))) ++local_bad;
        __syncthreads();
    }
    if (local_bad) atomicAdd(bad, local_bad);
    if (foreign) atomicAdd(seen_foreign, foreign);
}

int main(int argc, char **argv) {
    const unsigned tag = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
    const int kib = argc > 2 ? atoi(argv[2]) : 64, threads = argc > 3 ? atoi(argv[3]) : 256, bpc = argc > 4 ? atoi(argv[4]) : 2;
    const int millis = argc > 5 ? atoi(argv[5]) : 3000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * bpc, words = kib * 256;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(lds_guard), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    unsigned long long *bad, h[2] = {0, 0};
    CHECK(hipMalloc(reinterpret_cast<void **>(&bad), 16));
    CHECK(hipMemset(bad, 0, 16));
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() < millis) {
        hipLaunchKernelGGL(lds_guard, dim3(grid), dim3(threads), (size_t)kib * 1024, 0, tag, words, 20, bad, bad + 1);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        ++launches;
    }
    CHECK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    printf("tag %u: %d KiB LDS x %d threads, %d blocks, %ld launches: %llu words changed under the block (%llu of them not its own pattern)\n", tag, kib,
           threads, grid, launches, h[0], h[1]);
    return h[0] ? 3 : 0;
}
