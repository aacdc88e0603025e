// neighbour.hip — a synthetic process that loads ONE kind of resource on every CU, to find what a co-tenant must do to disturb
// a kernel of this library (tools/stress_determinism.py, profiles/NOTES_r04.md).
//   bin/neighbour <kind> <millis> [blocks_per_cu] [lds_kib]     kind: valu | trans | lds | mfma16 | mfma32 | hbm | sleep
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void load_kernel(float *buf, size_t words, int rounds) {
    extern __shared__ float sm[];
    float x = threadIdx.x * 0.001f + 1.0f, y = 0.5f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rounds; ++r) {
        if (KIND == 0) {  // plain VALU
#pragma unroll
            for (int i = 0; i < 64; ++i) x = fmaf(x, 1.000001f, y);
        } else if (KIND == 1) {  // transcendental unit
#pragma unroll
            for (int i = 0; i < 16; ++i) x = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 0.001f));
        } else if (KIND == 2) {  // LDS traffic
            sm[(threadIdx.x * 33 + r) & 8191] = x;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) x += sm[(threadIdx.x * 17 + i * 257 + r) & 8191];
            __syncthreads();
        } else if (KIND == 3) {  // bf16 matrix cores
            bf16x8 a, b;
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)x; b[i] = (__bf16)y; }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        } else if (KIND == 4) {  // fp32 matrix cores
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);
        } else if (KIND == 5) {  // HBM streaming
            const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x + (size_t)r * gridDim.x * 256) % words;
            x += buf[i];
            buf[i] = x;
        } else {
            __builtin_amdgcn_s_sleep(64);
        }
    }
    x += acc[0] + acc[1] + acc[2] + acc[3];
    if (x == 12345.678f) buf[0] = x;
}

int main(int argc, char **argv) {
    const char *kind = argc > 1 ? argv[1] : "valu";
    const int millis = argc > 2 ? atoi(argv[2]) : 3000, bpc = argc > 3 ? atoi(argv[3]) : 2, lds_kib = argc > 4 ? atoi(argv[4]) : 32;
    const char *names[] = {"valu", "trans", "lds", "mfma16", "mfma32", "hbm", "sleep"};
    int k = -1;
    for (int i = 0; i < 7; ++i)
        if (!strcmp(kind, names[i])) k = i;
    if (k < 0) { fprintf(stderr, "unknown kind %s\n", kind); return 2; }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * bpc;
    const size_t words = (size_t)1 << 28;
    float *buf;
    CHECK(hipMalloc(reinterpret_cast<void **>(&buf), words * 4));
    CHECK(hipMemset(buf, 0, words * 4));
    void (*kern[])(float *, size_t, int) = {load_kernel<0>, load_kernel<1>, load_kernel<2>, load_kernel<3>, load_kernel<4>, load_kernel<5>, load_kernel<6>};
    for (auto f : kern) CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(f), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() < millis) {
        hipLaunchKernelGGL(kern[k], dim3(grid), dim3(256), (size_t)lds_kib * 1024, 0, buf, words, 400);
        CHECK(hipGetLastError());
        if ((launches & 7) == 7) CHECK(hipDeviceSynchronize());
        ++launches;
    }
    CHECK(hipDeviceSynchronize());
    printf("neighbour %s: %ld launches of %d blocks, %d KiB LDS each\n", kind, launches, grid, lds_kib);
    return 0;
}
