// pk_f32_check.hip — do the packed fp32 VALU instructions give the results of their scalar forms while ANOTHER process keeps the
// bf16 matrix cores of the same SIMDs busy?  Round 4: sig3_front_kernel / front_sig_kernel (v_pk_fma_f32 producers) returned
// damaged activations only next to a process issuing v_mfma_f32_16x16x32_bf16 (tools/ubench/neighbour mfma16); built with
// v_fma_f32 pairs instead they never did (profiles/NOTES_r04.md).  Every thread runs the same chain of operations twice - packed
// and scalar - and counts the rounds whose bits differ.
//   bin/pk_f32_check <millis> [blocks_per_cu] [partner: 0 none, 1 bf16 MFMA waves in the same block, 2 fp32 MFMA waves]
//   (or start  bin/neighbour mfma16 <millis>  beside it)
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

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sfma(float a, float b, float c) {
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float smul(float a, float b) {
    float r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float sadd(float a, float b) {
    float r;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// counts[0]: v_pk_fma_f32 with a broadcast operand (op_sel), [1]: v_pk_fma_f32 on two-element operands, [2]: v_pk_mul_f32, [3]: v_pk_add_f32
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// partner = 1: blocks of 8 waves, waves 0-3 (one per SIMD) issue v_mfma_f32_16x16x32_bf16 back to back while waves 4-7 of the
// same block - their SIMD partners - run the comparison; partner = 2: the partners issue v_mfma_f32_16x16x4_f32 instead
template <int PARTNER>
__global__ __launch_bounds__(512) void pk_check(unsigned long long *counts, int rounds, unsigned seed, float *sink) {
    if (PARTNER && threadIdx.x < 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * threadIdx.x + i); b[i] = (__bf16)(0.5f - 0.001f * threadIdx.x); }
        for (int r = 0; r < rounds * 3; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (PARTNER == 1) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[0], (float)b[0], acc, 0, 0, 0);
            }
        }
        if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
        return;
    }
    const unsigned id = blockIdx.x * 256 + (threadIdx.x & 255);
    unsigned long long bad[4] = {0, 0, 0, 0};
    float w[8][2];
    for (int i = 0; i < 8; ++i) {
        w[i][0] = 0.37f + 0.011f * (float)((id * 7 + i * 13 + seed) & 63);
        w[i][1] = -0.41f + 0.009f * (float)((id * 5 + i * 29 + seed) & 63);
    }
    for (int r = 0; r < rounds; ++r) {
        const float x0 = 0.001f * (float)((id + r * 977u) & 1023) - 0.5f;
        f32x2 p = {0.1f, -0.2f}, pv = {0.3f, 0.05f}, pm = {1.0f, 1.0f}, pa = {0.f, 0.f};
        float s0 = 0.1f, s1 = -0.2f, v0 = 0.3f, v1 = 0.05f, m0 = 1.0f, m1 = 1.0f, a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = x0 + 0.01f * i;
            const f32x2 wv = {w[i][0], w[i][1]};
            p = __builtin_elementwise_fma(wv, f32x2{x, x}, p);       // broadcast operand
            s0 = sfma(w[i][0], x, s0);
            s1 = sfma(w[i][1], x, s1);
            pv = __builtin_elementwise_fma(wv, f32x2{x, -x}, pv);    // two-element operand
            v0 = sfma(w[i][0], x, v0);
            v1 = sfma(w[i][1], -x, v1);
            pm = pm * (wv + f32x2{0.7f, 1.4f});
            m0 = smul(m0, sadd(w[i][0], 0.7f));
            m1 = smul(m1, sadd(w[i][1], 1.4f));
            pa = pa + wv * f32x2{x, x};
            a0 = sadd(a0, smul(w[i][0], x));
            a1 = sadd(a1, smul(w[i][1], x));
        }
        bad[0] += (__float_as_uint(p.x) != __float_as_uint(s0)) | (__float_as_uint(p.y) != __float_as_uint(s1));
        bad[1] += (__float_as_uint(pv.x) != __float_as_uint(v0)) | (__float_as_uint(pv.y) != __float_as_uint(v1));
        bad[2] += (__float_as_uint(pm.x) != __float_as_uint(m0)) | (__float_as_uint(pm.y) != __float_as_uint(m1));
        bad[3] += (__float_as_uint(pa.x) != __float_as_uint(a0)) | (__float_as_uint(pa.y) != __float_as_uint(a1));
    }
    for (int k = 0; k < 4; ++k)
        if (bad[k]) atomicAdd(counts + k, bad[k]);
}

int main(int argc, char **argv) {
    const int millis = argc > 1 ? atoi(argv[1]) : 3000, bpc = argc > 2 ? atoi(argv[2]) : 4, partner = argc > 3 ? atoi(argv[3]) : 0;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * bpc, rounds = 2000;
    unsigned long long *counts, h[4];
    CHECK(hipMalloc(reinterpret_cast<void **>(&counts), 64));
    CHECK(hipMemset(counts, 0, 64));
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() < millis) {
        if (partner == 1) hipLaunchKernelGGL(pk_check<1>, dim3(grid), dim3(512), 0, 0, counts, rounds, (unsigned)launches, reinterpret_cast<float *>(counts + 4));
        else if (partner == 2) hipLaunchKernelGGL(pk_check<2>, dim3(grid), dim3(512), 0, 0, counts, rounds, (unsigned)launches, reinterpret_cast<float *>(counts + 4));
        else hipLaunchKernelGGL(pk_check<0>, dim3(grid), dim3(256), 0, 0, counts, rounds, (unsigned)launches, reinterpret_cast<float *>(counts + 4));
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        ++launches;
    }
    CHECK(hipMemcpy(h, counts, 32, hipMemcpyDeviceToHost));
    const double total = (double)launches * grid * 256 * rounds;
    printf("partner %d: %ld launches, %.3g thread-rounds of 8 operations: rounds whose packed result differs from the scalar one: "
           "pk_fma(broadcast) %llu, pk_fma %llu, pk_mul %llu, pk_add(+mul) %llu\n", partner, launches, total, h[0], h[1], h[2], h[3]);
    return (h[0] || h[1] || h[2] || h[3]) ? 3 : 0;
}
