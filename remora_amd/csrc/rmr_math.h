// Device math shared by the kernels: the gate / activation functions on the raw hardware
// transcendentals (v_exp_f32 = 2^x, v_rcp_f32: 1 ulp each, no denormal fix-up sequences, no
// IEEE division) — ~2 VALU instructions per exp or reciprocal instead of ~10.
// Error budget: the networks' logits must match the reference within 1e-4 (fp32); these
// functions are accurate to ~2 ulp relative (sigmoid, swish) / ~1e-7 absolute (tanh).
#pragma once
#include <hip/hip_runtime.h>

namespace rmr {

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// 1 / (1 + e^-x): e^-x -> +inf gives rcp(inf) = 0, -> 0 gives 1
__device__ __forceinline__ float sigmoid_f(float x) { return fast_rcp(1.0f + fast_exp(-x)); }
// 1 - 2 / (1 + e^{2x}); saturates to +-1
__device__ __forceinline__ float tanh_f(float x) { return fmaf(-2.0f, fast_rcp(1.0f + fast_exp(2.0f * x)), 1.0f); }
// x * sigmoid(x)   (src/remora/activations.py:4-18)
__device__ __forceinline__ float swish_f(float x) { return x * sigmoid_f(x); }

// Packed fp32 (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two IEEE operations per lane and instruction, same
// rounding as the scalar forms).  v_mfma_f32_16x16x4_f32 shares the SIMD's fp32 datapath with the VALU, so every
// VALU instruction saved in the fp32 pipeline is matrix time gained.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_splat(float x) { return f32x2{x, x}; }
// pk_fma: the multiply-add of the VALU signal producers (front_sig_kernel, sig3_front_kernel).  As v_pk_fma_f32 those two
// kernels - and only they - returned damaged activations in a few per cent of the calls while ANOTHER process kept the bf16
// matrix cores busy (tools/ubench/neighbour mfma16; tools/stress_determinism.py): one lane's result off in scattered chunks,
// never alone on the GPU, never beside VALU / LDS / HBM / fp32-MFMA loads, and never when built as the two v_fma_f32 below
// (0 of 600 calls against 27 of 300; profiles/NOTES_r04.md has the whole account - a stand-alone v_pk_fma_f32 check beside
// the same neighbour stays clean, so the cause is not pinned to the instruction).  The scalar pair is the shipped form:
// bit-identical, front_sig_kernel 1.67 against 1.62 ns per chunk, sig3_front_kernel 5.81 against 5.47 (no longer the default
// producer, k_conv_front.hip).  -DRMR_PACKED_F32_FMA restores the packed one.
#ifndef RMR_PACKED_F32_FMA
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    float x, y;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x) : "v"(a.x), "v"(b.x), "v"(c.x));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(a.y), "v"(b.y), "v"(c.y));
    return f32x2{x, y};
}
#elif RMR_PACKED_F32_FMA + 0 == 2
// experiment build: the packed instruction under the SAME scheduling constraints as the shipped scalar pair (volatile asm):
// if this form is damaged beside a bf16-MFMA tenant and the scalar pair is not, the difference is the instruction, not the
// schedule the compiler picks around it (profiles/NOTES_r05.md)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#else
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#endif
// swish of four values: the multiplies and the add packed, the four exp / rcp stay scalar (no packed transcendentals);
// element for element the same operations as swish_f
__device__ __forceinline__ void swish_pk(f32x2 &a, f32x2 &b) {
    const f32x2 ta = a * pk_splat(-1.4426950408889634f), tb = b * pk_splat(-1.4426950408889634f);
    const f32x2 da = f32x2{__builtin_amdgcn_exp2f(ta.x), __builtin_amdgcn_exp2f(ta.y)} + pk_splat(1.0f);
    const f32x2 db = f32x2{__builtin_amdgcn_exp2f(tb.x), __builtin_amdgcn_exp2f(tb.y)} + pk_splat(1.0f);
    a = a * f32x2{__builtin_amdgcn_rcpf(da.x), __builtin_amdgcn_rcpf(da.y)};
    b = b * f32x2{__builtin_amdgcn_rcpf(db.x), __builtin_amdgcn_rcpf(db.y)};
}

// ---- RMR_SYNC: the block barrier of the model kernels -----------------------------------------------------------------
// Plain __syncthreads() in the shipped library.  `make jitter` (-DRMR_JITTER, libremora_hip_jitter.so) puts a per-wave
// pseudo-random sleep in front of and behind every barrier and in front of every intra-wave LDS hand-off (wave_sync): most
// waves go straight on, a fifth are held for 0.2-3 us, one in thirty-two for 7-14 us - longer than a whole
// stage of any of these kernels.  A hand-off through LDS that relies on timing instead of a barrier (round 4's LSTM bug:
// a wave still reading x_0 while the stager wrote x_2, visible only when foreign waves held it up) then goes wrong with
// the kernel ALONE on the GPU; a correctly synchronised kernel returns the same bits as the shipped build
// (tests/test_gpu_jitter.py, tools/stress_determinism.py --jitter).
#ifdef RMR_JITTER
__device__ __forceinline__ void jitter() {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    const unsigned id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));  // HW_ID: wave / SIMD / CU / SE of this wave
    unsigned h = ((unsigned)t ^ (unsigned)(t >> 21) ^ (id * 0x9E3779B9u)) * 2654435761u;
    h = (unsigned)__builtin_amdgcn_readfirstlane((int)h) >> 26;  // 0..63, one value per wave (unsigned: the builtin returns int,
                                                                 // whose arithmetic shift made half the draws a 4-billion-round sleep)
    if (h >= 62) {
        for (unsigned i = 61; i < h; ++i) {  // 1..2 x 2 x 8128 cycles: 7-14 us
            __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
        }
    } else if (h >= 48) {
        for (unsigned i = 47; i < h; ++i) __builtin_amdgcn_s_sleep(8);  // 1..14 x 512 cycles: 0.2-3 us
    }
}
#define RMR_SYNC()           \
    do {                     \
        ::rmr::jitter();     \
        __syncthreads();     \
        ::rmr::jitter();     \
    } while (0)
#define RMR_JITTER_POINT() ::rmr::jitter()
#else
#define RMR_SYNC() __syncthreads()
#define RMR_JITTER_POINT() ((void)0)
#endif

}  // namespace rmr
