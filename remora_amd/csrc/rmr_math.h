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

}  // namespace rmr
