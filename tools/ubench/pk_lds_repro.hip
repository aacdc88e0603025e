// pk_lds_repro.hip - candidate reproducer for round 4's open item: the VALU signal producers (front_sig_kernel,
// sig3_front_kernel) returned damaged activations in the upper half-wave when built on v_pk_fma_f32, only while another
// process kept the bf16 matrix cores busy; pk_f32_check.hip (packed against scalar on register operands) stayed clean.
// What those two kernels have and that check lacks: the packed FMA's broadcast operand comes STRAIGHT FROM AN LDS READ
// (ds_read_b32 -> s_waitcnt -> v_pk_fma_f32 ... op_sel_hi).  This kernel is that shape alone: 32 lanes per "chunk" write a
// known row into LDS, read it back as float4 (ds_read_b128) at pos + t and run sig_conv2's multiply-adds packed (or as v_fma_f32 pairs with -DSCALAR);
// the same sums are recomputed from registers only (the row is a function of its index) and compared bit for bit.
//   bin/pk_lds_repro <millis> [partner: 0 none, 1 bf16-MFMA waves in the same block]     (or run  bin/neighbour mfma16  beside it)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int KW = 5, L = 100, P1 = L - KW + 1;

__device__ __forceinline__ float row(unsigned chunk, int s, unsigned it) { return 0.001f * (float)((chunk * 131u + s * 17u + it * 7u) & 1023u) - 0.5f; }
__device__ __forceinline__ float sfma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float sadd(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int PARTNER>
__global__ __launch_bounds__(512) void repro(unsigned long long *bad, unsigned long long *bad_upper, int iters, unsigned seed, float *sink, const float *wsrc) {
    __shared__ __attribute__((aligned(16))) float smem4[8 * (P1 + 4) * 4];
    if (PARTNER && threadIdx.x >= 256) {  // waves 4-7: one per SIMD, bf16 MFMAs back to back
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * threadIdx.x + i); b[i] = (__bf16)(0.5f - 0.001f * threadIdx.x); }
        for (int r = 0; r < iters * 40; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
        return;
    }
    const int tid = threadIdx.x, c = tid >> 5, sub = tid & 31, quad = sub & 3;
    float *s_row = smem4 + c * (P1 + 4) * 4;  // [pos][4 channels], as sig1 in front_sig_kernel
    f32x2 w[KW][4][2];  // sig_conv2's slice of this lane: [tap][ic] -> 4 output channels as two pairs, in VGPRs
    for (int t = 0; t < KW; ++t)
        for (int ic = 0; ic < 4; ++ic)
            for (int o = 0; o < 2; ++o) w[t][ic][o] = f32x2{wsrc[(t * 4 + ic) * 16 + 4 * quad + 2 * o], wsrc[(t * 4 + ic) * 16 + 4 * quad + 2 * o + 1]};
    unsigned long long nbad = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned chunk = (blockIdx.x * 8u + c) * 1000003u + seed, u = it;
        wave_sync();
        for (int i = sub; i < P1 * 4; i += 32) s_row[i] = row(chunk, i, u);
        wave_sync();
        for (int i = sub; i < (P1 - KW + 1) * 4; i += 32) {  // i & 3 == quad
            const int pos = i >> 2;
            f32x2 lo = {0.1f, 0.2f}, hi = {0.3f, 0.4f};
            float r0 = 0.1f, r1 = 0.2f, r2 = 0.3f, r3 = 0.4f;
#pragma unroll
            for (int t = 0; t < KW; ++t) {
                const float4 xv = *reinterpret_cast<const float4 *>(s_row + (pos + t) * 4);  // ds_read_b128 feeding the packed FMAs
#ifdef NOPS  // the read waited for in full, then NOPS idle cycles before anything uses it
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop %0" ::"n"(NOPS) : "memory");
#endif
                float x4[4] = {xv.x, xv.y, xv.z, xv.w};
#ifdef PAIR_ADD  // v_pk_add_f32 on the ALIGNED register pairs of the LDS read (no op_sel): what seq2_front_kernel ships
                lo = lo + f32x2{xv.x, xv.y};
                hi = hi + f32x2{xv.z, xv.w};
                r0 = sadd(r0, row(chunk, (pos + t) * 4 + 0, u)); r1 = sadd(r1, row(chunk, (pos + t) * 4 + 1, u));
                r2 = sadd(r2, row(chunk, (pos + t) * 4 + 2, u)); r3 = sadd(r3, row(chunk, (pos + t) * 4 + 3, u));
                continue;
#endif
#pragma unroll
                for (int ic = 0; ic < 4; ++ic) {
#ifdef FROM_REG  // the operand from registers (the value the row holds), not from the LDS read
                    x4[ic] = row(chunk, (pos + t) * 4 + ic, u);
                    asm volatile("" : "+v"(x4[ic]));
#endif
                    const float x = x4[ic];
#ifdef SPLAT_MOV  // both halves of the broadcast operand as registers of their own: no op_sel on the packed FMA
                    f32x2 xs;
                    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %2" : "=&v"(xs.x), "=&v"(xs.y) : "v"(x));
                    lo = __builtin_elementwise_fma(w[t][ic][0], xs, lo);
                    hi = __builtin_elementwise_fma(w[t][ic][1], xs, hi);
#elif defined(SCALAR)
                    lo = f32x2{sfma(w[t][ic][0].x, x, lo.x), sfma(w[t][ic][0].y, x, lo.y)};
                    hi = f32x2{sfma(w[t][ic][1].x, x, hi.x), sfma(w[t][ic][1].y, x, hi.y)};
#else
                    lo = __builtin_elementwise_fma(w[t][ic][0], f32x2{x, x}, lo);
                    hi = __builtin_elementwise_fma(w[t][ic][1], f32x2{x, x}, hi);
#endif
                    const float xr = row(chunk, (pos + t) * 4 + ic, u);  // the same value from registers only
                    r0 = sfma(w[t][ic][0].x, xr, r0); r1 = sfma(w[t][ic][0].y, xr, r1);
                    r2 = sfma(w[t][ic][1].x, xr, r2); r3 = sfma(w[t][ic][1].y, xr, r3);
                }
            }
            nbad += (__float_as_uint(lo.x) != __float_as_uint(r0)) | (__float_as_uint(lo.y) != __float_as_uint(r1)) |
                    (__float_as_uint(hi.x) != __float_as_uint(r2)) | (__float_as_uint(hi.y) != __float_as_uint(r3));
        }
    }
    if (nbad) { atomicAdd(bad, nbad); if ((tid & 63) >= 32) atomicAdd(bad_upper, nbad); }
}

int main(int argc, char **argv) {
    const int millis = argc > 1 ? atoi(argv[1]) : 5000, partner = argc > 2 ? atoi(argv[2]) : 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int grid = prop.multiProcessorCount * 4, iters = 200;
    unsigned long long *d, h[2];
    (void)hipMalloc(reinterpret_cast<void **>(&d), 64); (void)hipMemset(d, 0, 64);
    float hw[KW * 4 * 16], *dw;
    for (int i = 0; i < KW * 4 * 16; ++i) hw[i] = 0.31f - 0.007f * (float)(i % 53) + 0.0003f * (float)i;
    (void)hipMalloc(reinterpret_cast<void **>(&dw), sizeof(hw)); (void)hipMemcpy(dw, hw, sizeof(hw), hipMemcpyHostToDevice);
    long launches = 0, bad_launches = 0; unsigned long long last = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() < millis) {
        if (partner) hipLaunchKernelGGL(repro<1>, dim3(grid), dim3(512), 0, 0, d, d + 1, iters, (unsigned)launches, reinterpret_cast<float *>(d + 4), dw);
        else hipLaunchKernelGGL(repro<0>, dim3(grid), dim3(256), 0, 0, d, d + 1, iters, (unsigned)launches, reinterpret_cast<float *>(d + 4), dw);
        if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "launch failed\n"); return 1; }
        (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        bad_launches += h[0] != last; last = h[0]; ++launches;
    }
#if defined(SCALAR)
    const char *form = "v_fma_f32 pairs";
#elif defined(SPLAT_MOV)
    const char *form = "v_pk_fma_f32 without op_sel (operand halves in registers of their own)";
#elif defined(FROM_REG)
    const char *form = "v_pk_fma_f32 op_sel, operand from registers instead of the LDS read";
#elif defined(NOPS)
    const char *form = "v_pk_fma_f32 op_sel, s_waitcnt + s_nop behind the LDS read";
#elif defined(PAIR_ADD)
    const char *form = "v_pk_add_f32 on the aligned pairs of the LDS read (no op_sel)";
#else
    const char *form = "v_pk_fma_f32";
#endif
    printf("%s fed from LDS, partner %d: %ld launches (%ld with a wrong sum), %.3g position sums checked, %llu wrong (%llu of them in lanes 32-63)\n",
           form, partner, launches, bad_launches, (double)launches * grid * 8 * (P1 - KW + 1) * 4 * iters, h[0], h[1]);
    return 0;
}
