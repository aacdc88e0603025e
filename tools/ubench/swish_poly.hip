// Micro-benchmark (measurement tool, not product code; round-5 review item 4): what a TRANSCENDENTAL-FREE swish would cost on
// gfx950 next to the instruction mix of the 16-bit kernels' stages - per f32x4 accumulator (4 activations -> 4 bf16), in a
// loop that also issues two v_mfma_f32_16x16x32_bf16 and two ds_read_b128 per accumulator (stage S3 of k_fused.hip issues
// 96 MFMAs for 56 accumulators), two waves per SIMD as in that kernel.
//   MODE 0  today's swish_pack (k_fused.hip:117-138): 4 v_exp_f32, 2 v_pk_add_f32, 4 v_rcp_f32, 2 v_pk_mul_f32, 2 packs
//   MODE 1  clamp to [-8, 8] + odd minimax polynomial of degree 15 for sigmoid - 1/2 (max error 1.76e-3: bf16's 2^-9 grade), all
//           arithmetic as v_pk_fma_f32 / v_pk_mul_f32 on register pairs
//   MODE 2  clamp to [-6, 6] + degree 11 (max error 1.99e-3 inside; 6 * sigmoid(-6) = 1.5e-2 outside: below the grade, the cheapest
//           form anyone could argue for)
//   MODE 3  no activation at all (the loop's floor: MFMAs, reads, the pack)
// Build / run: hipcc --offload-arch=gfx950 -O3 -o /tmp/swish_poly tools/ubench/swish_poly.hip && /tmp/swish_poly
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int DEG>
__device__ __forceinline__ f32x2 sig_poly(f32x2 z) {
    // sigmoid(z) ~ 1/2 + z Q(z^2) on the clamped range; coefficients: numpy Chebyshev least squares on Chebyshev nodes
    const f32x2 t = z * z;
    f32x2 q;
    if (DEG == 15) {
        q = f32x2{-1.290117938e-12f, -1.290117938e-12f};
        q = __builtin_elementwise_fma(q, t, f32x2{3.334165129e-10f, 3.334165129e-10f});
        q = __builtin_elementwise_fma(q, t, f32x2{-3.549187242e-08f, -3.549187242e-08f});
        q = __builtin_elementwise_fma(q, t, f32x2{2.021553350e-06f, 2.021553350e-06f});
        q = __builtin_elementwise_fma(q, t, f32x2{-6.753164914e-05f, -6.753164914e-05f});
        q = __builtin_elementwise_fma(q, t, f32x2{1.393447228e-03f, 1.393447228e-03f});
        q = __builtin_elementwise_fma(q, t, f32x2{-1.932566757e-02f, -1.932566757e-02f});
        q = __builtin_elementwise_fma(q, t, f32x2{2.493988919e-01f, 2.493988919e-01f});
    } else {
        q = f32x2{-1.049191026e-08f, -1.049191026e-08f};
        q = __builtin_elementwise_fma(q, t, f32x2{1.202210101e-06f, 1.202210101e-06f});
        q = __builtin_elementwise_fma(q, t, f32x2{-5.466715762e-05f, -5.466715762e-05f});
        q = __builtin_elementwise_fma(q, t, f32x2{1.300728705e-03f, 1.300728705e-03f});
        q = __builtin_elementwise_fma(q, t, f32x2{-1.907594523e-02f, -1.907594523e-02f});
        q = __builtin_elementwise_fma(q, t, f32x2{2.492840080e-01f, 2.492840080e-01f});
    }
    return __builtin_elementwise_fma(q, z, f32x2{0.5f, 0.5f});
}

template <int MODE>
__device__ __forceinline__ uint2 act_pack(const f32x4 acc) {
    f32x2 lo = {acc[0], acc[1]}, hi = {acc[2], acc[3]};
    if (MODE == 0) {
        const f32x2 elo = {__builtin_amdgcn_exp2f(-lo.x), __builtin_amdgcn_exp2f(-lo.y)};
        const f32x2 ehi = {__builtin_amdgcn_exp2f(-hi.x), __builtin_amdgcn_exp2f(-hi.y)};
        const f32x2 dlo = elo + 1.0f, dhi = ehi + 1.0f;
        const f32x2 rlo = {__builtin_amdgcn_rcpf(dlo.x), __builtin_amdgcn_rcpf(dlo.y)}, rhi = {__builtin_amdgcn_rcpf(dhi.x), __builtin_amdgcn_rcpf(dhi.y)};
        lo = lo * rlo;
        hi = hi * rhi;
    } else if (MODE == 1 || MODE == 2) {
        constexpr float R = MODE == 1 ? 8.0f : 6.0f;
        const f32x2 clo = {__builtin_amdgcn_fmed3f(lo.x, -R, R), __builtin_amdgcn_fmed3f(lo.y, -R, R)};
        const f32x2 chi = {__builtin_amdgcn_fmed3f(hi.x, -R, R), __builtin_amdgcn_fmed3f(hi.y, -R, R)};
        lo = lo * sig_poly<MODE == 1 ? 15 : 11>(clo);
        hi = hi * sig_poly<MODE == 1 ? 15 : 11>(chi);
    }
    const bf16x4 o = {(__bf16)lo.x, (__bf16)lo.y, (__bf16)hi.x, (__bf16)hi.y};
    return __builtin_bit_cast(uint2, o);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(uint2 *out, unsigned long long *cyc, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) uint4 lds[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 1024; i += 256) lds[i] = make_uint4(0x3F803F80u, 0x3F003F00u, 0x3E803F80u, 0x3F803E00u);
    __syncthreads();
    uint4 A = make_uint4(0x3C003C00u + tid, 0x3C803C00u, 0x3C003D00u, 0x3B003C00u);
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed + i, seed * 0.5f, -seed, 0.25f * i};
    uint2 sink = make_uint2(0, 0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // four accumulators in flight, as the stage's tile pairs
            const uint4 b0 = lds[(lane + 64 * g + it) & 1023], b1 = lds[(lane * 5 + 17 * g + it) & 1023];
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, b0), acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, b1), acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint2 o = act_pack<MODE>(acc[g]);
            sink.x ^= o.x;
            sink.y += o.y;
            acc[g] = f32x4{acc[g][0] * 0.001f + seed, acc[g][1] * 0.001f - seed, acc[g][2] * 0.001f + 0.5f, acc[g][3] * 0.001f - 0.25f};  // keep values O(1)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + tid] = sink;
    if (lane == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int MODE>
double run(const char *name, uint2 *d, unsigned long long *dc, int cus, int iters) {
    const int blocks = cus * 2;  // two blocks of four waves per CU: two waves per SIMD
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dc, iters, 0.37f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double per = sum / h.size() / (4.0 * iters);
    printf("  %-58s %8.1f cycles per accumulator (4 activations) per wave, 2 waves per SIMD\n", name, per);
    return per;
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint2 *d;
    unsigned long long *dc;
    (void)hipMalloc(&d, sizeof(uint2) * 256 * cus * 2);
    (void)hipMalloc(&dc, 8 * 4 * cus * 2);
    printf("%s, %d CUs; per accumulator: 2 x v_mfma_f32_16x16x32_bf16 + 2 x ds_read_b128 + the activation + pack to bf16\n", p.gcnArchName, cus);
    const int iters = 20000;
    const double floor_ = run<3>("no activation (floor of the loop)", d, dc, cus, iters);
    const double today = run<0>("swish today: 4 v_exp_f32 + 4 v_rcp_f32 + 4 packed ops", d, dc, cus, iters);
    const double p15 = run<1>("polynomial, degree 15 on [-8, 8] (1.8e-3: bf16 grade)", d, dc, cus, iters);
    const double p11 = run<2>("polynomial, degree 11 on [-6, 6] (2.0e-3 inside only)", d, dc, cus, iters);
    printf("swish alone (loop - floor): today %.1f, degree 15 %.1f (%.0f %% of today), degree 11 %.1f (%.0f %% of today)\n", today - floor_,
           p15 - floor_, 100.0 * (p15 - floor_) / (today - floor_), p11 - floor_, 100.0 * (p11 - floor_) / (today - floor_));
    printf("gate of the review: go only at <= 60 %% of today's cycles\n");
    // accuracy of the two polynomials as evaluated in fp32 (host twin of sig_poly)
    for (int deg : {15, 11}) {
        const double R = deg == 15 ? 8.0 : 6.0;
        const double c15[8] = {2.493988919e-01, -1.932566757e-02, 1.393447228e-03, -6.753164914e-05, 2.021553350e-06, -3.549187242e-08, 3.334165129e-10, -1.290117938e-12};
        const double c11[6] = {2.492840080e-01, -1.907594523e-02, 1.300728705e-03, -5.466715762e-05, 1.202210101e-06, -1.049191026e-08};
        double worst = 0, worst_swish = 0;
        for (int i = -200000; i <= 200000; ++i) {
            const double z = i * 1e-4, zc = fmin(fmax(z, -R), R), t = zc * zc;
            double q = 0;
            if (deg == 15) for (int j = 7; j >= 0; --j) q = q * t + c15[j];
            else for (int j = 5; j >= 0; --j) q = q * t + c11[j];
            const double s = 0.5 + zc * q, ref = 1.0 / (1.0 + exp(-z));
            worst = fmax(worst, fabs(s - ref));
            worst_swish = fmax(worst_swish, fabs(z * s - z * ref));
        }
        printf("degree %d: max |sigmoid error| %.2e, max |swish error| %.2e over z in [-20, 20]\n", deg, worst, worst_swish);
    }
    return 0;
}
