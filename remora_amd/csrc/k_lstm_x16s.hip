// k_lstm_x16s.hip — lstm1 (T steps) + lstm2 (ONE step) + fc head of ConvLSTM_w_ref (size 64) on the 16-bit matrix cores
// with SPLIT operands (dtypes f16x3, bf16x3, bf16x6), in the shape of k_lstm_x16.hip: block = 8 waves x 16 chunks, wave w
// owns hidden units 8w..8w+7 as two unit-major MFMA tiles, so the four gates of a unit of a chunk meet in one lane and the
// cell update is lane-local.  Replaces models/ConvLSTM_w_ref.py:51-56 like its siblings.
//
// Why it exists (round 4): lstm_bf16s_kernel (k_lstm_bf16s.hip) keeps the 16 hidden units of a wave as four 16-row tiles
// with every part of W_ih and W_hh resident - 188 to 376 registers, ONE wave per SIMD.  Here a wave holds 8 units: 32 VGPRs
// of fragments per part; with two parts the kernel is held to 128 VGPRs (22 spilled) so that two blocks = four waves per
// SIMD are resident.  Measured (524 k chunks, ns per chunk): f16x3 7.27 -> 6.15 at C100, bf16x6 8.87 -> 8.40 (192 VGPRs,
// one block per CU); the size-32 models keep lstm_bf16s_kernel.
//
// x arrives as fp32 [n][T][64] (the split convolutions write fp32); it is split into NP 16-bit parts while it is staged,
// h when it is written: parts as in k_lstm_bf16s.hip (bf16: truncation chain, the last of two rounded; F16: hi = half(x),
// lo = half(x - hi)), products per MFMA site: NP = 2 -> (hi hi, hi lo, lo hi), NP = 3 -> six.  Gate rows of W and b are
// pre-scaled on the host (i, f, o by -log2 e, g by 2 log2 e) before they are split.  fp32 accumulation and cell state.
#include <type_traits>

#include "rmr_internal.h"
#include "rmr_math.h"

namespace rmr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

struct LstmXsArgs {
    const float *x;        // fp32 [n][T][64]
    float *logits;         // [n][num_out]
    const uint4 *a_ih, *a_hh, *a_ih2;  // 16-bit A fragments [8 waves][2 tiles][2 k-steps][NP][64 lanes]
    const float *b1, *b2;  // [8 waves][2 tiles][4 q][4 gates] pre-scaled b_ih + b_hh (lstm2: the f row is unused)
    const float *w_fc, *b_fc;
    int64_t n;
    int T, num_out;
};

template <int NP> struct ProdX;
template <> struct ProdX<2> { static constexpr int N = 3; static constexpr int A[3] = {0, 0, 1}; static constexpr int B[3] = {0, 1, 0}; };
template <> struct ProdX<3> { static constexpr int N = 6; static constexpr int A[6] = {0, 0, 1, 0, 2, 1}; static constexpr int B[6] = {0, 1, 0, 2, 0, 1}; };

// x -> NP parts, each in the HIGH 16 bits of its word (split_parts of k_lstm_bf16s.hip)
template <int NP, bool F16>
__device__ __forceinline__ void split16(float x, unsigned (&p)[NP]) {
    if constexpr (F16) {
        static_assert(NP == 2, "the half split has two parts");
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        p[0] = (unsigned)__builtin_bit_cast(unsigned short, hi) << 16;
        p[1] = (unsigned)__builtin_bit_cast(unsigned short, lo) << 16;
    } else {
        float r = x;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const unsigned b = __float_as_uint(r);
            p[i] = (i + 1 < NP || NP == 3) ? (b & 0xffff0000u) : ((b + 0x7fffu + ((b >> 16) & 1u)) & 0xffff0000u);
            r -= __uint_as_float(p[i]);
        }
    }
}

template <bool F16>
__device__ __forceinline__ f32x4 mma16(const uint4 a, const uint4 b, const f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// acc rows are pre-scaled: [0] i, [1] f, [3] o by -log2(e); [2] g by 2 log2(e); the lane's two units at once
__device__ __forceinline__ f32x2 exp2_2(const f32x2 v) { return f32x2{__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)}; }
__device__ __forceinline__ f32x2 rcp_2(const f32x2 v) { return f32x2{fast_rcp(v.x), fast_rcp(v.y)}; }
__device__ __forceinline__ f32x2 cell2(const f32x4 acc0, const f32x4 acc1, float &c0, float &c1) {
    const f32x2 ig = rcp_2(exp2_2(f32x2{acc0[0], acc1[0]}) + 1.0f);
    const f32x2 fg = rcp_2(exp2_2(f32x2{acc0[1], acc1[1]}) + 1.0f);
    const f32x2 gr = rcp_2(exp2_2(f32x2{acc0[2], acc1[2]}) + 1.0f);
    const f32x2 og = rcp_2(exp2_2(f32x2{acc0[3], acc1[3]}) + 1.0f);
    const f32x2 gg = __builtin_elementwise_fma(f32x2{-2.0f, -2.0f}, gr, f32x2{1.0f, 1.0f});
    const f32x2 c = __builtin_elementwise_fma(fg, f32x2{c0, c1}, ig * gg);
    c0 = c.x;
    c1 = c.y;
    const f32x2 tr = rcp_2(exp2_2(c * 2.8853900817779268f) + 1.0f);
    const f32x2 tc = __builtin_elementwise_fma(f32x2{-2.0f, -2.0f}, tr, f32x2{1.0f, 1.0f});
    return og * tc;
}

template <int NP, bool F16>
__global__ __launch_bounds__(512, NP == 2 ? 4 : 2) void lstm_x16s_kernel(LstmXsArgs a) {
    using P = ProdX<NP>;
    // B-operand images per part (8 x 16 bit = 16 B per slot): plane p = 8-channel group (channel / 8) % 4, slot = channel / 32,
    // rows = chunks; 3 slots per row (2 used) keep the 16-lane ds_read_b128 groups on distinct bank slots
    __shared__ uint4 xs[2][NP][4][16][3];
    __shared__ uint4 hs[2][NP][4][16][3];
    __shared__ float part[8][16][16];
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15;

    uint4 Aih[2][2][NP], Ahh[2][2][NP];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const size_t idx = ((((size_t)w * 2 + t) * 2 + ks) * NP + p) * 64 + lane;
                Aih[t][ks][p] = a.a_ih[idx];
                Ahh[t][ks][p] = a.a_hh[idx];
            }
    f32x4 bias[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) bias[t] = *reinterpret_cast<const f32x4 *>(a.b1 + ((w * 2 + t) * 4 + q) * 4);

    // x staging role (threads 0..127): chunk row = tid >> 3, 8-channel group c8 = tid & 7 -> plane c8 & 3, slot c8 >> 2
    const bool stager = tid < 128;
    const int st_row = tid >> 3, st_c8 = tid & 7;
    const int h_plane = w & 3, h_slot = w >> 2;

    auto stage_x = [&](const int buf, const float4 v0, const float4 v1) {
        unsigned e[8][NP];
        split16<NP, F16>(v0.x, e[0]); split16<NP, F16>(v0.y, e[1]); split16<NP, F16>(v0.z, e[2]); split16<NP, F16>(v0.w, e[3]);
        split16<NP, F16>(v1.x, e[4]); split16<NP, F16>(v1.y, e[5]); split16<NP, F16>(v1.z, e[6]); split16<NP, F16>(v1.w, e[7]);
#pragma unroll
        for (int p = 0; p < NP; ++p)
            xs[buf][p][st_c8 & 3][st_row][st_c8 >> 2] = make_uint4((e[0][p] >> 16) | e[1][p], (e[2][p] >> 16) | e[3][p],
                                                                    (e[4][p] >> 16) | e[5][p], (e[6][p] >> 16) | e[7][p]);
    };

    const int64_t n_groups = (a.n + 15) / 16;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t chunk0 = grp * 16;
        int64_t st_chunk = chunk0 + st_row;
        if (st_chunk >= a.n) st_chunk = a.n - 1;  // ragged tail: clamp (results masked)
        const float4 *xsrc = reinterpret_cast<const float4 *>(a.x + (size_t)st_chunk * a.T * 64) + 2 * st_c8;
        RMR_SYNC();  // the previous group's LDS traffic is done
        if (stager) {
            stage_x(0, xsrc[0], xsrc[1]);
            const size_t o1 = (size_t)(a.T > 1 ? 1 : 0) * 16;
            stage_x(1, xsrc[o1], xsrc[o1 + 1]);
        }
        RMR_SYNC();

        float c[2] = {0.f, 0.f};
        f32x4 accN[2];  // bias + W_ih x_t of the step about to run
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            accN[u] = bias[u];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pr = 0; pr < P::N; ++pr) accN[u] = mma16<F16>(Aih[u][ks][P::A[pr]], xs[0][P::B[pr]][q][nn][ks], accN[u]);
        }
        RMR_SYNC();  // x_0 read by every wave before step 0 ends with its tile overwritten (k_lstm_x16.hip)

        auto step = [&](const int t, auto last_c) {
            constexpr bool LAST = decltype(last_c)::value;
            const int tf = (t + 2 < a.T) ? t + 2 : a.T - 1;  // x_{t+2} (the last two fetches are redundant re-reads)
            float4 xn0 = make_float4(0.f, 0.f, 0.f, 0.f), xn1 = xn0;
            if (stager) {
                xn0 = xsrc[(size_t)tf * 16];
                xn1 = xsrc[(size_t)tf * 16 + 1];
            }
            f32x4 acc[2] = {accN[0], accN[1]};
            uint4 bx[2][NP];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int p = 0; p < NP; ++p) bx[ks][p] = xs[(t + 1) & 1][p][q][nn][ks];
            if (t > 0) {  // recurrent critical path: W_hh h_{t-1}
                uint4 bh[2][NP];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int p = 0; p < NP; ++p) bh[ks][p] = hs[(t - 1) & 1][p][q][nn][ks];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int pr = 0; pr < P::N; ++pr)
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[u] = mma16<F16>(Ahh[u][ks][P::A[pr]], bh[ks][P::B[pr]], acc[u]);
            }
            // input projection of the next step (in the last step it projects a stale, finite tile: dropped)
#pragma unroll
            for (int u = 0; u < 2; ++u) accN[u] = bias[u];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pr = 0; pr < P::N; ++pr)
#pragma unroll
                    for (int u = 0; u < 2; ++u) accN[u] = mma16<F16>(Aih[u][ks][P::A[pr]], bx[ks][P::B[pr]], accN[u]);
            const f32x2 hh = cell2(acc[0], acc[1], c[0], c[1]);
            float h0 = hh.x, h1 = hh.y;
            if constexpr (LAST) {  // lstm2 consumes swish(h1[T-1]) (models/ConvLSTM_w_ref.py:52)
                h0 = swish_f(h0);
                h1 = swish_f(h1);
            }
            unsigned e0[NP], e1[NP];
            split16<NP, F16>(h0, e0);
            split16<NP, F16>(h1, e1);
#pragma unroll
            for (int p = 0; p < NP; ++p) reinterpret_cast<unsigned *>(&hs[t & 1][p][h_plane][nn][h_slot])[q] = (e0[p] >> 16) | e1[p];
            if (stager) stage_x(t & 1, xn0, xn1);  // the buffer whose last reader was step t-1
            RMR_SYNC();
        };
        for (int t = 0; t + 1 < a.T; ++t) step(t, std::false_type{});
        step(a.T - 1, std::true_type{});

        // ---- lstm2: one step on swish(h1[T-1]) with zero state (the f gate meets c0 = 0) ----
        f32x4 acc2[2];
        {
            uint4 bh[2][NP];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int p = 0; p < NP; ++p) bh[ks][p] = hs[(a.T - 1) & 1][p][q][nn][ks];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                acc2[u] = *reinterpret_cast<const f32x4 *>(a.b2 + ((w * 2 + u) * 4 + q) * 4);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int pr = 0; pr < P::N; ++pr)
                        acc2[u] = mma16<F16>(a.a_ih2[((((size_t)w * 2 + u) * 2 + ks) * NP + P::A[pr]) * 64 + lane], bh[ks][P::B[pr]], acc2[u]);
            }
        }
        float c2a = 0.f, c2b = 0.f;
        const f32x2 h2 = cell2(acc2[0], acc2[1], c2a, c2b);  // c2 = sig(i) tanh(g); h2 = sig(o) tanh(c2)
        const float y[2] = {swish_f(h2.x), swish_f(h2.y)};
        // ---- fc: this lane's two hidden units, reduced over q (lanes) then over the 8 waves (LDS) ----
        const int u0 = 8 * w + 2 * q;
        for (int o = 0; o < a.num_out; ++o) {
            float p = a.w_fc[(size_t)o * 64 + u0] * y[0] + a.w_fc[(size_t)o * 64 + u0 + 1] * y[1];
            p += __shfl_xor(p, 16);
            p += __shfl_xor(p, 32);
            if (q == 0) part[w][nn][o] = p;
        }
        RMR_SYNC();
        if (tid < 16 * a.num_out) {
            const int ch = tid / a.num_out, o = tid - ch * a.num_out;
            if (chunk0 + ch < a.n) {
                float s = a.b_fc[o];
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) s += part[ww][ch][o];
                a.logits[(size_t)(chunk0 + ch) * a.num_out + o] = s;
            }
        }
    }
}

}  // namespace

bool lstm_x16s_supported(const rmr_model *m) { return m->desc.size == 64 && m->nparts >= 2 && m->lstm.xs_ih != nullptr; }

int launch_lstm_head_x16s(rmr_model *m, const float *x, int64_t n, float *logits) {
    rmr_engine *e = m->eng;
    if (!lstm_x16s_supported(m)) RMR_FAIL(RMR_ERR_INVALID, "split 16-bit LSTM: size 64 with two or three parts only");
    if (n <= 0) return 0;
    LstmXsArgs a;
    a.x = x; a.logits = logits; a.n = n; a.T = m->T; a.num_out = m->desc.num_out;
    a.a_ih = reinterpret_cast<const uint4 *>(m->lstm.xs_ih); a.a_hh = reinterpret_cast<const uint4 *>(m->lstm.xs_hh);
    a.a_ih2 = reinterpret_cast<const uint4 *>(m->lstm.xs_ih2);
    a.b1 = m->lstm.x_b1; a.b2 = m->lstm.x_b2; a.w_fc = m->lstm.w_fc; a.b_fc = m->lstm.b_fc;
    const int64_t groups = (n + 15) / 16;
    int64_t grid = (int64_t)e->num_cus * 8;
    if (grid > groups) grid = groups;
    ProfScope ps(e, K_LSTM_HEAD);
    if (m->split_f16) hipLaunchKernelGGL((lstm_x16s_kernel<2, true>), dim3((unsigned)grid), dim3(512), 0, e->stream, a);
    else if (m->nparts == 2) hipLaunchKernelGGL((lstm_x16s_kernel<2, false>), dim3((unsigned)grid), dim3(512), 0, e->stream, a);
    else hipLaunchKernelGGL((lstm_x16s_kernel<3, false>), dim3((unsigned)grid), dim3(512), 0, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

}  // namespace rmr
