// k_lstm_bf16s.hip — lstm1 + lstm2 + fc head of ConvLSTM_w_ref with the recurrent and input
// GEMMs on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16) using SPLIT operands:
//   every fp32 operand x is written as x = p0 + p1 (+ p2) with bf16 parts (exact for three
//   parts: 3 x 8 significand bits = fp32's 24), and a product a*b is accumulated in fp32 from
//   the part products  NP=3: a0b0 + a0b1 + a1b0 + a0b2 + a2b0 + a1b1   (error ~2^-24, fp32 class)
//                      NP=2: a0b0 + a0b1 + a1b0                         (error ~2^-16)
//                      NP=1: a0b0 (plain bf16 inputs, fp32 accumulate)
// Replaces the same reference lines as k_lstm.hip (models/ConvLSTM_w_ref.py:51-56); same
// wave/gate ownership (wave w = hidden units 16w..16w+15 of all four gates, cell state in
// registers).  Why: the fp32 MFMA shares the SIMD's fp32 datapath with VALU work (measured:
// gate transcendentals cost their full issue time), whereas the bf16 MFMA is a separate pipe
// that is 16x faster per instruction-K, so 6 part products still cost 2.7x fewer matrix cycles
// than the fp32 instruction and the gate math overlaps with them.
//
// LDS images: per part, 4 planes (plane q = channels {32ks+8q..+7}), rows of H/32 16-byte
// slots padded to an odd slot count -> conflict-free ds_read_b128 B fragments.
#include "rmr_internal.h"
#include "rmr_math.h"

namespace rmr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// split x into NP bf16 parts, returned as fp32 bit patterns whose low 16 bits are zero.  F16 (dtype f16x3, NP = 2): two IEEE
// half parts instead - hi = half(x), lo = half(x - hi), each in the high 16 bits of its word like the bf16 parts: 22
// significand bits from three products (hi hi, hi lo, lo hi), where two bf16 parts carry 16
template <int NP, bool F16 = false>
__device__ __forceinline__ void split_parts(float x, unsigned (&p)[NP]) {
    if constexpr (F16) {
        static_assert(NP == 2, "the half split has two parts");
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        p[0] = (unsigned)__builtin_bit_cast(unsigned short, hi) << 16;
        p[1] = (unsigned)__builtin_bit_cast(unsigned short, lo) << 16;
    } else if (NP == 1) {  // round to nearest even
        const unsigned b = __float_as_uint(x);
        p[0] = (b + 0x7fffu + ((b >> 16) & 1u)) & 0xffff0000u;
    } else {
        float r = x;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const unsigned b = __float_as_uint(r);
            if (i + 1 < NP || NP == 3) {
                p[i] = b & 0xffff0000u;  // truncation: exact remainder chain
            } else {
                p[i] = (b + 0x7fffu + ((b >> 16) & 1u)) & 0xffff0000u;  // last of two: round
            }
            r -= __uint_as_float(p[i]);
        }
    }
}
__device__ __forceinline__ unsigned pack2(unsigned lo_elem, unsigned hi_elem) {
    return (lo_elem >> 16) | hi_elem;  // two bf16 (given as fp32 patterns) -> one dword
}

// part-product schedule: pairs (a part, b part)
template <int NP> struct Prod;
template <> struct Prod<1> { static constexpr int N = 1; static constexpr int A[1] = {0}; static constexpr int B[1] = {0}; };
template <> struct Prod<2> { static constexpr int N = 3; static constexpr int A[3] = {0, 0, 1}; static constexpr int B[3] = {0, 1, 0}; };
template <> struct Prod<3> { static constexpr int N = 6; static constexpr int A[6] = {0, 0, 1, 0, 2, 1}; static constexpr int B[6] = {0, 1, 0, 2, 0, 1}; };

template <bool F16>
__device__ __forceinline__ f32x4 mma_part(const uint4 a, const bf16x8 b, const f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), b, c, 0, 0, 0);
}

// acc[gt] += sum over k-steps and part products of A[gt][ks][pa] * B(img)[ks][pb]
template <int KS32, int NP, int SL, bool F16>
__device__ __forceinline__ void mm_split_impl(const uint4 (&img)[NP][4][16][SL], int q, int nn,
                                              const uint4 (&A)[4][KS32][NP], f32x4 (&acc)[4]) {
    using P = Prod<NP>;
#pragma unroll
    for (int ks = 0; ks < KS32; ++ks) {
        bf16x8 b[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) b[p] = __builtin_bit_cast(bf16x8, img[p][q][nn][ks]);
#pragma unroll
        for (int pr = 0; pr < P::N; ++pr)
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
                acc[gt] = mma_part<F16>(A[gt][ks][P::A[pr]], b[P::B[pr]], acc[gt]);
    }
}
template <int KS32, int NP, int SL, bool F16>
__device__ __forceinline__ void mm_split(const uint4 (&img)[NP][4][16][SL], int q, int nn,
                                         const uint4 (&A)[4][KS32][NP], f32x4 (&acc)[4]) {
    mm_split_impl<KS32, NP, SL, F16>(img, q, nn, A, acc);
}

struct LstmSArgs {
    const float *x;       // [n][T][H] fp32, channel-last
    float *logits;        // [n][num_out]
    const uint4 *a_ih, *a_hh;  // [H/16 waves][4 gates][H/32 ks][NP][64 lanes] bf16x8 fragments (pre-scaled)
    const float *b1;           // [4H] pre-scaled b_ih + b_hh
    const float *a_ih2, *b2, *w_fc, *b_fc;  // lstm2 / fc exactly as in k_lstm.hip (fp32 MFMA, once per group)
    int64_t n;
    int T, num_out;
};

template <int H, int NP, bool F16>
__global__ __launch_bounds__(4 * H) void lstm_bf16s_kernel(LstmSArgs a) {
    constexpr int NW = H / 16;
    constexpr int KS32 = H / 32;                 // bf16 k-steps of 32 channels
    constexpr int SL = (KS32 % 2 == 0) ? KS32 + 1 : KS32;  // 16-byte slots per row per plane (odd)
    constexpr int KS = H / 4, G = H / 16;        // fp32 path constants for lstm2
    constexpr int RS = (G % 2 == 0) ? H / 4 + 4 : H / 4;
    static_assert(H % 32 == 0, "split-bf16 LSTM needs H multiple of 32");
    using P = Prod<NP>;
    // bf16 part images: [buf][part][plane q][row n][slot] (uint4 = 8 bf16)
    __shared__ uint4 xs[2][NP][4][16][SL];
    __shared__ uint4 hs[2][NP][4][16][SL];
    __shared__ __attribute__((aligned(16))) float hlast[4][16][RS];  // fp32 h_{T-1} for lstm2
    __shared__ float part[NW][16][16];

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15;

    uint4 Aih[4][KS32][NP], Ahh[4][KS32][NP];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const size_t idx = ((((size_t)w * 4 + gt) * KS32 + ks) * NP + p) * 64 + lane;
                Aih[gt][ks][p] = a.a_ih[idx];
                Ahh[gt][ks][p] = a.a_hh[idx];
            }
    f32x4 bias[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) bias[gt] = *reinterpret_cast<const f32x4 *>(a.b1 + gt * H + 16 * w + 4 * q);

    // staging role: chunk row = tid / (H/4), 4-channel piece c4 = tid % (H/4)
    const int st_row = tid / (H / 4), st_c4 = tid - st_row * (H / 4);
    const int st_grp = st_c4 >> 1;                 // 8-channel group
    const int st_q = st_grp & 3, st_ks = st_grp >> 2, st_half = st_c4 & 1;

    auto stage_x = [&](int buf, const float4 v) {
        unsigned e[4][NP];
        split_parts<NP, F16>(v.x, e[0]); split_parts<NP, F16>(v.y, e[1]);
        split_parts<NP, F16>(v.z, e[2]); split_parts<NP, F16>(v.w, e[3]);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            uint2 wv = make_uint2(pack2(e[0][p], e[1][p]), pack2(e[2][p], e[3][p]));
            *(reinterpret_cast<uint2 *>(&xs[buf][p][st_q][st_row][st_ks]) + st_half) = wv;
        }
    };

    const int64_t n_groups = (a.n + 15) / 16;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t chunk0 = grp * 16;
        int64_t st_chunk = chunk0 + st_row;
        if (st_chunk >= a.n) st_chunk = a.n - 1;
        const float4 *xsrc = reinterpret_cast<const float4 *>(a.x + (size_t)st_chunk * a.T * H) + st_c4;
        RMR_SYNC();
        stage_x(0, xsrc[0]);
        stage_x(1, xsrc[(size_t)(a.T > 1 ? 1 : 0) * (H / 4)]);
        RMR_SYNC();

        // Software pipeline as in k_lstm.hip: accN = b + W_ih x_{t+1} is issued in slices between
        // the gate-math slices of step t (bf16 MFMA and VALU are separate pipes, but a wave issues
        // in order: without the interleave the matrix pipe idles while the wave does its VALU work).
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        f32x4 accN[4] = {bias[0], bias[1], bias[2], bias[3]};
        mm_split<KS32, NP, SL, F16>(xs[0], q, nn, Aih, accN);
        RMR_SYNC();  // xs[0] is overwritten at the end of step 0: all x_0 reads first (see k_lstm_x16.hip)
        for (int t = 0; t < a.T; ++t) {
            const int tf = (t + 2 < a.T) ? t + 2 : a.T - 1;
            const float4 xnext = xsrc[(size_t)tf * (H / 4)];
            f32x4 acc[4] = {accN[0], accN[1], accN[2], accN[3]};
            bf16x8 bxn[KS32][NP];
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks)
#pragma unroll
                for (int p = 0; p < NP; ++p) bxn[ks][p] = __builtin_bit_cast(bf16x8, xs[(t + 1) & 1][p][q][nn][ks]);
            if (t > 0) mm_split<KS32, NP, SL, F16>(hs[(t - 1) & 1], q, nn, Ahh, acc);  // recurrent critical path
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) accN[gt] = bias[gt];
            f32x4 h;
            float ig[4], fg[4], gg[4];
            constexpr int NM = KS32 * P::N * 4;  // projection MFMAs of the next step
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                // (in the last step this projects a stale, finite tile and the result is dropped:
                //  keeps the step body branch-free so the slices stay in one scheduling region)
#pragma unroll
                for (int mi = s * NM / 16; mi < (s + 1) * NM / 16; ++mi) {
                    const int gt = mi & 3, pr = (mi >> 2) % P::N, ks = (mi >> 2) / P::N;
                    accN[gt] = mma_part<F16>(Aih[gt][ks][P::A[pr]], bxn[ks][P::B[pr]], accN[gt]);
                }
                const int r = s >> 2, st = s & 3;
                if (st == 0) ig[r] = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[0][r]));
                if (st == 1) fg[r] = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[1][r]));
                if (st == 2) {
                    gg[r] = fmaf(-2.0f, fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[2][r])), 1.0f);
                    c[r] = fmaf(fg[r], c[r], ig[r] * gg[r]);
                }
                if (st == 3) {
                    const float og = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[3][r]));
                    h[r] = og * tanh_f(c[r]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // h parts -> LDS: this lane's 4 units are channels 16w+4q..+3 of chunk nn
            {
                unsigned e[4][NP];
#pragma unroll
                for (int r = 0; r < 4; ++r) split_parts<NP, F16>(h[r], e[r]);
                const int grp8 = 2 * w + (q >> 1);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    uint2 wv = make_uint2(pack2(e[0][p], e[1][p]), pack2(e[2][p], e[3][p]));
                    *(reinterpret_cast<uint2 *>(&hs[t & 1][p][grp8 & 3][nn][grp8 >> 2]) + (q & 1)) = wv;
                }
            }
            if (t + 1 == a.T) *reinterpret_cast<f32x4 *>(&hlast[q][nn][4 * w]) = h;
            stage_x(t & 1, xnext);  // x_{t+2} into the buffer whose last reader was step t-1
            RMR_SYNC();
        }

        // ---- lstm2: one step on swish(h1[T-1]), fp32 MFMA (48 instructions per group) ----
        f32x4 acc2[3];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) acc2[gt] = *reinterpret_cast<const f32x4 *>(a.b2 + gt * H + 16 * w + 4 * q);
        {
            const float *hb = &hlast[q][nn][0];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                f32x4 z = *reinterpret_cast<const f32x4 *>(hb + 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = swish_f(z[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) {
                        const float aw = a.a_ih2[((size_t)(w * 3 + gt) * KS + g * 4 + j) * 64 + lane];
                        acc2[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, z[j], acc2[gt], 0, 0, 0);
                    }
            }
        }
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float c2 = sigmoid_f(acc2[0][r]) * tanh_f(acc2[1][r]);
            const float h2 = sigmoid_f(acc2[2][r]) * tanh_f(c2);
            y[r] = swish_f(h2);
        }
        for (int o = 0; o < a.num_out; ++o) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(a.w_fc + (size_t)o * H + 16 * w + 4 * q);
            float p = wv[0] * y[0] + wv[1] * y[1] + wv[2] * y[2] + wv[3] * y[3];
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
                for (int ww = 0; ww < NW; ++ww) s += part[ww][ch][o];
                a.logits[(size_t)(chunk0 + ch) * a.num_out + o] = s;
            }
        }
    }
}

template <int H, int NP, bool F16 = false>
static int launch_lstm_s_t(rmr_model *m, const float *x, int64_t n, float *logits) {
    rmr_engine *e = m->eng;
    LstmSArgs a;
    a.x = x; a.logits = logits; a.n = n; a.T = m->T; a.num_out = m->desc.num_out;
    a.a_ih = reinterpret_cast<const uint4 *>(m->lstm.s_ih1); a.a_hh = reinterpret_cast<const uint4 *>(m->lstm.s_hh1);
    a.b1 = m->lstm.b1; a.a_ih2 = m->lstm.a_ih2; a.b2 = m->lstm.b2; a.w_fc = m->lstm.w_fc; a.b_fc = m->lstm.b_fc;
    const int64_t groups = (n + 15) / 16;
    int64_t grid = (int64_t)e->num_cus * 2;
    if (grid > groups) grid = groups;
    if (grid < 1) return 0;
    ProfScope ps(e, K_LSTM_HEAD);
    hipLaunchKernelGGL((lstm_bf16s_kernel<H, NP, F16>), dim3((unsigned)grid), dim3(4 * H), 0, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

int launch_lstm_head_split(rmr_model *m, const float *x, int64_t n, float *logits) {
    const int np = m->nparts;
    if (m->split_f16) {  // dtype f16x3: two half parts
        if (m->desc.size == 64) return launch_lstm_s_t<64, 2, true>(m, x, n, logits);
        if (m->desc.size == 32) return launch_lstm_s_t<32, 2, true>(m, x, n, logits);
    }
    if (m->desc.size == 64) {
        if (np == 1) return launch_lstm_s_t<64, 1>(m, x, n, logits);
        if (np == 2) return launch_lstm_s_t<64, 2>(m, x, n, logits);
        if (np == 3) return launch_lstm_s_t<64, 3>(m, x, n, logits);
    } else if (m->desc.size == 32) {
        if (np == 1) return launch_lstm_s_t<32, 1>(m, x, n, logits);
        if (np == 2) return launch_lstm_s_t<32, 2>(m, x, n, logits);
        if (np == 3) return launch_lstm_s_t<32, 3>(m, x, n, logits);
    }
    RMR_FAIL(RMR_ERR_INVALID, "split-bf16 LSTM supports size 32/64 (got %d, parts %d)", m->desc.size, np);
}

}  // namespace rmr
