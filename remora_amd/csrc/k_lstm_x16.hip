// k_lstm_x16.hip — lstm1 (T steps) + lstm2 (ONE step) + fc head of ConvLSTM_w_ref (size 64) on the bf16 matrix
// cores, for the bf16 x tensor the fused front kernel writes (k_fused.hip): x bf16[n][T][64] -> logits f32[n][num_out].
// Replaces models/ConvLSTM_w_ref.py:51-56 (same algebra as k_lstm.hip: after the two flips z[-1] is the first output of
// lstm2 on the reversed sequence, i.e. ONE lstm2 step on swish(h1[T-1]) with zero state).
//
// The gate non-linearities (5 exp + 5 rcp + ~12 plain VALU per hidden unit, chunk and step) outweigh the matrix work
// (8 MFMAs per wave and step), and one wave issues at most one VALU instruction per ~4.5 cycles (tools/ubench):
// the kernel is built for FOUR waves per SIMD so that the VALU pipe always has a second wave to issue from.
//   block = 8 waves x 16 chunks; wave w owns hidden units 8w..8w+7 as two 16-row MFMA tiles whose rows are
//   UNIT-MAJOR: row r of tile t = (unit 8w + 2(r>>2) + t, gate r&3).  The D fragment of lane (q, n) is then the four
//   gates i,f,g,o of ONE unit for ONE chunk: the cell update is lane-local, c stays in a register, and the lane's two
//   units (tiles 0/1) are neighbours 8w+2q, 8w+2q+1 -> h leaves as ONE packed 4-byte LDS store.
//   Per wave: 32 VGPRs of A fragments (W_ih, W_hh), ~90 VGPRs in all -> 2 blocks (16 waves) per CU.
// The input projection of step t+1 (W_ih x_{t+1}, independent of h_t) is issued before the gate math of step t, so only
// W_hh h_{t-1} (4 MFMAs) sits on the recurrent critical path.  x_{t+2} is fetched from HBM while step t runs.
// Gate rows of W and b are pre-scaled on the host (i,f,o by -log2 e, g by 2 log2 e): sigmoid(a) = 1/(1+2^a'),
// tanh(a) = 1 - 2/(1+2^a').  bf16 operands, fp32 accumulation, fp32 cell state.
#include <type_traits>

#include "rmr_internal.h"
#include "rmr_math.h"

namespace rmr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct LstmXArgs {
    const uint16_t *x;     // bf16 [n][T][64]
    float *logits;         // [n][num_out]
    const uint4 *a_ih, *a_hh, *a_ih2;  // bf16 A fragments [8 waves][2 tiles][2 k-steps][64 lanes]
    const float *b1, *b2;  // [8 waves][2 tiles][4 q][4 gates] pre-scaled b_ih + b_hh (lstm2: the f row is unused)
    const float *w_fc, *b_fc;  // [num_out][64], [num_out]
    int64_t n;
    int T, num_out;
};

// F16: the operands (x from the fused front kernel, h, the weights) are IEEE half instead of bf16 (k_fused.hip)
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const uint4 a, const uint4 b, const f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// acc rows are pre-scaled: [0] i, [1] f, [3] o by -log2(e); [2] g by 2 log2(e)
__device__ __forceinline__ float lstm_cell(const f32x4 acc, float &c) {
    const float ig = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[0]));
    const float fg = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[1]));
    const float gg = fmaf(-2.0f, fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[2])), 1.0f);
    const float og = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[3]));
    c = fmaf(fg, c, ig * gg);
    const float tc = fmaf(-2.0f, fast_rcp(1.0f + __builtin_amdgcn_exp2f(c * 2.8853900817779268f)), 1.0f);
    return og * tc;
}

// The lane's TWO units at once: everything that is not an exp or a rcp works on register pairs (v_pk_add_f32,
// v_pk_fma_f32, v_pk_mul_f32: 11 packed + 20 transcendental instructions instead of 22 + 20; the VALU is this kernel's
// bound).  Packed and scalar fp32 operations round alike: same bits as two lstm_cell calls.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef RMR_LSTM_PAIRS
#define RMR_LSTM_PAIRS 1
#endif
__device__ __forceinline__ f32x2 exp2_2(const f32x2 v) { return f32x2{__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)}; }
__device__ __forceinline__ f32x2 rcp_2(const f32x2 v) { return f32x2{fast_rcp(v.x), fast_rcp(v.y)}; }
// Tried and dropped (profiles/r03_lstm_shared_rcp_ab.log): sharing reciprocals - sig(i) tanh(g) = (G - 1) / ((1 + A)(1 + G)),
// sig(o) tanh(c) = (C - 1) / ((1 + O)(1 + C)), 5 exp + 3 rcp per unit instead of 5 + 5 - ran 2.42 against 2.37 ns/chunk: the
// step is bound by the recurrent chain (h -> MFMA -> gates -> h), which the extra multiply in front of each rcp lengthens.
__device__ __forceinline__ f32x2 lstm_cell2(const f32x4 acc0, const f32x4 acc1, float &c0, float &c1) {
    if (!RMR_LSTM_PAIRS) return f32x2{lstm_cell(acc0, c0), lstm_cell(acc1, c1)};
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

// Two 16-chunk groups per block iteration (round 4; the one-group kernel it replaced - 2.38 against 2.22 ns per chunk at C100 -
// left the library in round 6): a wave issues
// the MFMAs of both groups before the gate math of the first, so the matrix pipe works on group B while the VALU works on
// group A, every dependent chain (LDS read -> MFMA -> gates -> LDS write) has an independent twin to fill its bubbles, and a
// block barrier is paid once per two groups.  Per chunk the same operations in the same order: bit-identical logits.
// 128 VGPRs (two spilled), still four waves per SIMD.
template <bool F16>
__global__ __launch_bounds__(512, 4) void lstm_x16_g2_kernel(LstmXArgs a) {
    // B-operand images (8 bf16 = 16 B per slot): plane p = 8-channel group (channel / 8) % 4, slot = channel / 32, rows = chunks;
    // 3 slots per row (2 used) keep the 16-lane ds_read_b128 groups on distinct bank slots
    __shared__ uint4 xs[2][2][4][16][3];  // [buffer][group][plane][chunk row][slot]
    __shared__ uint4 hs[2][2][4][16][3];
    __shared__ float part[2][8][16][16];
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15;

    uint4 Aih[2][2], Ahh[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Aih[t][ks] = a.a_ih[((w * 2 + t) * 2 + ks) * 64 + lane];
            Ahh[t][ks] = a.a_hh[((w * 2 + t) * 2 + ks) * 64 + lane];
        }
    f32x4 bias[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) bias[t] = *reinterpret_cast<const f32x4 *>(a.b1 + ((w * 2 + t) * 4 + q) * 4);

    // x staging: threads 0..255, 128 per group (waves 0,1 -> group 0; waves 2,3 -> group 1)
    const bool stager = tid < 256;
    const int sg = (tid >> 7) & 1, st = tid & 127, st_row = st >> 3, st_c8 = st & 7;
    // this lane's two hidden units 8w + 2q, 8w + 2q + 1 sit in 8-channel group w: plane w & 3, slot w >> 2, bytes 4q..4q+3
    const int h_plane = w & 3, h_slot = w >> 2;

    const int64_t n_pairs = (a.n + 31) / 32;
    for (int64_t pr = blockIdx.x; pr < n_pairs; pr += gridDim.x) {
        const int64_t chunk0 = pr * 32;
        int64_t st_chunk = chunk0 + sg * 16 + st_row;
        if (st_chunk >= a.n) st_chunk = a.n - 1;
        const uint4 *xsrc = reinterpret_cast<const uint4 *>(a.x + (size_t)st_chunk * a.T * 64) + st_c8;
        RMR_SYNC();
        if (stager) {
            xs[0][sg][st_c8 & 3][st_row][st_c8 >> 2] = xsrc[0];
            xs[1][sg][st_c8 & 3][st_row][st_c8 >> 2] = xsrc[(size_t)(a.T > 1 ? 1 : 0) * 8];
        }
        RMR_SYNC();

        float c[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        f32x4 accN[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                accN[g][t] = bias[t];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) accN[g][t] = mfma16<F16>(Aih[t][ks], xs[0][g][q][nn][ks], accN[g][t]);
            }
        RMR_SYNC();  // x_0 read by every wave before step 0 overwrites its tile
        auto step = [&](const int t, auto last_c) {
            constexpr bool LAST = decltype(last_c)::value;
            const int tf = (t + 2 < a.T) ? t + 2 : a.T - 1;
            uint4 xnext = make_uint4(0, 0, 0, 0);
            if (stager) xnext = xsrc[(size_t)tf * 8];
            f32x4 acc[2][2];
            uint4 bx[2][2], bh[2][2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                bx[g][0] = xs[(t + 1) & 1][g][q][nn][0];
                bx[g][1] = xs[(t + 1) & 1][g][q][nn][1];
                if (t > 0) {
                    bh[g][0] = hs[(t - 1) & 1][g][q][nn][0];
                    bh[g][1] = hs[(t - 1) & 1][g][q][nn][1];
                }
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc[g][u] = accN[g][u];
                    if (t > 0) {
                        acc[g][u] = mfma16<F16>(Ahh[u][0], bh[g][0], acc[g][u]);
                        acc[g][u] = mfma16<F16>(Ahh[u][1], bh[g][1], acc[g][u]);
                    }
                }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    accN[g][u] = mfma16<F16>(Aih[u][0], bx[g][0], bias[u]);
                    accN[g][u] = mfma16<F16>(Aih[u][1], bx[g][1], accN[g][u]);
                }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const f32x2 hh = lstm_cell2(acc[g][0], acc[g][1], c[g][0], c[g][1]);
                float h0 = hh.x, h1 = hh.y;
                if constexpr (LAST) {
                    h0 = swish_f(h0);
                    h1 = swish_f(h1);
                }
                unsigned hp;
                if constexpr (F16) hp = __builtin_bit_cast(unsigned, f16x2{(_Float16)h0, (_Float16)h1});
                else hp = __builtin_bit_cast(unsigned, bf16x2{(__bf16)h0, (__bf16)h1});
                reinterpret_cast<unsigned *>(&hs[t & 1][g][h_plane][nn][h_slot])[q] = hp;
            }
            if (stager) xs[t & 1][sg][st_c8 & 3][st_row][st_c8 >> 2] = xnext;
            RMR_SYNC();
        };
        for (int t = 0; t + 1 < a.T; ++t) step(t, std::false_type{});
        step(a.T - 1, std::true_type{});
        // ---- lstm2 (one step) + fc, per group ----
        const int u0 = 8 * w + 2 * q;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 acc2[2];
            const uint4 bh0 = hs[(a.T - 1) & 1][g][q][nn][0], bh1 = hs[(a.T - 1) & 1][g][q][nn][1];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                acc2[u] = *reinterpret_cast<const f32x4 *>(a.b2 + ((w * 2 + u) * 4 + q) * 4);
                acc2[u] = mfma16<F16>(a.a_ih2[((w * 2 + u) * 2 + 0) * 64 + lane], bh0, acc2[u]);
                acc2[u] = mfma16<F16>(a.a_ih2[((w * 2 + u) * 2 + 1) * 64 + lane], bh1, acc2[u]);
            }
            float y[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float c2 = 0.f;
                y[u] = swish_f(lstm_cell(acc2[u], c2));
            }
            for (int o = 0; o < a.num_out; ++o) {
                float p = a.w_fc[(size_t)o * 64 + u0] * y[0] + a.w_fc[(size_t)o * 64 + u0 + 1] * y[1];
                p += __shfl_xor(p, 16);
                p += __shfl_xor(p, 32);
                if (q == 0) part[g][w][nn][o] = p;
            }
        }
        RMR_SYNC();
        if (tid < 32 * a.num_out) {
            const int g = tid / (16 * a.num_out), r = tid - g * 16 * a.num_out;
            const int ch = r / a.num_out, o = r - ch * a.num_out;
            if (chunk0 + g * 16 + ch < a.n) {
                float s = a.b_fc[o];
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) s += part[g][ww][ch][o];
                a.logits[(size_t)(chunk0 + g * 16 + ch) * a.num_out + o] = s;
            }
        }
    }
}

int launch_lstm_head_x16(rmr_model *m, const uint16_t *x, int64_t n, float *logits) {
    rmr_engine *e = m->eng;
    if (m->desc.size != 64 || m->nparts != 1 || !m->lstm.x_ih) RMR_FAIL(RMR_ERR_INVALID, "bf16-activation LSTM: size 64, plain bf16 only");
    if (n <= 0) return 0;
    LstmXArgs a;
    a.x = x; a.logits = logits; a.n = n; a.T = m->T; a.num_out = m->desc.num_out;
    a.a_ih = reinterpret_cast<const uint4 *>(m->lstm.x_ih); a.a_hh = reinterpret_cast<const uint4 *>(m->lstm.x_hh);
    a.a_ih2 = reinterpret_cast<const uint4 *>(m->lstm.x_ih2);
    a.b1 = m->lstm.x_b1; a.b2 = m->lstm.x_b2; a.w_fc = m->lstm.w_fc; a.b_fc = m->lstm.b_fc;
    // two 16-chunk groups per block iteration (the one-group kernel measured 2.38 against 2.22 ns per chunk at C100, 5.58 against
    // 4.84 at C200 and is gone)
    const int64_t pairs = (n + 31) / 32;
    int64_t grid2 = (int64_t)e->num_cus * 4;
    if (grid2 > pairs) grid2 = pairs;
    ProfScope ps(e, K_LSTM_HEAD);
    if (m->f16) hipLaunchKernelGGL(lstm_x16_g2_kernel<true>, dim3((unsigned)grid2), dim3(512), 0, e->stream, a);
    else hipLaunchKernelGGL(lstm_x16_g2_kernel<false>, dim3((unsigned)grid2), dim3(512), 0, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

}  // namespace rmr
