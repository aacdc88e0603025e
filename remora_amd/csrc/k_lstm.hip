// k_lstm.hip — lstm1 (T steps) + lstm2 (ONE step) + fc head of ConvLSTM_w_ref on the fp32
// matrix cores, and the fc head of Conv_w_ref.
//
// Replaces models/ConvLSTM_w_ref.py:51-56:
//     z = swish(lstm1(z)[0]); z = flip(swish(lstm2(flip(z))[0])); z = z[-1]; z = fc(z)
// After the second flip, z[-1] is the FIRST output of lstm2 run on the reversed sequence:
// it depends only on swish(h1[T-1]) with zero initial state, so lstm2 is evaluated as a
// single step (gates = W_ih x + b_ih + b_hh; c = sig(i) tanh(g); h = sig(o) tanh(c)); the
// other T-1 steps of lstm2 never reach the output.  torch gate order i,f,g,o.
//
// One block = H/16 waves handles 16 chunks at a time (one 16-column MFMA tile).
// Wave w owns hidden units 16w..16w+15 for ALL four gates: its four 16x16 D tiles
// (i,f,g,o) put the four gates of one (hidden unit, chunk) in the same lane and register
// index, so the cell update is lane-local and c stays in registers for all T steps.
// W_ih and W_hh slices (2 * 4 * H/4 VGPRs) are register-resident for the block lifetime.
// h_t goes through LDS once per step (every wave needs all H values as its B operand):
// D fragment = 4 consecutive hidden units of one chunk = one ds_write_b128; B fragment =
// one ds_read_b128 per 16-wide k group, each feeding 4 MFMAs x 4 gates.
#include "rmr_internal.h"
#include "rmr_math.h"

namespace rmr {

typedef float f32x4 __attribute__((ext_vector_type(4)));


struct LstmArgs {
    const float *x;  // [n][T][H] channel-last merge_conv1 output
    float *logits;   // [n][num_out]
    const float *a_ih1, *a_hh1, *b1, *a_ih2, *b2, *w_fc, *b_fc;
    int64_t n;
    int T, num_out;
};

// acc[gt] += A[gt][:] (register-resident weight slice) x B fragments read from one LDS image
template <int KS, int G, int RS>
__device__ __forceinline__ void xproj(const float (&buf)[4][16][RS], int q, int nn,
                                      const float (&A)[4][KS], f32x4 (&acc)[4]) {
    const float *b = &buf[q][nn][0];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const f32x4 bx = *reinterpret_cast<const f32x4 *>(b + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
                acc[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[gt][g * 4 + j], bx[j], acc[gt], 0, 0, 0);
    }
}

// fc partial sum over four consecutive hidden units, as ONE stated chain of operations (the small-batch kernel below forms the
// same chain from other lanes' values and has to return the same bits)
__device__ __forceinline__ float fc_dot4(const f32x4 wv, const f32x4 y) {
    return fmaf(wv[3], y[3], fmaf(wv[2], y[2], fmaf(wv[1], y[1], wv[0] * y[0])));
}

// LSTM cell update for this lane's 4 hidden units (torch gate order i, f, g, o)
__device__ __forceinline__ void gates(const f32x4 (&acc)[4], f32x4 &c, f32x4 &h) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float ig = sigmoid_f(acc[0][r]), fg = sigmoid_f(acc[1][r]);
        const float gg = tanh_f(acc[2][r]), og = sigmoid_f(acc[3][r]);
        c[r] = fg * c[r] + ig * gg;
        h[r] = og * tanh_f(c[r]);
    }
}

// One LSTM step for this wave's 16 hidden units x 16 chunks.
//   acc  = accN (bias + W_ih x_t, computed during the previous step) [+ W_hh h_{t-1} if HAS_H]
//   accN = bias + W_ih x_{t+1}   [if HAS_X]   -- independent of h_t
// The fp32 MFMA shares the SIMD's fp32 datapath with ordinary VALU work (ablation: removing
// the gate transcendentals raised the kernel from 114 to 134 TFLOP/s), so the gate math is
// kept minimal: the i/f/o rows of W and b are pre-scaled by -log2(e) and the g rows by
// 2 log2(e) on the host (engine.hip), making sigmoid = rcp(1 + exp2(a)) and
// tanh = 1 - 2 rcp(1 + exp2(a)) three/four instructions each.  The projection MFMAs of step
// t+1 are sliced between the gate slices so that MFMA issue never waits for a long
// dependent VALU chain.
template <int H, bool HAS_H, bool HAS_X, int ABL>
__device__ __forceinline__ void lstm_step(const float (&xb_img)[4][16][(H / 16 % 2 == 0) ? H / 4 + 4 : H / 4],
                                          const float (&hb_img)[4][16][(H / 16 % 2 == 0) ? H / 4 + 4 : H / 4],
                                          int q, int nn, const float (&Aih)[4][H / 4], const float (&Ahh)[4][H / 4],
                                          const f32x4 (&bias)[4], f32x4 (&accN)[4], f32x4 &c, f32x4 &h) {
    constexpr int KS = H / 4, G = H / 16;
    constexpr int RS = (G % 2 == 0) ? H / 4 + 4 : H / 4;
    f32x4 acc[4] = {accN[0], accN[1], accN[2], accN[3]};
    f32x4 bx[G];
    if (HAS_X) {  // B fragments of x_{t+1} first (LDS latency hidden behind the recurrent MFMAs)
        const float *xb = &xb_img[q][nn][0];
#pragma unroll
        for (int g = 0; g < G; ++g) bx[g] = *reinterpret_cast<const f32x4 *>(xb + 4 * g);
    }
    if (HAS_H) xproj<KS, G, RS>(hb_img, q, nn, Ahh, acc);  // recurrent critical path
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) accN[gt] = bias[gt];
    float ig[4], fg[4], gg[4];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        __builtin_amdgcn_sched_barrier(0);
        if (HAS_X) {
#pragma unroll
            for (int gt = 0; gt < 4; ++gt)
                accN[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Aih[gt][s], bx[s >> 2][s & 3], accN[gt], 0, 0, 0);
        }
        // gate work: 16 pieces (r = 0..3 x {i, f, g + c, o + h}) spread over the KS slices
        for (int piece = s * 16 / KS; piece < (s + 1) * 16 / KS; ++piece) {
            const int r = piece >> 2, st = piece & 3;
            if (ABL & 1) {  // timing ablation: no transcendental work
                if (st == 3) { c[r] += acc[0][r] * acc[1][r]; h[r] = acc[2][r] + acc[3][r] * c[r]; }
                continue;
            }
            // acc rows are pre-scaled: i,f,o by -log2(e); g by 2 log2(e)
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
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int H, int ABL = 0>
__global__ __launch_bounds__(4 * H, 2) void lstm_head_kernel(LstmArgs a) {
    constexpr int NW = H / 16;   // waves
    constexpr int KS = H / 4;    // MFMA k-steps per operand
    constexpr int G = H / 16;    // 16-wide k groups
    // LDS images are 4 planes (plane q = channels {16g+4q+j}), rows of H/4 floats padded so that
    // rowstride/4 is odd, plane size a multiple of 64 floats: every 16-lane ds_read_b128 group
    // hits 16 distinct bank slots (see k_conv.hip).
    constexpr int RS = (G % 2 == 0) ? H / 4 + 4 : H / 4;
    static_assert((16 * RS) % 64 == 0, "plane must be a multiple of 64 floats");
    __shared__ __attribute__((aligned(16))) float xbuf[2][4][16][RS];
    __shared__ __attribute__((aligned(16))) float hbuf[2][4][16][RS];
    __shared__ float part[NW][16][16];
    // lstm2's W_ih fragments of this wave ([3 gates][KS][64 lanes], 12 KB at H = 64), copied once: the one step of lstm2 per 16-chunk
    // group used to fetch each fragment from L2 right in front of its MFMA - 48 dependent loads (their 24 address pairs spilled
    // to scratch: the kernel's 55 spilled registers) at the end of every group
    extern __shared__ __attribute__((aligned(16))) float w2_lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15;
    float *const w2 = w2_lds + (size_t)w * 3 * KS * 64 + lane;
    {
        const float *src = a.a_ih2 + (size_t)w * 3 * KS * 64 + lane;
#pragma unroll 8
        for (int i = 0; i < 3 * KS; ++i) w2[i * 64] = src[(size_t)i * 64];  // (read by this wave only; the group loop's first barrier orders it anyway)
    }

    float Aih[4][KS], Ahh[4][KS];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) {
        const float *pi = a.a_ih1 + ((size_t)(w * 4 + gt) * KS) * 64 + lane;
        const float *ph = a.a_hh1 + ((size_t)(w * 4 + gt) * KS) * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Aih[gt][s] = pi[(size_t)s * 64];
            Ahh[gt][s] = ph[(size_t)s * 64];
        }
    }
    f32x4 bias[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
        bias[gt] = *reinterpret_cast<const f32x4 *>(a.b1 + gt * H + 16 * w + 4 * q);

    // staging role of this thread: chunk row = tid / (H/4), 16-byte piece = tid % (H/4)
    const int st_row = tid / (H / 4), st_c4 = tid - st_row * (H / 4);
    const int st_q = st_c4 & 3, st_g = st_c4 >> 2;  // float4 index 4g+q -> plane q, group g

    // (a one-off delay of every other block, to de-phase co-resident blocks, measured no gain in round 2 and is gone)
    const int64_t n_groups = (a.n + 15) / 16;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t chunk0 = grp * 16;
        int64_t st_chunk = chunk0 + st_row;
        if (st_chunk >= a.n) st_chunk = a.n - 1;  // clamp ragged tail (results masked)
        const float4 *xsrc = reinterpret_cast<const float4 *>(a.x + (size_t)st_chunk * a.T * H) + st_c4;
        RMR_SYNC();  // previous group's LDS traffic is done
        *reinterpret_cast<float4 *>(&xbuf[0][st_q][st_row][4 * st_g]) = xsrc[0];
        // second x tile (prefetch distance 2: x_{t+2} is fetched while step t runs)
        *reinterpret_cast<float4 *>(&xbuf[1][st_q][st_row][4 * st_g]) = xsrc[(size_t)(a.T > 1 ? 1 : 0) * (H / 4)];
        RMR_SYNC();

        // Software pipeline: the input projection of step t+1 (accN = b + W_ih x_{t+1}) does not
        // depend on h_t, so its 4*KS MFMAs are issued in the same basic block as the gate
        // non-linearities of step t and keep the matrix pipe busy while the VALU/transcendental
        // work runs; only W_hh h_{t-1} (4*KS MFMAs) sits on the recurrent critical path.
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        f32x4 accN[4] = {bias[0], bias[1], bias[2], bias[3]};
        xproj<KS, G, RS>(xbuf[0], q, nn, Aih, accN);
        // step 0 ends with xbuf[0] overwritten (x_2): every wave has to be past its x_0 reads first (k_lstm_x16.hip has the
        // account of what happened without this barrier when processes shared the GPU)
        RMR_SYNC();
        for (int t = 0; t < a.T; ++t) {
            // x_{t+2} (clamped: the last two fetches are redundant re-reads, never consumed)
            const int tf = (t + 2 < a.T) ? t + 2 : a.T - 1;
            const float4 xnext = xsrc[(size_t)tf * (H / 4)];
            f32x4 h;
            if (t == 0)
                lstm_step<H, false, true, ABL>(xbuf[1], hbuf[1], q, nn, Aih, Ahh, bias, accN, c, h);
            else if (t + 1 < a.T)
                lstm_step<H, true, true, ABL>(xbuf[(t + 1) & 1], hbuf[(t + 1) & 1], q, nn, Aih, Ahh, bias, accN, c, h);
            else
                lstm_step<H, true, false, ABL>(xbuf[(t + 1) & 1], hbuf[(t + 1) & 1], q, nn, Aih, Ahh, bias, accN, c, h);
            *reinterpret_cast<f32x4 *>(&hbuf[t & 1][q][nn][4 * w]) = h;
            *reinterpret_cast<float4 *>(&xbuf[t & 1][st_q][st_row][4 * st_g]) = xnext;
            if (!(ABL & 2)) RMR_SYNC();  // ABL&2: timing ablation without the per-step barrier
        }

        // ---- lstm2: one step on swish(h1[T-1]), gates i, g, o only (c0 = 0 kills f) ----
        f32x4 acc2[3];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
            acc2[gt] = *reinterpret_cast<const f32x4 *>(a.b2 + gt * H + 16 * w + 4 * q);
        {
            const float *hb = &hbuf[(a.T - 1) & 1][q][nn][0];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                f32x4 z = *reinterpret_cast<const f32x4 *>(hb + 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = swish_f(z[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) {
                        const float aw = w2[(gt * KS + g * 4 + j) * 64];
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
        // ---- fc: partial dot over this lane's 4 hidden units, reduce over q then waves ----
        for (int o = 0; o < a.num_out; ++o) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(a.w_fc + (size_t)o * H + 16 * w + 4 * q);
            float p = fc_dot4(wv, y);
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

// ---------------------------------------------------------------------------------------
// Small batches (one read's few hundred chunks: rmr_call_read, inference.call_read_mods): FOUR chunks per block on
// v_mfma_f32_4x4x1_16b_f32.  lstm_head_kernel above fills a 16-column MFMA tile per block: a 5 kb read's 312 chunks are 20
// blocks on 256 CUs, each walking its T steps at 2 x 64 MFMAs of 32 cycles per step and wave - 64 of the call's 188 us.  The
// 16-block 4x4x1 form (8 cycles) computes 64 rows x 4 columns x 1 k per instruction: with 4 chunks per block the same read is
// 78 blocks and a step's recurrent product is 64 instructions of 8 cycles.
//   lane l of wave w: hidden unit u = 16 w + (l >> 2); A operand = W[gate = l & 3][u][k] (block l >> 2 = the unit, rows = its four
//   gates), B operand = h[k][chunk = l & 3] (the same for every block), D = f32x4 {i, f, g, o} of (u, chunk l & 3): the cell
//   update is lane-local, one cell per lane.
// One MFMA per k, issued in lstm_head_kernel's k order (16-channel group g, MFMA j, k-lane q: k = 16 g + 4 q + j), bias first,
// the x projection before the recurrent product - an fp32 MFMA is a k-ordered fmaf chain, so every gate pre-activation, hence
// every logit, equals lstm_head_kernel's bit for bit (tests/test_gpu_parity.py), whatever the batch size a chunk arrives in.
// The fc partial sums are formed in the same tree: fc_dot4 over the four units of a (wave, q) group, (q0 + q1) + (q2 + q3), then
// the waves in order behind the bias.
// ---------------------------------------------------------------------------------------
struct LstmSmallArgs {
    const float *x;  // [n][T][64]
    float *logits;
    const float *s_ih1, *s_hh1, *s_ih2;  // [4 waves][64 k in issue order][64 lanes]; lstm2: gates i, g, o, (zero)
    const float *b1, *b2, *w_fc, *b_fc;  // as LstmArgs
    int64_t n;
    int T, num_out;
};

// Two chains, instruction by instruction: the recurrent product of this step (acc += W_hh h_{t-1}) and the input projection of
// the next (accN += W_ih x_{t+1}) - each keeps its own k order, neither waits for the other's 4x4x1 result, so the matrix pipe
// issues back to back.  `hrow` / `xrow`: the 64 values of this lane's chunk.
template <bool HAS_H, bool HAS_X>
__device__ __forceinline__ void small_step(const float (&Ahh)[64], const float (&Aih)[64], const float *hrow, const float *xrow, f32x4 &acc,
                                           f32x4 &accN) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 hv[4], xv[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            if (HAS_H) hv[qq] = *reinterpret_cast<const f32x4 *>(hrow + 16 * g + 4 * qq);
            if (HAS_X) xv[qq] = *reinterpret_cast<const f32x4 *>(xrow + 16 * g + 4 * qq);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {  // k = 16 g + 4 qq + j
                if (HAS_H) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(Ahh[(g * 4 + j) * 4 + qq], hv[qq][j], acc, 0, 0, 0);
                if (HAS_X) accN = __builtin_amdgcn_mfma_f32_4x4x1f32(Aih[(g * 4 + j) * 4 + qq], xv[qq][j], accN, 0, 0, 0);
            }
    }
}

__global__ __launch_bounds__(256, 1) void lstm_small_kernel(LstmSmallArgs a) {
    constexpr int H = 64;
    // (rows padded by four floats: the four chunk rows a ds_read_b128 touches - every lane reads its own chunk's row - start 16 bytes
    //  apart in the bank map instead of on the same banks)
    __shared__ __attribute__((aligned(16))) float xbuf[2][4][H + 4];
    __shared__ __attribute__((aligned(16))) float hbuf[2][4][H + 4];
    __shared__ __attribute__((aligned(16))) float ybuf[4][H + 4];   // swish(h2) per chunk and unit, for the fc tree
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 3, u = 16 * w + (lane >> 2);
    float Aih[H], Ahh[H];
#pragma unroll
    for (int p = 0; p < H; ++p) {
        Aih[p] = a.s_ih1[((size_t)w * H + p) * 64 + lane];
        Ahh[p] = a.s_hh1[((size_t)w * H + p) * 64 + lane];
    }
    const f32x4 bias = {a.b1[0 * H + u], a.b1[1 * H + u], a.b1[2 * H + u], a.b1[3 * H + u]};
    const f32x4 bias2 = {a.b2[0 * H + u], a.b2[1 * H + u], a.b2[2 * H + u], 0.0f};
    // acc += W (this lane's A values, issue order) x v (the 64 values of this lane's chunk, natural order in LDS)
    auto mm = [&](const float (&A)[H], const float *row, f32x4 acc, bool swish) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v[4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                v[qq] = *reinterpret_cast<const f32x4 *>(row + 16 * g + 4 * qq);
                if (swish) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[qq][j] = swish_f(v[qq][j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)  // k = 16 g + 4 qq + j
                    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[(g * 4 + j) * 4 + qq], v[qq][j], acc, 0, 0, 0);
        }
        return acc;
    };
    const int st_c = tid >> 6, st_k = tid & 63;  // staging role: one float of the block's 4 x 64 tile
    const int64_t n_groups = (a.n + 3) / 4;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t chunk0 = grp * 4;
        int64_t st_chunk = chunk0 + st_c;
        if (st_chunk >= a.n) st_chunk = a.n - 1;  // clamp the ragged tail (results masked)
        const float *xsrc = a.x + (size_t)st_chunk * a.T * H + st_k;
        RMR_SYNC();  // the previous group's LDS traffic is done
        xbuf[0][st_c][st_k] = xsrc[0];
        xbuf[1][st_c][st_k] = xsrc[(size_t)(a.T > 1 ? 1 : 0) * H];  // prefetch distance 2: x_{t+2} is fetched while step t runs
        RMR_SYNC();
        float cst = 0.0f;
        f32x4 accN = mm(Aih, &xbuf[0][c][0], bias, false);  // bias + W_ih x_0
        RMR_SYNC();  // step 0 ends with xbuf[0] overwritten (x_2): every wave is past its x_0 reads (as in lstm_head_kernel)
        for (int t = 0; t < a.T; ++t) {
            const float xnext = xsrc[(size_t)(t + 2 < a.T ? t + 2 : a.T - 1) * H];
            f32x4 acc = accN;
            accN = bias;
            const float *hrow = &hbuf[(t + 1) & 1][c][0], *xrow = &xbuf[(t + 1) & 1][c][0];
            if (t == 0) small_step<false, true>(Ahh, Aih, hrow, xrow, acc, accN);
            else if (t + 1 < a.T) small_step<true, true>(Ahh, Aih, hrow, xrow, acc, accN);
            else small_step<true, false>(Ahh, Aih, hrow, xrow, acc, accN);
            // acc rows are pre-scaled: i, f, o by -log2(e); g by 2 log2(e) (lstm_step)
            const float ig = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[0]));
            const float fg = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[1]));
            const float gg = fmaf(-2.0f, fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[2])), 1.0f);
            cst = fmaf(fg, cst, ig * gg);
            const float og = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[3]));
            hbuf[t & 1][c][u] = og * tanh_f(cst);
            xbuf[t & 1][st_c][st_k] = xnext;  // x_{t+2} over x_t, whose projection was read during step t - 1
            RMR_SYNC();
        }
        // ---- lstm2: one step on swish(h1[T-1]), gates i, g, o (c0 = 0 kills f) ----
        float A2[H];
#pragma unroll
        for (int p = 0; p < H; ++p) A2[p] = a.s_ih2[((size_t)w * H + p) * 64 + lane];
        const f32x4 acc2 = mm(A2, &hbuf[(a.T - 1) & 1][c][0], bias2, true);
        const float c2 = sigmoid_f(acc2[0]) * tanh_f(acc2[1]);
        const float h2 = sigmoid_f(acc2[2]) * tanh_f(c2);
        ybuf[c][u] = swish_f(h2);
        RMR_SYNC();
        // ---- fc: thread (chunk, class) sums in lstm_head_kernel's tree ----
        if (tid < 4 * a.num_out) {
            const int ch = tid / a.num_out, o = tid - ch * a.num_out;
            if (chunk0 + ch < a.n) {
                float s = a.b_fc[o];
                for (int ww = 0; ww < 4; ++ww) {
                    float pq[4];
                    for (int qq = 0; qq < 4; ++qq) {
                        const f32x4 wv = *reinterpret_cast<const f32x4 *>(a.w_fc + (size_t)o * H + 16 * ww + 4 * qq);
                        const f32x4 yv = *reinterpret_cast<const f32x4 *>(&ybuf[ch][16 * ww + 4 * qq]);
                        pq[qq] = fc_dot4(wv, yv);
                    }
                    s += (pq[0] + pq[1]) + (pq[2] + pq[3]);
                }
                a.logits[(size_t)(chunk0 + ch) * a.num_out + o] = s;
            }
        }
    }
}

// batches up to this many chunks take the four-chunk kernel: one block per CU in one wave of blocks
static constexpr int64_t kLstmSmallMaxChunks = 1024;

template <int H>
static int launch_lstm_t(rmr_model *m, const float *x, int64_t n, float *logits) {
    rmr_engine *e = m->eng;
    LstmArgs a;
    a.x = x; a.logits = logits; a.n = n; a.T = m->T; a.num_out = m->desc.num_out;
    a.a_ih1 = m->lstm.a_ih1; a.a_hh1 = m->lstm.a_hh1; a.b1 = m->lstm.b1;
    a.a_ih2 = m->lstm.a_ih2; a.b2 = m->lstm.b2; a.w_fc = m->lstm.w_fc; a.b_fc = m->lstm.b_fc;
    const int64_t groups = (n + 15) / 16;
    int64_t grid = (int64_t)e->num_cus * 16;
    if (grid > groups) grid = groups;
    if (grid < 1) return 0;
    const size_t lds = (size_t)(H / 16) * 3 * (H / 4) * 64 * sizeof(float);  // lstm2's fragments; + 24 KB of static images: 72 KB at H = 64, two blocks per CU
    RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(lstm_head_kernel<H, 0>), lds));
    ProfScope ps(e, K_LSTM_HEAD);
#ifdef RMR_TIMING_ABLATIONS  // experiment builds only (make CXXFLAGS+=-DRMR_TIMING_ABLATIONS): variants that skip work
    const int abl = abl_int("RMR_LSTM_ABLATE", 0);
    if (abl >= 1 && abl <= 3) {
        const void *k = abl == 1 ? reinterpret_cast<const void *>(lstm_head_kernel<H, 1>) : abl == 2 ? reinterpret_cast<const void *>(lstm_head_kernel<H, 2>) : reinterpret_cast<const void *>(lstm_head_kernel<H, 3>);
        RMR_TRY(e->allow_big_lds(k, lds));
    }
    if (abl == 1) hipLaunchKernelGGL((lstm_head_kernel<H, 1>), dim3((unsigned)grid), dim3(4 * H), lds, e->stream, a);
    else if (abl == 2) hipLaunchKernelGGL((lstm_head_kernel<H, 2>), dim3((unsigned)grid), dim3(4 * H), lds, e->stream, a);
    else if (abl == 3) hipLaunchKernelGGL((lstm_head_kernel<H, 3>), dim3((unsigned)grid), dim3(4 * H), lds, e->stream, a);
    else
#endif
    hipLaunchKernelGGL((lstm_head_kernel<H, 0>), dim3((unsigned)grid), dim3(4 * H), lds, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

int launch_lstm_head(rmr_model *m, const float *x, int64_t n, float *logits) {
    if (m->desc.size > 64) return launch_lstm_stream(m, x, n, logits);  // k_stream.hip
    if (m->desc.size == 64 && n <= kLstmSmallMaxChunks && m->lstm.q_ih1) {
        rmr_engine *e = m->eng;
        LstmSmallArgs a;
        a.x = x; a.logits = logits; a.n = n; a.T = m->T; a.num_out = m->desc.num_out;
        a.s_ih1 = m->lstm.q_ih1; a.s_hh1 = m->lstm.q_hh1; a.s_ih2 = m->lstm.q_ih2;
        a.b1 = m->lstm.b1; a.b2 = m->lstm.b2; a.w_fc = m->lstm.w_fc; a.b_fc = m->lstm.b_fc;
        const int64_t groups = (n + 3) / 4;
        if (groups < 1) return 0;
        ProfScope ps(e, K_LSTM_HEAD);
        hipLaunchKernelGGL(lstm_small_kernel, dim3((unsigned)groups), dim3(256), 0, e->stream, a);
        RMR_HIP(hipGetLastError());
        return 0;
    }
    switch (m->desc.size) {
        case 64: return launch_lstm_t<64>(m, x, n, logits);
        case 32: return launch_lstm_t<32>(m, x, n, logits);
        case 16: return launch_lstm_t<16>(m, x, n, logits);
    }
    RMR_FAIL(RMR_ERR_INVALID, "unsupported LSTM size %d", m->desc.size);
}

// ---- Conv_w_ref head: flatten (channel-major, models/Conv_w_ref.py:59) + fc (:60) --------
// one wavefront per chunk: coalesced read of the chunk's t4*size activations, per-lane partial
// dot products for every class, butterfly reduction over the 64 lanes
__global__ __launch_bounds__(256) void fc_head_kernel(const float *m4, const float *w, const float *b, float *logits,
                                                      int64_t n, int size, int t4, int num_out) {
    const int lane = threadIdx.x & 63;
    const int64_t ch = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ch >= n) return;
    const float *src = m4 + (size_t)ch * t4 * size;  // channel-last [t4][size]
    const int total = t4 * size;
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = 0.0f;
    for (int e = lane; e < total; e += 64) {
        const float x = src[e];
        const int t = e / size, c = e - t * size;  // torch flatten index = c * t4 + t
        for (int o = 0; o < num_out; ++o) acc[o] += w[(size_t)o * total + c * t4 + t] * x;
    }
    for (int o = 0; o < num_out; ++o) {
        float v = acc[o];
        for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
        if (lane == 0) logits[ch * num_out + o] = v + b[o];
    }
}

int launch_fc_head(rmr_model *m, const float *m4, int64_t n, float *logits) {
    rmr_engine *e = m->eng;
    const int64_t total = n * m->desc.num_out;
    if (total == 0) return 0;
    ProfScope ps(e, K_FC_HEAD);
    hipLaunchKernelGGL(fc_head_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, e->stream,
                       m4, m->w_fc, m->b_fc, logits, n, m->desc.size, m->T4, m->desc.num_out);
    RMR_HIP(hipGetLastError());
    return 0;
}

}  // namespace rmr
