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

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15;

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
        // ---- fc: partial dot over this lane's 4 hidden units, reduce over q then waves ----
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
    ProfScope ps(e, K_LSTM_HEAD);
#ifdef RMR_TIMING_ABLATIONS  // experiment builds only (make CXXFLAGS+=-DRMR_TIMING_ABLATIONS): variants that skip work
    const int abl = abl_int("RMR_LSTM_ABLATE", 0);
    if (abl == 1) hipLaunchKernelGGL((lstm_head_kernel<H, 1>), dim3((unsigned)grid), dim3(4 * H), 0, e->stream, a);
    else if (abl == 2) hipLaunchKernelGGL((lstm_head_kernel<H, 2>), dim3((unsigned)grid), dim3(4 * H), 0, e->stream, a);
    else if (abl == 3) hipLaunchKernelGGL((lstm_head_kernel<H, 3>), dim3((unsigned)grid), dim3(4 * H), 0, e->stream, a);
    else
#endif
    hipLaunchKernelGGL((lstm_head_kernel<H, 0>), dim3((unsigned)grid), dim3(4 * H), 0, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

int launch_lstm_head(rmr_model *m, const float *x, int64_t n, float *logits) {
    if (m->desc.size > 64) return launch_lstm_stream(m, x, n, logits);  // k_stream.hip
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
