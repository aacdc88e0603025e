// k_stream.hip — the networks with MORE THAN 64 CHANNELS (`--size` is any int in the reference: src/remora/parsers.py:858-862,
// models/ConvLSTM_w_ref.py:11-37, models/Conv_w_ref.py:11-42): the same implicit GEMMs as k_conv.hip / k_lstm.hip on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32), with the weight fragments STREAMED from L2 instead of living in VGPRs.
//
// k_conv.hip / k_lstm.hip keep a wave's whole weight slice in registers (merge_conv1 at size 64: 160 VGPRs; lstm1: 128), which
// is what fixed those kernels to 16 / 32 / 64 channels.  At 128 channels merge_conv1's slice is 320 registers and the LSTM's
// 256: here a wave fetches the A fragment of one k-step group (16 bytes per lane = the four MFMAs of one ds_read_b128 B
// fragment) two groups ahead of its use and amortises it over NT column tiles whose accumulators it holds (NT x 4 VGPRs), so
// the L2 -> CU weight traffic is 1 / NT of a fetch per MFMA.  Channel counts are runtime values (any multiple of 16 up to 256;
// engine.hip pads other sizes with zero-weight channels), LDS images are the four-plane layout of k_conv.hip with the row
// stride computed at launch (rowstride / 4 odd: every 16-lane ds_read_b128 group on 16 distinct bank slots).
//
// Arithmetic: identical operations in identical order to the resident kernels (bias first, k ascending over (tap, channel),
// the x projection before the recurrent one; gates on exp2 / rcp with the pre-scaled rows of engine.hip) - fp32 MFMA is a
// k-ordered fmaf chain, so a network padded from 64 to 80 channels returns the 64-channel kernels' bits.
#include "rmr_internal.h"
#include "rmr_math.h"

namespace rmr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int fdiv(int x, FastDiv d) { return (int)(((float)x + 0.5f) * d.inv); }

// =========================================================================================
// convolution + folded BatchNorm + swish
// =========================================================================================
struct ConvSArgs {
    const float *in;
    float *out;
    const float *apack4;  // [oc/16][KW * ic/16 steps][64 lanes][4]: W[16 ot + m][16 g + 4 q + j][tap], step = tap * G + g
    const float *bias;
    int64_t n;
    int ic, oc;           // multiples of 16
    int pin, pout;        // positions per chunk in / out
    int out_row, out_coff;
    int cb;               // chunks per block iteration
    int plane;            // LDS plane stride in floats (multiple of 64)
    int rs;               // floats per row per plane (rs / 4 odd)
    FastDiv div_pout, div_r4, div_g;
    // Position windows, as in k_conv.hip (a chunk whose rows do not fit a block's LDS: chunk contexts of several hundred samples
    // at these channel counts).  nwin > 1: an iteration is ONE window of ONE chunk (cb == 1); `pin` / `pout` are the rows staged /
    // the columns computed per window, the chunk's own extents pin_total / pout_total; window w covers output positions
    // [w * pout, (w + 1) * pout).
    int nwin, pin_total, pout_total;
};

// One work item of a wave: output channels 16 ot .. 16 ot + 15 x NTV column tiles (16 columns each) from column tile `tile0`.
template <int KW, int STRIDE, int NTV>
__device__ __forceinline__ void conv_stream_item(const ConvSArgs &a, const float *smem, int64_t chunk0, int pbase, int ncols, int ot, int tile0,
                                                 int lane, int q, int nn) {
    const int G = a.ic >> 4, RS = a.rs, S4 = KW * G;
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + 16 * ot + 4 * q);
    f32x4 acc[NTV];
    int roff[NTV], ch[NTV], pp[NTV];
    bool valid[NTV];
#pragma unroll
    for (int t = 0; t < NTV; ++t) {
        int col = (tile0 + t) * 16 + nn;
        valid[t] = col < ncols;
        col = valid[t] ? col : ncols - 1;
        ch[t] = fdiv(col, a.div_pout);
        pp[t] = col - ch[t] * a.pout;
        roff[t] = q * a.plane + (ch[t] * a.pin + pp[t] * STRIDE) * RS;
        acc[t] = b4;
    }
    const f32x4 *ap = reinterpret_cast<const f32x4 *>(a.apack4) + (size_t)ot * S4 * 64 + lane;
    f32x4 A0 = ap[0], A1 = ap[(size_t)(S4 > 1 ? 1 : 0) * 64];
    f32x4 x[NTV];
#pragma unroll
    for (int t = 0; t < NTV; ++t) x[t] = *reinterpret_cast<const f32x4 *>(smem + roff[t]);
    int g = 0, tap = 0;
    for (int s = 0; s < S4; ++s) {
        // A fragment two steps ahead (an L2 hit is ~ 500 cycles; a step is 4 NTV MFMAs of 32 cycles)
        const f32x4 A2 = ap[(size_t)(s + 2 < S4 ? s + 2 : S4 - 1) * 64];
        // B fragments of the next step before this step's MFMAs
        int gn = g + 1, tapn = tap;
        if (gn == G) { gn = 0; tapn = tap + 1; }
        const int offn = (s + 1 < S4) ? tapn * RS + 4 * gn : tap * RS + 4 * g;
        f32x4 y[NTV];
#pragma unroll
        for (int t = 0; t < NTV; ++t) y[t] = *reinterpret_cast<const f32x4 *>(smem + roff[t] + offn);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < NTV; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[j], x[t][j], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTV; ++t) x[t] = y[t];
        A0 = A1;
        A1 = A2;
        g = gn;
        tap = tapn;
    }
#pragma unroll
    for (int t = 0; t < NTV; ++t) {
        if (valid[t]) {
            f32x2 lo = f32x2{acc[t][0], acc[t][1]}, hi = f32x2{acc[t][2], acc[t][3]};
            swish_pk(lo, hi);
            float *dst = a.out + ((size_t)(chunk0 + ch[t]) * a.pout_total + pbase + pp[t]) * a.out_row + a.out_coff + 16 * ot + 4 * q;
            *reinterpret_cast<f32x4 *>(dst) = f32x4{lo.x, lo.y, hi.x, hi.y};
        }
    }
}

constexpr int kConvStreamNT = 4;

template <int KW, int STRIDE>
__global__ __launch_bounds__(512) void conv_stream_kernel(ConvSArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = kConvStreamNT;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15, nw = nthr >> 6;
    const int G = a.ic >> 4, RS = a.rs, R4 = a.ic >> 2, OT = a.oc >> 4;
    const int64_t n_iters = a.nwin > 1 ? a.n * a.nwin : (a.n + a.cb - 1) / a.cb;
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        // whole chunks (nwin == 1): cb chunks from chunk0;  windows: window `win` of chunk `chunk0`, rows clipped to the chunk
        int64_t chunk0 = it * a.cb;
        int nch = (int)((a.n - chunk0) < a.cb ? (a.n - chunk0) : a.cb), pbase = 0, rows = nch * a.pin, ncols = nch * a.pout;
        size_t src_row = (size_t)chunk0 * a.pin;
        if (a.nwin > 1) {
            chunk0 = it / a.nwin;
            const int win = (int)(it - chunk0 * a.nwin);
            pbase = win * a.pout;
            nch = 1;
            ncols = a.pout_total - pbase < a.pout ? a.pout_total - pbase : a.pout;
            rows = a.pin_total - pbase * STRIDE < a.pin ? a.pin_total - pbase * STRIDE : a.pin;
            src_row = (size_t)chunk0 * a.pin_total + (size_t)pbase * STRIDE;
        }
        RMR_SYNC();  // all reads of the previous iteration are done
        {            // stage `rows` rows of ic floats into the 4 planes (plane q = channels {16 g + 4 q + j})
            constexpr int UNR = 4;
            const int total4 = rows * R4;
            const float4 *src = reinterpret_cast<const float4 *>(a.in + src_row * a.ic);
            for (int base = tid; base < total4; base += UNR * nthr) {
                float4 v[UNR];
                int dsto[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int i = base + u * nthr;
                    const int ii = i < total4 ? i : total4 - 1;
                    const int row = fdiv(ii, a.div_r4), c = ii - row * R4;
                    const int qq = fdiv(c, a.div_g), g = c - qq * G;  // consecutive lanes: same plane, consecutive g
                    v[u] = src[row * R4 + 4 * g + qq];
                    dsto[u] = i < total4 ? qq * a.plane + row * RS + 4 * g : 4 * a.plane;  // trash slot
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) *reinterpret_cast<float4 *>(smem + dsto[u]) = v[u];
            }
        }
        RMR_SYNC();
        const int ntiles = (ncols + 15) >> 4, ntg = (ntiles + NT - 1) / NT;
        for (int item = w; item < OT * ntg; item += nw) {  // (wave-uniform)
            const int tg = item / OT, ot = item - tg * OT;
            const int tile0 = tg * NT, nt = ntiles - tile0 < NT ? ntiles - tile0 : NT;
            if (nt == 4) conv_stream_item<KW, STRIDE, 4>(a, smem, chunk0, pbase, ncols, ot, tile0, lane, q, nn);
            else if (nt == 3) conv_stream_item<KW, STRIDE, 3>(a, smem, chunk0, pbase, ncols, ot, tile0, lane, q, nn);
            else if (nt == 2) conv_stream_item<KW, STRIDE, 2>(a, smem, chunk0, pbase, ncols, ot, tile0, lane, q, nn);
            else conv_stream_item<KW, STRIDE, 1>(a, smem, chunk0, pbase, ncols, ot, tile0, lane, q, nn);
        }
    }
}

template <int KW, int STRIDE>
static int launch_conv_stream_t(rmr_engine *e, const ConvLayer &c, const float *in, int pin, float *out, int out_row, int out_coff,
                                int pout, int64_t n) {
    const int G = c.ic / 16;
    const int RS = (G % 2 == 0) ? c.ic / 4 + 4 : c.ic / 4;
    const int OT = c.oc / 16;
    // waves per block: one output-channel tile each where that is at most 8, else a divisor of the tile count (equal work)
    int nw = OT;
    if (nw > 8) {
        nw = 8;
        for (int d = 8; d >= 4; --d)
            if (OT % d == 0) { nw = d; break; }
    }
    const size_t row_bytes = (size_t)pin * RS * 4 * sizeof(float);  // all four planes
    const size_t budget = (size_t)65536;  // two blocks per CU
    int cb_max = (int)(budget / row_bytes);
    if (cb_max < 1) cb_max = 1;
    if (cb_max > 8) cb_max = 8;
    int cb = cb_max;
    double best = -1.0;
    for (int k = cb_max; k >= (cb_max + 1) / 2; --k) {  // the chunk count whose columns fill their 16-column tiles best
        const int cols = k * pout;
        const double eff = (double)cols / (16.0 * ((cols + 15) / 16));
        if (eff > best + 1e-9) { best = eff; cb = k; }
    }
    // a chunk whose rows do not fit one block's LDS at all goes through position windows: the most output positions whose input
    // rows ((win - 1) * STRIDE + KW of them) fit the two-blocks-per-CU share, one window of one chunk per iteration
    int nwin = 1, pin_w = pin, pout_w = pout;
    if (row_bytes + 64 > 150 * 1024) {
        const int rows_fit = (int)(budget / ((size_t)RS * 4 * sizeof(float)));
        pout_w = (rows_fit - KW) / STRIDE + 1;
        if (pout_w < 16) RMR_FAIL(RMR_ERR_INVALID, "conv layer %d -> %d channels: not even a 16-column window fits %zu B of LDS", c.ic, c.oc, budget);
        pout_w &= ~15;  // whole column tiles
        nwin = (pout + pout_w - 1) / pout_w;
        pin_w = (pout_w - 1) * STRIDE + KW;
        cb = 1;
    }
    const int plane = ((cb * pin_w * RS) + 63) & ~63;
    const size_t lds = (size_t)plane * 4 * sizeof(float) + 64;  // + trash slot for masked staging writes
    if (lds > 160 * 1024 - 256) RMR_FAIL(RMR_ERR_INVALID, "conv layer %d -> %d channels needs %zu B of LDS", c.ic, c.oc, lds);
    ConvSArgs a;
    a.in = in; a.out = out; a.apack4 = c.apack4; a.bias = c.bias; a.n = n;
    a.ic = c.ic; a.oc = c.oc; a.pin = pin_w; a.pout = pout_w; a.out_row = out_row; a.out_coff = out_coff;
    a.cb = cb; a.plane = plane; a.rs = RS;
    a.div_pout = make_fastdiv(pout_w); a.div_r4 = make_fastdiv(c.ic / 4); a.div_g = make_fastdiv(G);
    a.nwin = nwin; a.pin_total = pin; a.pout_total = pout;
    const int64_t iters = nwin > 1 ? n * nwin : (n + cb - 1) / cb;
    int64_t grid = (int64_t)e->num_cus * 4;
    if (grid > iters) grid = iters;
    if (grid < 1) return 0;
    auto kern = conv_stream_kernel<KW, STRIDE>;
    RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(kern)));
    ProfScope ps(e, c.kid);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * nw), lds, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

int launch_conv_stream(rmr_engine *e, const ConvLayer &c, const float *in, int in_row, int pin, float *out, int out_row, int out_coff,
                       int pout, int64_t n) {
    if (in_row != c.ic) RMR_FAIL(RMR_ERR_INVALID, "conv input row %d != ic %d", in_row, c.ic);
    if (!c.apack4 || c.ic % 16 || c.oc % 16) RMR_FAIL(RMR_ERR_INVALID, "internal: layer %d -> %d not packed for the streamed kernel", c.ic, c.oc);
#define RMR_CONVS_CASE(KW_, ST_) \
    if (c.kw == KW_ && c.stride == ST_) return launch_conv_stream_t<KW_, ST_>(e, c, in, pin, out, out_row, out_coff, pout, n);
    RMR_CONVS_CASE(9, 3)   // sig_conv3; Conv_w_ref seq_conv3
    RMR_CONVS_CASE(13, 3)  // seq_conv2
    RMR_CONVS_CASE(5, 1)   // merge_conv1; Conv_w_ref merge_conv2
    RMR_CONVS_CASE(3, 2)   // Conv_w_ref merge_conv3 / 4
#undef RMR_CONVS_CASE
    RMR_FAIL(RMR_ERR_INVALID, "no streamed conv kernel for kw=%d stride=%d", c.kw, c.stride);
}

// =========================================================================================
// lstm1 (T steps) + lstm2 (ONE step, see k_lstm.hip) + fc
// =========================================================================================
struct LstmSArgs {
    const float *x;  // [n][T][H] channel-last merge_conv1 output
    float *logits;   // [n][num_out]
    // a_ih1 / a_hh1: [H/16 waves][H/16 k groups][4 gates][64 lanes][4] (rows pre-scaled, engine.hip lstm1_gate_scale);
    // a_ih2: [H/16][H/16][3 gates i, g, o][64][4]; b1 [4H] (b_ih + b_hh, pre-scaled), b2 [3H]
    const float *a_ih1, *a_hh1, *b1, *a_ih2, *b2, *w_fc, *b_fc;
    int64_t n;
    int T, num_out, H, rs;
    FastDiv div_r4;
};

// acc[gate][tile] += A (streamed: one 16-byte fragment per gate and k group, fetched a group ahead) x B fragments from one
// LDS image.  SWISH: the B operand is swish(h) (lstm2's input, models/ConvLSTM_w_ref.py:52-53).
template <int NG, int NT, bool SWISH>
__device__ __forceinline__ void stream_mm(const float *img, int plane, int RS, int G, const float *apack, int w, int lane, int q, int nn,
                                          f32x4 (&acc)[NG][NT]) {
    const f32x4 *ap = reinterpret_cast<const f32x4 *>(apack) + (size_t)w * G * NG * 64 + lane;
    const float *b = img + q * plane + nn * RS;
    f32x4 A[NG], An[NG];
#pragma unroll
    for (int gt = 0; gt < NG; ++gt) A[gt] = ap[(size_t)gt * 64];
    for (int g = 0; g < G; ++g) {
        const int gn = g + 1 < G ? g + 1 : g;
#pragma unroll
        for (int gt = 0; gt < NG; ++gt) An[gt] = ap[(size_t)(gn * NG + gt) * 64];
        f32x4 bx[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            bx[t] = *reinterpret_cast<const f32x4 *>(b + t * 16 * RS + 4 * g);
            if (SWISH) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bx[t][j] = swish_f(bx[t][j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int gt = 0; gt < NG; ++gt)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[gt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[gt][j], bx[t][j], acc[gt][t], 0, 0, 0);
#pragma unroll
        for (int gt = 0; gt < NG; ++gt) A[gt] = An[gt];
    }
}

// One block = H/16 waves x 16 NT chunks.  Wave w owns hidden units 16 w .. 16 w + 15 of all four gates (a cell's i, f, g, o in
// one lane: the update is lane-local, c never leaves registers), for NT column tiles.
template <int NT, int MAXT>
__global__ __launch_bounds__(MAXT) void lstm_stream_kernel(LstmSArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, G = H >> 4, RS = a.rs, R4 = H >> 2;
    constexpr int ROWS = 16 * NT;
    const int plane = ROWS * RS, img = 4 * plane;
    float *xbuf = smem, *hbuf = smem + 2 * img;
    float *part = hbuf;  // [H/16][ROWS][16]: the fc partial sums, once the recurrence no longer needs hbuf
    const int tid = threadIdx.x, nthr = blockDim.x;  // nthr == 4 H
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15, NW = nthr >> 6;

    // staging role of this thread: NT 16-byte pieces of the block's ROWS x H tile (ROWS * H / 4 pieces over 4 H threads)
    int st_row[NT], st_dst[NT], st_c4[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int i = tid + u * nthr;
        st_row[u] = fdiv(i, a.div_r4);
        st_c4[u] = i - st_row[u] * R4;
        st_dst[u] = (st_c4[u] & 3) * plane + st_row[u] * RS + 4 * (st_c4[u] >> 2);  // float4 index 4 g + q -> plane q, group g
    }
    const int64_t n_groups = (a.n + ROWS - 1) / ROWS;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t chunk0 = grp * ROWS;
        const float4 *xsrc[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            int64_t chn = chunk0 + st_row[u];
            if (chn >= a.n) chn = a.n - 1;  // clamp the ragged tail (results masked)
            xsrc[u] = reinterpret_cast<const float4 *>(a.x + (size_t)chn * a.T * H) + st_c4[u];
        }
        RMR_SYNC();  // the previous group's LDS traffic is done
#pragma unroll
        for (int u = 0; u < NT; ++u) *reinterpret_cast<float4 *>(xbuf + st_dst[u]) = xsrc[u][0];
        RMR_SYNC();
        f32x4 c[NT], h[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) c[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < a.T; ++t) {
            float4 xn[NT];  // x_{t+1} (clamped: the last fetch is a redundant re-read, never consumed)
            const int tf = t + 1 < a.T ? t + 1 : t;
#pragma unroll
            for (int u = 0; u < NT; ++u) xn[u] = xsrc[u][(size_t)tf * R4];
            f32x4 acc[4][NT];
#pragma unroll
            for (int gt = 0; gt < 4; ++gt) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.b1 + gt * H + 16 * w + 4 * q);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) acc[gt][tt] = b4;
            }
            stream_mm<4, NT, false>(xbuf + (t & 1) * img, plane, RS, G, a.a_ih1, w, lane, q, nn, acc);
            if (t > 0) stream_mm<4, NT, false>(hbuf + ((t - 1) & 1) * img, plane, RS, G, a.a_hh1, w, lane, q, nn, acc);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {  // rows pre-scaled: i, f, o by -log2(e); g by 2 log2(e) (k_lstm.hip lstm_step)
                    const float ig = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[0][tt][r]));
                    const float fg = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[1][tt][r]));
                    const float gg = fmaf(-2.0f, fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[2][tt][r])), 1.0f);
                    c[tt][r] = fmaf(fg, c[tt][r], ig * gg);
                    const float og = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[3][tt][r]));
                    h[tt][r] = og * tanh_f(c[tt][r]);
                }
                *reinterpret_cast<f32x4 *>(hbuf + (t & 1) * img + q * plane + (tt * 16 + nn) * RS + 4 * w) = h[tt];
            }
            if (t + 1 < a.T) {  // xbuf[(t + 1) & 1] was last read in step t - 1, a barrier ago
#pragma unroll
                for (int u = 0; u < NT; ++u) *reinterpret_cast<float4 *>(xbuf + ((t + 1) & 1) * img + st_dst[u]) = xn[u];
            }
            RMR_SYNC();
        }
        // ---- lstm2: one step on swish(h1[T-1]), gates i, g, o only (c0 = 0 kills f) ----
        f32x4 acc2[3][NT];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.b2 + gt * H + 16 * w + 4 * q);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) acc2[gt][tt] = b4;
        }
        stream_mm<3, NT, true>(hbuf + ((a.T - 1) & 1) * img, plane, RS, G, a.a_ih2, w, lane, q, nn, acc2);
        f32x4 y[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float c2 = sigmoid_f(acc2[0][tt][r]) * tanh_f(acc2[1][tt][r]);
                const float h2 = sigmoid_f(acc2[2][tt][r]) * tanh_f(c2);
                y[tt][r] = swish_f(h2);
            }
        RMR_SYNC();  // every wave has read h1[T-1]: its image becomes `part`
        // ---- fc: partial dot over this lane's 4 hidden units, reduce over q then over the waves ----
        for (int o = 0; o < a.num_out; ++o) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(a.w_fc + (size_t)o * H + 16 * w + 4 * q);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                float p = wv[0] * y[tt][0] + wv[1] * y[tt][1] + wv[2] * y[tt][2] + wv[3] * y[tt][3];
                p += __shfl_xor(p, 16);
                p += __shfl_xor(p, 32);
                if (q == 0) part[((size_t)w * ROWS + tt * 16 + nn) * 16 + o] = p;
            }
        }
        RMR_SYNC();
        for (int idx = tid; idx < ROWS * a.num_out; idx += nthr) {
            const int col = idx / a.num_out, o = idx - col * a.num_out;
            if (chunk0 + col < a.n) {
                float s = a.b_fc[o];
                for (int ww = 0; ww < NW; ++ww) s += part[((size_t)ww * ROWS + col) * 16 + o];
                a.logits[(size_t)(chunk0 + col) * a.num_out + o] = s;
            }
        }
    }
}

template <int NT, int MAXT>
static int launch_lstm_stream_t(rmr_model *m, const LstmSArgs &a, int64_t n) {
    rmr_engine *e = m->eng;
    const int H = a.H;
    const size_t lds = (size_t)4 * 4 * 16 * NT * a.rs * sizeof(float);
    if (lds > 160 * 1024 - 256) RMR_FAIL(RMR_ERR_INVALID, "streamed LSTM: %zu B of LDS for %d hidden units", lds, H);
    const int64_t groups = (n + 16 * NT - 1) / (16 * NT);
    int64_t grid = (int64_t)e->num_cus * 4;
    if (grid > groups) grid = groups;
    if (grid < 1) return 0;
    auto kern = lstm_stream_kernel<NT, MAXT>;
    RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(kern)));
    ProfScope ps(e, K_LSTM_HEAD);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(4 * H), lds, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

int launch_lstm_stream(rmr_model *m, const float *x, int64_t n, float *logits) {
    const int H = m->desc.size;
    if (H % 16 || H > 256 || !m->lstm.t_ih1) RMR_FAIL(RMR_ERR_INVALID, "internal: LSTM of %d units not packed for the streamed kernel", H);
    const int G = H / 16;
    LstmSArgs a;
    a.x = x; a.logits = logits; a.n = n; a.T = m->T; a.num_out = m->desc.num_out; a.H = H;
    a.rs = (G % 2 == 0) ? H / 4 + 4 : H / 4;
    a.div_r4 = make_fastdiv(H / 4);
    a.a_ih1 = m->lstm.t_ih1; a.a_hh1 = m->lstm.t_hh1; a.b1 = m->lstm.b1;
    a.a_ih2 = m->lstm.t_ih2; a.b2 = m->lstm.b2; a.w_fc = m->lstm.w_fc; a.b_fc = m->lstm.b_fc;
    // column tiles per wave: four halve the weight traffic of two, and fit (registers: 512 threads; LDS) up to 128 units
    const int nt = (H <= 128 ? 4 : 2);
    if (H <= 128 && nt == 4) return launch_lstm_stream_t<4, 512>(m, a, n);
    if (H <= 128) return launch_lstm_stream_t<2, 512>(m, a, n);
    return launch_lstm_stream_t<2, 1024>(m, a, n);
}

}  // namespace rmr
