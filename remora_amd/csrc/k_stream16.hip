// k_stream16.hip — ConvLSTM_w_ref with MORE THAN 64 CHANNELS in the 16-bit dtypes (bf16 / f16 operands, fp32 accumulation):
// the streamed-weight kernels of k_stream.hip on v_mfma_f32_16x16x32_bf16 / _f16.
//
// k_fused.hip is the 16-bit pipeline of the 64-channel network: its LDS plan (every intermediate of four chunks resident) and
// its register-resident merge_conv1 fragments do not stretch past 64 channels.  Here the pipeline of a larger network is
//   front_sig / front_seq (k_front.hip, fp32 VALU: sig_conv1/2, seq_conv1)  ->  sig2, seq1 fp32 [16 channels]
//   conv_stream16 x 3 (sig_conv3, seq_conv2 -> cat, merge_conv1 -> x): 16-bit activations in HBM and LDS
//   lstm_stream16: lstm1 (T steps) + lstm2 (one step) + fc
// with the A fragments (8 consecutive k of one output row per lane, 16 bytes) streamed from L2 two k-steps ahead of their use
// and amortised over NT column tiles.  The 16-bit MFMA eats K 16 times faster than the fp32 one, so these kernels lean on
// the L2 -> CU path where k_stream.hip leans on the matrix pipe; they are the way a 128-channel network runs at all in the
// dtypes BASELINE configs[3] / [4] name, not a roofline exhibit (measured: DESIGN section 4).
//
// LDS images are flat rows [row][channels] of 16-bit values + 16 bytes of padding per row (row bytes / 16 is odd: the 16 lanes
// of a ds_read_b128 service group - consecutive columns - land on distinct 16-byte bank slots); K = (tap, channel) of column
// (chunk, pos) is the run of rows chunk * pin + STRIDE * pos + tap, so a B fragment is ONE ds_read_b128 at
// row (.. + tap) * rowbytes + channel * 2 with tap = k / channels.  K is padded to a multiple of 32 with zero weights; the
// padded k read finite values (the image is zeroed once, then only ever holds finite activations).
#include "rmr_internal.h"
#include "rmr_math.h"

namespace rmr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ int fdiv16(int x, FastDiv d) { return (int)(((float)x + 0.5f) * d.inv); }

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const uint4 a, const uint4 b, const f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool F16>
__device__ __forceinline__ uint2 pack4(const float a, const float b, const float c, const float d) {
    if constexpr (F16) {
        const f16x4 o = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
        return __builtin_bit_cast(uint2, o);
    } else {
        const bf16x4 o = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
        return __builtin_bit_cast(uint2, o);
    }
}

// acc rows are pre-scaled: [0] i, [1] f, [3] o by -log2(e); [2] g by 2 log2(e)  (k_lstm_x16.hip lstm_cell)
__device__ __forceinline__ float cell16(const f32x4 acc, float &c) {
    const float ig = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[0]));
    const float fg = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[1]));
    const float gg = fmaf(-2.0f, fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[2])), 1.0f);
    const float og = fast_rcp(1.0f + __builtin_amdgcn_exp2f(acc[3]));
    c = fmaf(fg, c, ig * gg);
    const float tc = fmaf(-2.0f, fast_rcp(1.0f + __builtin_amdgcn_exp2f(c * 2.8853900817779268f)), 1.0f);
    return og * tc;
}

// =========================================================================================
// convolution + folded BatchNorm + swish
// =========================================================================================
struct ConvS16Args {
    const void *in;       // IN16: 16-bit rows [n][pin][ic]; else fp32 rows (the front kernels' sig2 / seq1)
    uint16_t *out;        // 16-bit [n][pout][out_row]
    const uint4 *apack;   // [oc/16][ks][64 lanes]: 8 consecutive k of row 16 ot + (lane & 15), k = 32 s + 8 (lane >> 4) + j = tap * ic + channel
    const float *bias;
    int64_t n;
    int ic, oc, ks;
    int pin, pout, out_row, out_coff, cb;
    int rb;               // LDS row bytes = 2 ic + 16
    int lds_bytes;
    FastDiv div_pout, div_c8, div_ic;
};

template <int STRIDE, bool F16, int NTV>
__device__ __forceinline__ void conv16_item(const ConvS16Args &a, const unsigned char *smem, int64_t chunk0, int ncols, int ot, int tile0,
                                            int lane, int q, int nn) {
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + 16 * ot + 4 * q);
    f32x4 acc[NTV];
    int roff[NTV], ch[NTV], pp[NTV];
    bool valid[NTV];
#pragma unroll
    for (int t = 0; t < NTV; ++t) {
        int col = (tile0 + t) * 16 + nn;
        valid[t] = col < ncols;
        col = valid[t] ? col : ncols - 1;
        ch[t] = fdiv16(col, a.div_pout);
        pp[t] = col - ch[t] * a.pout;
        roff[t] = (ch[t] * a.pin + pp[t] * STRIDE) * a.rb;
        acc[t] = b4;
    }
    const uint4 *ap = a.apack + (size_t)ot * a.ks * 64 + lane;
    uint4 A0 = ap[0], A1 = ap[(size_t)(a.ks > 1 ? 1 : 0) * 64];
    auto koff = [&](int s) {  // byte offset of the 8-channel group k0 = 32 s + 8 q inside a column's run of rows
        const int k0 = 32 * s + 8 * q, tap = fdiv16(k0, a.div_ic);
        return tap * a.rb + (k0 - tap * a.ic) * 2;
    };
    uint4 x[NTV];
    {
        const int o0 = koff(0);
#pragma unroll
        for (int t = 0; t < NTV; ++t) x[t] = *reinterpret_cast<const uint4 *>(smem + roff[t] + o0);
    }
    for (int s = 0; s < a.ks; ++s) {
        const uint4 A2 = ap[(size_t)(s + 2 < a.ks ? s + 2 : a.ks - 1) * 64];
        const int on = koff(s + 1 < a.ks ? s + 1 : s);
        uint4 y[NTV];
#pragma unroll
        for (int t = 0; t < NTV; ++t) y[t] = *reinterpret_cast<const uint4 *>(smem + roff[t] + on);
#pragma unroll
        for (int t = 0; t < NTV; ++t) acc[t] = mfma16<F16>(A0, x[t], acc[t]);
#pragma unroll
        for (int t = 0; t < NTV; ++t) x[t] = y[t];
        A0 = A1;
        A1 = A2;
    }
#pragma unroll
    for (int t = 0; t < NTV; ++t) {
        if (valid[t]) {
            f32x2 lo = f32x2{acc[t][0], acc[t][1]}, hi = f32x2{acc[t][2], acc[t][3]};
            swish_pk(lo, hi);
            uint16_t *dst = a.out + ((size_t)(chunk0 + ch[t]) * a.pout + pp[t]) * a.out_row + a.out_coff + 16 * ot + 4 * q;
            *reinterpret_cast<uint2 *>(dst) = pack4<F16>(lo.x, lo.y, hi.x, hi.y);
        }
    }
}

template <int STRIDE, bool F16, bool IN16>
__global__ __launch_bounds__(512) void conv_stream16_kernel(ConvS16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    constexpr int NT = 4;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15, nw = nthr >> 6;
    const int C8 = a.ic >> 3, OT = a.oc >> 4;
    for (int i = tid * 16; i < a.lds_bytes; i += nthr * 16) *reinterpret_cast<uint4 *>(smem16 + i) = make_uint4(0, 0, 0, 0);  // finite everywhere
    const int64_t n_iters = (a.n + a.cb - 1) / a.cb;
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        const int64_t chunk0 = it * a.cb;
        const int nch = (int)((a.n - chunk0) < a.cb ? (a.n - chunk0) : a.cb), rows = nch * a.pin, ncols = nch * a.pout;
        RMR_SYNC();  // all reads of the previous iteration (and the zero fill) are done
        {
            const int total = rows * C8;  // 16-byte pieces of 8 channels
            for (int i = tid; i < total; i += nthr) {
                const int row = fdiv16(i, a.div_c8), c8 = i - row * C8;
                uint4 v;
                if constexpr (IN16) {
                    v = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(a.in) + ((size_t)chunk0 * a.pin + row) * a.ic)[c8];
                } else {
                    const float4 *src = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(a.in) + ((size_t)chunk0 * a.pin + row) * a.ic) + 2 * c8;
                    const float4 f0 = src[0], f1 = src[1];
                    const uint2 lo = pack4<F16>(f0.x, f0.y, f0.z, f0.w), hi = pack4<F16>(f1.x, f1.y, f1.z, f1.w);
                    v = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
                *reinterpret_cast<uint4 *>(smem16 + (size_t)row * a.rb + c8 * 16) = v;
            }
        }
        RMR_SYNC();
        const int ntiles = (ncols + 15) >> 4, ntg = (ntiles + NT - 1) / NT;
        for (int item = w; item < OT * ntg; item += nw) {  // (wave-uniform)
            const int tg = item / OT, ot = item - tg * OT;
            const int tile0 = tg * NT, nt = ntiles - tile0 < NT ? ntiles - tile0 : NT;
            if (nt == 4) conv16_item<STRIDE, F16, 4>(a, smem16, chunk0, ncols, ot, tile0, lane, q, nn);
            else if (nt == 3) conv16_item<STRIDE, F16, 3>(a, smem16, chunk0, ncols, ot, tile0, lane, q, nn);
            else if (nt == 2) conv16_item<STRIDE, F16, 2>(a, smem16, chunk0, ncols, ot, tile0, lane, q, nn);
            else conv16_item<STRIDE, F16, 1>(a, smem16, chunk0, ncols, ot, tile0, lane, q, nn);
        }
    }
}

template <int STRIDE, bool F16, bool IN16>
int launch_conv16_t(rmr_engine *e, const ConvLayer &c, const void *in, int pin, uint16_t *out, int out_row, int out_coff, int pout, int64_t n) {
    const int OT = c.oc / 16;
    int nw = OT;
    if (nw > 8) {
        nw = 8;
        for (int d = 8; d >= 4; --d)
            if (OT % d == 0) { nw = d; break; }
    }
    const int rb = c.ic * 2 + 16;
    const size_t row_bytes = (size_t)pin * rb;
    const size_t budget = 65536;  // two blocks per CU
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
    while (cb > 1 && (n + cb - 1) / cb < e->num_cus) cb = (cb + 1) / 2;  // a small batch spread over the CUs
    const size_t lds = ((size_t)cb * pin + c.kw + 2) * rb;  // + the rows a padded k-step and a clamped column may touch
    if (lds > 160 * 1024 - 256)
        RMR_FAIL(RMR_ERR_INVALID, "conv layer %d -> %d channels: one chunk of %d positions needs %zu B of LDS (chunk contexts this long run in "
                                  "fp32 for networks of more than 64 channels)", c.ic, c.oc, pin, lds);
    ConvS16Args a;
    a.in = in; a.out = out; a.apack = reinterpret_cast<const uint4 *>(c.apack16); a.bias = c.bias; a.n = n;
    a.ic = c.ic; a.oc = c.oc; a.ks = (c.kw * c.ic + 31) / 32;
    a.pin = pin; a.pout = pout; a.out_row = out_row; a.out_coff = out_coff; a.cb = cb; a.rb = rb; a.lds_bytes = (int)((lds + 15) & ~(size_t)15);
    a.div_pout = make_fastdiv(pout); a.div_c8 = make_fastdiv(c.ic / 8); a.div_ic = make_fastdiv(c.ic);
    const int64_t iters = (n + cb - 1) / cb;
    int64_t grid = (int64_t)e->num_cus * 4;
    if (grid > iters) grid = iters;
    if (grid < 1) return 0;
    auto kern = conv_stream16_kernel<STRIDE, F16, IN16>;
    RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(kern)));
    ProfScope ps(e, c.kid);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * nw), (size_t)a.lds_bytes, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// `in16`: the input rows are 16-bit (cat); otherwise fp32 (the front kernels' sig2 / seq1, 16 channels)
int launch_conv_stream16(rmr_model *m, const ConvLayer &c, const void *in, bool in16, int pin, uint16_t *out, int out_row, int out_coff, int pout,
                         int64_t n) {
    rmr_engine *e = m->eng;
    if (!c.apack16 || c.ic % 16 || c.oc % 16) RMR_FAIL(RMR_ERR_INVALID, "internal: layer %d -> %d not packed for the streamed 16-bit kernel", c.ic, c.oc);
    if (c.stride == 3 && !in16) {
        return m->f16 ? launch_conv16_t<3, true, false>(e, c, in, pin, out, out_row, out_coff, pout, n)
                      : launch_conv16_t<3, false, false>(e, c, in, pin, out, out_row, out_coff, pout, n);
    }
    if (c.stride == 1 && in16) {
        return m->f16 ? launch_conv16_t<1, true, true>(e, c, in, pin, out, out_row, out_coff, pout, n)
                      : launch_conv16_t<1, false, true>(e, c, in, pin, out, out_row, out_coff, pout, n);
    }
    RMR_FAIL(RMR_ERR_INVALID, "no streamed 16-bit conv kernel for stride %d, %s input", c.stride, in16 ? "16-bit" : "fp32");
}

// =========================================================================================
// lstm1 (T steps) + lstm2 (ONE step, see k_lstm.hip) + fc
// =========================================================================================
namespace {

struct LstmS16Args {
    const uint16_t *x;  // 16-bit [n][T][H]
    float *logits;
    // a_ih / a_hh / a_ih2: [H/16 waves][4 tiles][ksh][64 lanes] x 16 B; row m of tile t of wave w = (unit 16 w + 4 (m >> 2) + t, gate m & 3),
    // rows pre-scaled (engine.hip lstm1_gate_scale), lstm2 with a zero f row; b1 / b2: [H/16][4 tiles][4 q][4 gates]
    const uint4 *a_ih, *a_hh, *a_ih2;
    const float *b1, *b2, *w_fc, *b_fc;
    int64_t n;
    int T, num_out, H, ksh, rb, lds_bytes;
    FastDiv div_c8;
};

// acc[tile][nt] += A (streamed, one k-step ahead) x B fragments from one LDS image
template <int NT, bool F16>
__device__ __forceinline__ void mm16(const unsigned char *img, int rb, int ksh, const uint4 *apack, int w, int lane, int q, int nn, f32x4 (&acc)[4][NT]) {
    const uint4 *ap = apack + (size_t)w * 4 * ksh * 64 + lane;
    const unsigned char *b = img + (size_t)nn * rb + 16 * q;
    uint4 A[4], An[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) A[t] = ap[(size_t)(t * ksh) * 64];
    for (int s = 0; s < ksh; ++s) {
        const int sn = s + 1 < ksh ? s + 1 : s;
#pragma unroll
        for (int t = 0; t < 4; ++t) An[t] = ap[(size_t)(t * ksh + sn) * 64];
        uint4 bx[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) bx[c] = *reinterpret_cast<const uint4 *>(b + (size_t)c * 16 * rb + 64 * s);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[t][c] = mfma16<F16>(A[t], bx[c], acc[t][c]);
#pragma unroll
        for (int t = 0; t < 4; ++t) A[t] = An[t];
    }
}

// One block = H/16 waves x 16 NT chunks.  Wave w owns hidden units 16 w .. 16 w + 15 as four unit-major tiles: a lane's D fragment of
// tile t is the four gates of unit 16 w + 4 q + t for one chunk, so the cell update is lane-local and the lane's four units leave as
// one 8-byte store.
template <int NT, bool F16, int MAXT>
__global__ __launch_bounds__(MAXT) void lstm_stream16_kernel(LstmS16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    const int H = a.H, rb = a.rb, C8 = H >> 3;
    constexpr int ROWS = 16 * NT;
    const int img = ROWS * rb;
    unsigned char *xbuf = smem16, *hbuf = smem16 + 2 * img;
    float *part = reinterpret_cast<float *>(hbuf);  // [H/16][ROWS][16] floats, once the recurrence is over: H * ROWS * 4 B <= 2 img
    const int tid = threadIdx.x, nthr = blockDim.x;  // nthr == 4 H
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15, NW = nthr >> 6;
    for (int i = tid * 16; i < a.lds_bytes; i += nthr * 16) *reinterpret_cast<uint4 *>(smem16 + i) = make_uint4(0, 0, 0, 0);  // k >= H reads zeros
    constexpr int PIECES = NT / 2;  // 16-byte pieces of the block's ROWS x H tile per thread (ROWS * H / 8 over 4 H threads)
    int st_row[PIECES], st_c8[PIECES];
#pragma unroll
    for (int u = 0; u < PIECES; ++u) {
        const int i = tid + u * nthr;
        st_row[u] = fdiv16(i, a.div_c8);
        st_c8[u] = i - st_row[u] * C8;
    }
    f32x4 bias[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bias[t] = *reinterpret_cast<const f32x4 *>(a.b1 + ((w * 4 + t) * 4 + q) * 4);
    const int64_t n_groups = (a.n + ROWS - 1) / ROWS;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t chunk0 = grp * ROWS;
        const uint4 *xsrc[PIECES];
#pragma unroll
        for (int u = 0; u < PIECES; ++u) {
            int64_t chn = chunk0 + st_row[u];
            if (chn >= a.n) chn = a.n - 1;  // clamp the ragged tail (results masked)
            xsrc[u] = reinterpret_cast<const uint4 *>(a.x + (size_t)chn * a.T * H) + st_c8[u];
        }
        RMR_SYNC();  // the previous group's LDS traffic (and the zero fill) is done
#pragma unroll
        for (int u = 0; u < PIECES; ++u) *reinterpret_cast<uint4 *>(xbuf + (size_t)st_row[u] * rb + st_c8[u] * 16) = xsrc[u][0];
        RMR_SYNC();
        float cst[4][NT];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < NT; ++c) cst[t][c] = 0.0f;
        for (int t = 0; t < a.T; ++t) {
            uint4 xn[PIECES];
            const int tf = t + 1 < a.T ? t + 1 : t;
#pragma unroll
            for (int u = 0; u < PIECES; ++u) xn[u] = xsrc[u][(size_t)tf * C8];
            f32x4 acc[4][NT];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int c = 0; c < NT; ++c) acc[tt][c] = bias[tt];
            mm16<NT, F16>(xbuf + (size_t)(t & 1) * img, rb, a.ksh, a.a_ih, w, lane, q, nn, acc);
            if (t > 0) mm16<NT, F16>(hbuf + (size_t)((t - 1) & 1) * img, rb, a.ksh, a.a_hh, w, lane, q, nn, acc);
            const bool last = t + 1 == a.T;
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                float h[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    h[tt] = cell16(acc[tt][c], cst[tt][c]);
                    if (last) h[tt] = swish_f(h[tt]);  // lstm2 reads swish(h1[T-1]) (models/ConvLSTM_w_ref.py:52-53)
                }
                *reinterpret_cast<uint2 *>(hbuf + (size_t)(t & 1) * img + (size_t)(c * 16 + nn) * rb + (16 * w + 4 * q) * 2) = pack4<F16>(h[0], h[1], h[2], h[3]);
            }
            if (!last) {  // xbuf[(t + 1) & 1] was last read in step t - 1, a barrier ago
#pragma unroll
                for (int u = 0; u < PIECES; ++u) *reinterpret_cast<uint4 *>(xbuf + (size_t)((t + 1) & 1) * img + (size_t)st_row[u] * rb + st_c8[u] * 16) = xn[u];
            }
            RMR_SYNC();
        }
        // ---- lstm2: one step on swish(h1[T-1]) (zero f row: c0 = 0) ----
        f32x4 acc2[4][NT];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.b2 + ((w * 4 + tt) * 4 + q) * 4);
#pragma unroll
            for (int c = 0; c < NT; ++c) acc2[tt][c] = b4;
        }
        mm16<NT, F16>(hbuf + (size_t)((a.T - 1) & 1) * img, rb, a.ksh, a.a_ih2, w, lane, q, nn, acc2);
        f32x4 y[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                float c2 = 0.0f;
                y[c][tt] = swish_f(cell16(acc2[tt][c], c2));
            }
        RMR_SYNC();  // every wave has read h1[T-1]: its image becomes `part`
        for (int o = 0; o < a.num_out; ++o) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(a.w_fc + (size_t)o * H + 16 * w + 4 * q);
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                float p = fmaf(wv[3], y[c][3], fmaf(wv[2], y[c][2], fmaf(wv[1], y[c][1], wv[0] * y[c][0])));
                p += __shfl_xor(p, 16);
                p += __shfl_xor(p, 32);
                if (q == 0) part[((size_t)w * ROWS + c * 16 + nn) * 16 + o] = p;
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
        // (`part` only overwrote bytes that the next group's h stores rewrite before they are read - every unit of every row - and the 16
        //  padding bytes of a row are never read; H is a multiple of 32 here, so no k-step reads past a row's units)
    }
}

template <int NT, bool F16, int MAXT>
int launch_lstm16_t(rmr_model *m, LstmS16Args &a, int64_t n) {
    rmr_engine *e = m->eng;
    const size_t lds = (size_t)4 * 16 * NT * a.rb;
    if (lds > 160 * 1024 - 256) RMR_FAIL(RMR_ERR_INVALID, "streamed 16-bit LSTM: %zu B of LDS for %d hidden units", lds, a.H);
    if ((size_t)a.H * 16 * NT * 4 > lds / 2) RMR_FAIL(RMR_ERR_INVALID, "internal: fc partial sums do not fit the h images");
    a.lds_bytes = (int)lds;
    const int64_t groups = (n + 16 * NT - 1) / (16 * NT);
    int64_t grid = (int64_t)e->num_cus * 4;
    if (grid > groups) grid = groups;
    if (grid < 1) return 0;
    auto kern = lstm_stream16_kernel<NT, F16, MAXT>;
    RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(kern)));
    ProfScope ps(e, K_LSTM_HEAD);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(4 * a.H), lds, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

}  // namespace

int launch_lstm_stream16(rmr_model *m, const uint16_t *x, int64_t n, float *logits) {
    const int H = m->desc.size;
    if (H % 32 || H > 256 || !m->lstm.s16_ih) RMR_FAIL(RMR_ERR_INVALID, "internal: LSTM of %d units not packed for the streamed 16-bit kernel", H);
    LstmS16Args a;
    a.x = x; a.logits = logits; a.n = n; a.T = m->T; a.num_out = m->desc.num_out; a.H = H;
    a.ksh = H / 32; a.rb = H * 2 + 16; a.div_c8 = make_fastdiv(H / 8);
    a.a_ih = reinterpret_cast<const uint4 *>(m->lstm.s16_ih); a.a_hh = reinterpret_cast<const uint4 *>(m->lstm.s16_hh);
    a.a_ih2 = reinterpret_cast<const uint4 *>(m->lstm.s16_ih2);
    a.b1 = m->lstm.s16_b1; a.b2 = m->lstm.s16_b2; a.w_fc = m->lstm.w_fc; a.b_fc = m->lstm.b_fc;
    if (H <= 128) return m->f16 ? launch_lstm16_t<4, true, 512>(m, a, n) : launch_lstm16_t<4, false, 512>(m, a, n);
    return m->f16 ? launch_lstm16_t<2, true, 1024>(m, a, n) : launch_lstm16_t<2, false, 1024>(m, a, n);
}

}  // namespace rmr
