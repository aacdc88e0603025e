// k_wino.hip — the 5-tap stride-1 convolutions between `size`-wide layers (merge_conv1 of both networks, Conv_w_ref's
// merge_conv2) + folded BatchNorm + swish in fp32 as a Winograd / Toom-Cook F(2, 5) minimal-filtering convolution on the fp32
// matrix cores.
//
// Replaces, at 64 output channels: models/ConvLSTM_w_ref.py:36-37,50 (merge_conv1 / merge_bn), models/Conv_w_ref.py:35-38,54-55
// (merge_conv1 / merge_conv2) - the layers k_conv.hip's conv_mfma<128,5,1> / <64,5,1> compute in direct form
// (RMR_WINOGRAD=0 keeps those: the comparand of tests/test_gpu_wino.py).
//
// Two neighbouring output positions of one channel need 2 x 5 = 10 multiplications per input channel in direct form and
// 2 + 5 - 1 = 6 in the minimal form (Toom-Cook at the points 0, 1, -1, 2, -2, inf):
//
//     y[2t + i][oc] = sum_x AT[i][x] * ( sum_ic U[x][oc][ic] * V[x][t][ic] ),     i = 0, 1,   x = 0..5
//     U[x][oc][ic]  = sum_tap G[x][tap] * W[oc][ic][tap]          (host, float64, one rounding: engine.hip pack_conv_wino)
//     V[x][t][ic]   = sum_j  BT[x][j]  * in[2t + j][ic]           (this kernel, exact small-integer coefficients)
//
// i.e. SIX independent GEMMs with K = IC over columns t = (chunk, position pair) instead of one GEMM with K = 5 IC over
// single positions: 0.6 of the direct form's MFMAs (576 against 960 v_mfma_f32_16x16x4_f32 per chunk at C100) for 14 extra
// VALU operations per (pair, channel) and 8 per (pair, output channel).  The transforms are exact in real arithmetic; in
// fp32 the result differs from the direct form's by rounding only - about twice the direct form's own distance to float64
// (oracle/winograd_f25.py: 1.3e-6 against 5.9e-7 on unit-scale activations; logits of the golden models <= 1e-4 as before).
//
// Structure: one block of EIGHT waves per CU, two per SIMD.  Wave (w, h) = output channels 16 w .. 16 w + 15, x in {3 h, 3 h + 1,
// 3 h + 2}: its weight slice is 3 IC / 4 registers (96 at IC = 128), which leaves room for two waves per SIMD - one wave alone
// cannot cover its own LDS latency (first form of this kernel: one wave per SIMD with all six x, 192 + 70 AGPR-parked weight
// registers, 12.4 ns per chunk).  An iteration = NC columns (position pairs, flattened over the chunks; 16 at IC = 128, 32 at
// IC = 64).  V lives in LDS twice (double buffer, 2 x 55 KB at IC = 128) in k_conv.hip's four-plane layout per x (plane q =
// channels {16 g + 4 q + j}: one ds_read_b128 feeds four MFMAs, conflict-free).  Per iteration:
//   1. the rows of iteration i + 1 are already in registers (global loads issued an iteration ahead): transform, write V[next]
//   2. issue the global loads of iteration i + 2
//   3. 3 IC / 4 MFMAs per column tile on V[cur]: three independent accumulator chains, B fragments fetched a step ahead
//   4. h = 1 waves: their half of AT m (two 16-byte partial sums per lane) into the exchange buffer (double-buffered too)
//   5. ONE barrier (V[next] complete, V[cur] free, partial sums visible)
//   6. h = 0 waves: the other half of AT m + the partner's sums, swish, two 16-byte channel-last stores per column
// Columns are tracked incrementally (chunk, pair) - one 64-bit division per thread and launch, none per iteration.
#include "rmr_internal.h"
#include "rmr_math.h"

namespace rmr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WinoArgs {
    const float *in;
    float *out;
    const float *wpack;  // [oc/16][6 * IC / 4][64 lanes]
    const float *bias;
    int64_t ncols;       // n * tpc
    int pin, pout, tpc;  // input rows / output positions / position pairs per chunk
    int out_row, out_coff;
    int abl;             // experiment build (make abl, RMR_WINO_ABLATE): 1 no transform / fetch, 2 no MFMAs, 4 no output stage
};

// BT' d for four channels at once.  Rows 1 and 2 are computed NEGATED (a' = d4 - 4 d2, b' = d3 - 4 d1: v1' = a' + b' = -v1,
// v2' = a' - b' = -v2) and the sign sits in the packed filter rows U1, U2 instead (engine.hip pack_conv): every term is one
// multiply-add with a literal, no negation instructions; negation is exact, the products are the same.
__device__ __forceinline__ void wino_in_transform(const f32x4 (&d)[6], f32x4 (&v)[6]) {
    const f32x4 c4 = {4.0f, 4.0f, 4.0f, 4.0f}, m4 = -c4, m5 = {-5.0f, -5.0f, -5.0f, -5.0f}, c2 = {2.0f, 2.0f, 2.0f, 2.0f}, m1 = {-1.0f, -1.0f, -1.0f, -1.0f};
    v[0] = __builtin_elementwise_fma(m5, d[2], __builtin_elementwise_fma(c4, d[0], d[4]));  // 4 d0 - 5 d2 + d4
    const f32x4 a = __builtin_elementwise_fma(m4, d[2], d[4]);                               // d4 - 4 d2
    const f32x4 b = __builtin_elementwise_fma(m4, d[1], d[3]);                               // d3 - 4 d1
    v[1] = a + b;
    v[2] = __builtin_elementwise_fma(m1, b, a);
    const f32x4 c = __builtin_elementwise_fma(m1, d[2], d[4]), e = __builtin_elementwise_fma(m1, d[1], d[3]);
    v[3] = __builtin_elementwise_fma(c2, e, c);                                              // (d4 - d2) + 2 (d3 - d1)
    v[4] = __builtin_elementwise_fma(-c2, e, c);
    v[5] = __builtin_elementwise_fma(m5, d[3], __builtin_elementwise_fma(c4, d[1], d[5]));  // 4 d1 - 5 d3 + d5
}

// a column (position pair t of a chunk) as its index, its first row in a [chunk][P rows] tensor and t, advanced by a fixed
// number of columns without dividing (P = pin for the rows fetched, pout for the rows stored)
struct WinoCol {
    int64_t col, row;
    int t;
    __device__ __forceinline__ void init(int64_t c, int tpc, int P) {
        const int64_t ch = c / tpc;
        col = c; t = (int)(c - ch * tpc); row = ch * P + 2 * t;
    }
    // dcol columns on: drow = (dcol / tpc) * P + 2 * (dcol % tpc), dr = dcol % tpc, wrap = P - 2 * tpc
    __device__ __forceinline__ void advance(int64_t dcol, int64_t drow, int dr, int tpc, int wrap) {
        col += dcol; row += drow; t += dr;
        if (t >= tpc) { t -= tpc; row += wrap; }
    }
};

template <int IC>
__global__ __launch_bounds__(512, 2) void wino_conv_kernel(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NC = IC == 128 ? 16 : 32;                 // columns per iteration
    constexpr int CT = NC / 16;                             // column tiles per wave
    constexpr int G = IC / 16;
    constexpr int RS = (G % 2 == 0) ? IC / 4 + 4 : IC / 4;  // floats per column per plane, RS / 4 odd
    constexpr int PL = NC * RS;                             // plane stride (multiple of 64 floats)
    constexpr int XI = 4 * PL;                              // one x image
    constexpr int BUF = 6 * XI;                             // one V buffer
    constexpr int PBUF = CT * 2 * 4 * 64 * 4;               // one exchange buffer: [ct][2 sums][4 w][64 lanes] x 16 B
    constexpr int S = 3 * IC / 4;                           // MFMA k-steps of a wave
    constexpr int Q4 = IC / 4;                              // channel quads per row
    static_assert(PL % 64 == 0 && NC * Q4 == 512, "layout: one (column, quad) transform item per thread");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, w = wv & 3, h = wv >> 2, q = lane >> 4, nn = lane & 15;
    float *pex = smem + 2 * BUF;

    float A[S];
    {
        const float *ap = a.wpack + ((size_t)w * 2 * S + (size_t)h * S) * 64 + lane;  // x = 3 h .. 3 h + 2 are consecutive
#pragma unroll
        for (int s = 0; s < S; ++s) A[s] = ap[(size_t)s * 64];
    }
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + 16 * w + 4 * q);
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};

    const int64_t n_iters = (a.ncols + NC - 1) / NC;
    const int64_t dcol = (int64_t)NC * gridDim.x, dq = dcol / a.tpc;
    const int dr = (int)(dcol - dq * a.tpc);
    const int64_t drow_in = dq * a.pin + 2 * dr, drow_out = dq * a.pout + 2 * dr;
    const int wrap_in = a.pin - 2 * a.tpc, wrap_out = a.pout - 2 * a.tpc;
    const int q16 = 16 / a.tpc, r16 = 16 - q16 * a.tpc;  // from column tile 0 to tile 1 (CT == 2)
    // This thread's transform item: column cl of the iteration, channels 16 g + 4 qq .. + 3.  The 16 lanes of a ds_write_b128
    // service group share the plane (qq = lane >> 4) and differ in (g, column): 16-byte bank slots g + (RS / 4) * column, distinct
    // but for a pair or three (lanes that differ in the plane only would all fall on one slot: the plane stride is a multiple of 256 B
    // for the readers' sake).  A wave's loads still cover whole 512-byte rows (its four groups = the four planes).
    constexpr int CPW = 16 / G;  // columns per wave
    const int qq = lane >> 4, tg = (lane & 15) % G, cl = CPW * wv + (lane & 15) / G;
    float *const vdst = smem + qq * PL + cl * RS + 4 * tg;
    const float *const src0 = a.in + 16 * tg + 4 * qq;
    const int64_t last_row = (a.ncols / a.tpc) * a.pin - 1;  // of the whole input
    WinoCol fc;  // the column this thread fetches next
    fc.init((int64_t)blockIdx.x * NC + cl, a.tpc, a.pin);
    f32x4 d[6];
    // the six input rows of the item.  A column beyond the end repeats the last one (finite data, results never stored).  Row 5 of the
    // odd last pair of an odd pout lies behind its chunk: the next chunk's first row (finite; it only reaches the y1 that is not
    // stored) - clamped where that would be behind the whole input.
    auto fetch = [&]() {
        const bool valid = fc.col < a.ncols;
        const int64_t row = valid ? fc.row : last_row - 5;
        const float *src = src0 + (size_t)row * IC;
#pragma unroll
        for (int j = 0; j < 5; ++j) d[j] = *reinterpret_cast<const f32x4 *>(src + j * IC);
        d[5] = *reinterpret_cast<const f32x4 *>(src0 + (size_t)(row + 5 < last_row ? row + 5 : last_row) * IC);
        fc.advance(dcol, drow_in, dr, a.tpc, wrap_in);
    };
    auto transform_to = [&](int buf) {
        f32x4 v[6];
        wino_in_transform(d, v);
#pragma unroll
        for (int x = 0; x < 6; ++x) *reinterpret_cast<f32x4 *>(vdst + buf * BUF + x * XI) = v[x];
    };

    int64_t it = blockIdx.x;
    if (it >= n_iters) return;
    WinoCol oc;  // the first column this lane finishes (h == 0 waves): column nn of tile 0
    oc.init(it * NC + nn, a.tpc, a.pout);
    float *const dst0 = a.out + a.out_coff + 16 * w + 4 * q;
    fetch();
    transform_to(0);
    if (it + gridDim.x < n_iters) fetch();
    RMR_SYNC();
    int cur = 0;
    for (; it < n_iters; it += gridDim.x) {
        const int64_t nxt = it + gridDim.x;
#ifdef RMR_TIMING_ABLATIONS
        if (!(a.abl & 1))
#endif
        if (nxt < n_iters) {
            transform_to(cur ^ 1);
            if (nxt + gridDim.x < n_iters) fetch();
        }
        // ---- three GEMMs of K = IC on this wave's 16 channels x NC columns
        f32x4 acc[CT][3];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            acc[ct][0] = h == 0 ? b4 : zero;  // AT[0][0] = 1: the bias enters y0 through m0 ...
            acc[ct][1] = zero;
            acc[ct][2] = h == 1 ? b4 : zero;  // ... and y1 through m5 (AT[1][5] = 1)
        }
        const float *r = smem + cur * BUF + 3 * h * XI + q * PL + nn * RS;
        constexpr int NS = G * CT;  // steps: (g, column tile)
        f32x4 x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = *reinterpret_cast<const f32x4 *>(r + k * XI);
#ifdef RMR_TIMING_ABLATIONS
        if (!(a.abl & 2))
#endif
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int g = st / CT, ct = st % CT;
            f32x4 y[3];
            if (st + 1 < NS) {
                const int g1 = (st + 1) / CT, ct1 = (st + 1) % CT;
#pragma unroll
                for (int k = 0; k < 3; ++k) y[k] = *reinterpret_cast<const f32x4 *>(r + k * XI + ct1 * 16 * RS + 4 * g1);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[ct][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(k * G + g) * 4 + j], x[k][j], acc[ct][k], 0, 0, 0);
            if (st + 1 < NS) {
#pragma unroll
                for (int k = 0; k < 3; ++k) x[k] = y[k];
            }
        }
        // pin the software pipeline: the reads of step st + 1 are issued before the MFMAs of step st
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            if (st + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        }
        // ---- AT m: y0 = m0 + m1 + m2 + (m3 + m4),  y1 = m1 - m2 + (2 (m3 - m4) + m5)
        float *pb = pex + cur * PBUF + (w * 64 + lane) * 4;
#ifdef RMR_TIMING_ABLATIONS
        if ((a.abl & 4) && acc[0][0][0] != 123456.75f) { RMR_SYNC(); cur ^= 1; continue; }
#endif
        if (h == 1) {
            const f32x4 two = {2.0f, 2.0f, 2.0f, 2.0f};
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                *reinterpret_cast<f32x4 *>(pb + (ct * 2 + 0) * 1024) = acc[ct][0] + acc[ct][1];
                *reinterpret_cast<f32x4 *>(pb + (ct * 2 + 1) * 1024) = __builtin_elementwise_fma(two, acc[ct][0] - acc[ct][1], acc[ct][2]);
            }
        }
        RMR_SYNC();  // V[next] complete, V[cur] free, the partial sums of this iteration visible
        if (h == 0) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                WinoCol c = oc;
                if (ct > 0) c.advance(16, (int64_t)q16 * a.pout + 2 * r16, r16, a.tpc, wrap_out);
                if (c.col < a.ncols) {
                    const f32x4 p0 = *reinterpret_cast<const f32x4 *>(pb + (ct * 2 + 0) * 1024);
                    const f32x4 p1 = *reinterpret_cast<const f32x4 *>(pb + (ct * 2 + 1) * 1024);
                    const f32x4 y0 = ((acc[ct][0] + acc[ct][1]) + acc[ct][2]) + p0;
                    const f32x4 y1 = (acc[ct][1] - acc[ct][2]) + p1;
                    float *dst = dst0 + (size_t)c.row * a.out_row;
                    f32x2 lo = f32x2{y0[0], y0[1]}, hi = f32x2{y0[2], y0[3]};
                    swish_pk(lo, hi);
                    *reinterpret_cast<f32x4 *>(dst) = f32x4{lo.x, lo.y, hi.x, hi.y};
                    if (2 * c.t + 1 < a.pout) {
                        lo = f32x2{y1[0], y1[1]}; hi = f32x2{y1[2], y1[3]};
                        swish_pk(lo, hi);
                        *reinterpret_cast<f32x4 *>(dst + a.out_row) = f32x4{lo.x, lo.y, hi.x, hi.y};
                    }
                }
            }
            oc.advance(dcol, drow_out, dr, a.tpc, wrap_out);
        }
        cur ^= 1;
    }
}

template <int IC>
static int launch_wino_t(rmr_engine *e, const ConvLayer &c, const float *in, int pin, float *out, int out_row, int out_coff, int pout,
                         int64_t n) {
    constexpr int NC = IC == 128 ? 16 : 32, CT = NC / 16;
    constexpr int G = IC / 16;
    constexpr int RS = (G % 2 == 0) ? IC / 4 + 4 : IC / 4;
    const size_t lds = ((size_t)2 * 6 * 4 * NC * RS + (size_t)2 * CT * 2 * 4 * 64 * 4) * sizeof(float);
    WinoArgs a;
    a.in = in; a.out = out; a.wpack = c.wpack; a.bias = c.bias;
    a.pin = pin; a.pout = pout; a.tpc = (pout + 1) / 2;
    a.ncols = n * a.tpc;
    a.out_row = out_row; a.out_coff = out_coff;
    a.abl = abl_int("RMR_WINO_ABLATE", 0);  // ignored unless built with -DRMR_TIMING_ABLATIONS
    const int64_t iters = (a.ncols + NC - 1) / NC;
    int64_t grid = (int64_t)e->num_cus;  // one persistent block per CU (its registers and LDS fill the CU)
    if (grid > iters) grid = iters;
    if (grid < 1) return 0;
    auto kern = wino_conv_kernel<IC>;
    RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(kern)));
    ProfScope ps(e, c.kid);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

bool conv_wino_supported(const ConvLayer &c, int pin, int pout) {
    return c.wpack && c.kw == 5 && c.stride == 1 && c.oc == 64 && (c.ic == 128 || c.ic == 64) && pin == pout + 4 && pout >= 2;
}

int launch_conv_wino(rmr_engine *e, const ConvLayer &c, const float *in, int in_row, int pin, float *out, int out_row, int out_coff,
                     int pout, int64_t n) {
    if (in_row != c.ic) RMR_FAIL(RMR_ERR_INVALID, "conv input row %d != ic %d", in_row, c.ic);
    if (c.ic == 128) return launch_wino_t<128>(e, c, in, pin, out, out_row, out_coff, pout, n);
    if (c.ic == 64) return launch_wino_t<64>(e, c, in, pin, out, out_row, out_coff, pout, n);
    RMR_FAIL(RMR_ERR_INVALID, "no Winograd kernel for ic=%d", c.ic);
}

}  // namespace rmr
