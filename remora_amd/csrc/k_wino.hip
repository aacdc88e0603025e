// k_wino.hip — the 5-tap stride-1 convolutions between `size`-wide layers (merge_conv1 of both networks, Conv_w_ref's
// merge_conv2) + folded BatchNorm + swish in fp32 as a Winograd / Toom-Cook F(4, 5) minimal-filtering convolution on the fp32
// matrix cores.
//
// Replaces, at 64 output channels: models/ConvLSTM_w_ref.py:36-37,50 (merge_conv1 / merge_bn), models/Conv_w_ref.py:35-38,54-55
// (merge_conv1 / merge_conv2) - the layers k_conv.hip's conv_mfma<128,5,1> / <64,5,1> compute in direct form
// (RMR_WINOGRAD=0 keeps those: the comparand of tests/test_gpu_wino.py).  The second kernel of this file takes Conv_w_ref's
// stride-3 seq_conv3 (models/Conv_w_ref.py:31-32,51; direct form: conv_mfma<32,9,3>) in polyphase F(4, 3) form.
//
// Four neighbouring output positions of one channel need 4 x 5 = 20 multiplications per input channel in direct form and
// 4 + 5 - 1 = 8 in the minimal form (Toom-Cook at the points 0, 1, -1, 2, -2, 1/2, -1/2, inf; oracle/winograd.py derives the
// three matrices from the points in exact rational arithmetic and a CPU test holds the constants below to them):
//
//     y[4t + i][oc] = sum_x AT[i][x] * ( sum_ic U[x][oc][ic] * V[x][t][ic] ),     i = 0..3,   x = 0..7
//     U[x][oc][ic]  = sum_tap G[x][tap] * W[oc][ic][tap]          (host, float64, one rounding: engine.hip pack_conv)
//     V[x][t][ic]   = sum_j  BT[x][j]  * in[4t + j][ic]           (this kernel; small-integer coefficients)
//
//     BT =  4   0 -21   0  21   0  -4   0        G = 1/4    0      0     0     0        AT = 1  1  1  1  1   1    1   0
//           0  -4  -4  17  17  -4  -4   0            1/18   1/18   1/18  1/18  1/18          0  1 -1  2 -2  1/2 -1/2  0
//           0   4  -4 -17  17   4  -4   0            1/18  -1/18   1/18 -1/18  1/18          0  1  1  4  4  1/4  1/4  0
//           0   2   1 -10  -5   8   4   0            1/360  1/180  1/90  1/45  2/45          0  1 -1  8 -8  1/8 -1/8  1
//           0  -2   1  10  -5  -8   4   0            1/360 -1/180  1/90 -1/45  2/45
//           0   4   8  -5 -10   1   2   0            16/45  8/45   4/45  2/45  1/45
//           0  -4   8   5 -10  -1   2   0            16/45 -8/45   4/45 -2/45  1/45
//           0  -4   0  21   0 -21   0   4            0      0      0     0     1/4
//
// i.e. EIGHT independent GEMMs with K = IC over columns t = (chunk, group of four positions) instead of one GEMM with K = 5 IC
// over single positions: 0.4 of the direct form's MFMAs (384 against 960 v_mfma_f32_16x16x4_f32 per chunk at C100) for 28 extra
// VALU operations per (group, input channel) and 18 per (group, output channel).  The transforms are exact in real arithmetic;
// in fp32 the result differs from the direct form's by rounding only - about 1.5-2 x the direct form's own distance to float64
// (the first form of this file, F(2, 5) at the points 0, +-1, +-2, inf with 0.6 of the MFMAs, measured the same error level
// and 2.5 ms per 250 k chunks; profiles/NOTES_r06.md section 8).
//
// Structure: one block of EIGHT waves per CU, two per SIMD.  Wave (w, h) = output channels 16 w .. 16 w + 15 and four of the
// eight x (h = 0: the points +-1, +-2; h = 1: 0, +-1/2, inf): its weight slice is 4 IC / 4 registers (128 at IC = 128), which
// leaves room for two waves per SIMD - one wave alone cannot cover its own LDS latency.  An iteration = 16 columns.  V lives in
// LDS twice (double buffer, 2 x 64 KB at IC = 128): per x four planes (plane q = channels {16 g + 4 q + j}: one ds_read_b128
// feeds four MFMAs), rows of IC / 4 floats per column WITHOUT padding, the 16-byte slots of a row XOR-swizzled by the column so
// that the 16 columns of a tile read - and the 16 lanes of a transform group write - sixteen distinct bank slots.  Per iteration:
//   1. the rows of iteration i + 1 are already in registers (global loads issued an iteration ahead): transform, write V[next]
//   2. issue the global loads of iteration i + 2
//   3. 4 IC / 4 MFMAs on V[cur]: four independent accumulator chains, B fragments fetched a step ahead
//   4. h = 1 waves: their half of AT m (four 16-byte partial sums per lane, bias included) into the exchange buffer (double-buffered too)
//   5. ONE barrier (V[next] complete, V[cur] free, partial sums visible)
//   6. h = 0 waves: the other half of AT m + the partner's sums, swish, four 16-byte channel-last stores per column
// Columns are tracked incrementally (row of the group's first position) - one 64-bit division per thread and launch.
#include "rmr_internal.h"
#include "rmr_math.h"

namespace rmr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WinoArgs {
    const float *in;
    float *out;
    const float *wpack;  // [oc/16][8 * IC / 4][64 lanes], x in the order 1, 2, 3, 4 | 0, 5, 6, 7 (the two wave halves)
    const float *bias;
    int64_t ncols;       // n * tpc
    int pin, pout, tpc;  // input rows / output positions / groups of four positions per chunk
    int out_row, out_coff;
    int abl;             // experiment build (make abl, RMR_WINO_ABLATE): 1 no transform / fetch, 2 no MFMAs, 4 no output stage
};

__device__ __forceinline__ f32x4 splat4(float x) { return f32x4{x, x, x, x}; }
__device__ __forceinline__ f32x4 fma4(float c, f32x4 a, f32x4 b) { return __builtin_elementwise_fma(splat4(c), a, b); }

// BT d for four channels at once; v in the kernel's x order (1, 2, 3, 4, 0, 5, 6, 7).  Even / odd parts of the +- point pairs are
// shared: 28 multiply-adds for the eight rows.
__device__ __forceinline__ void wino_in_transform(const f32x4 (&d)[8], f32x4 (&v)[8]) {
    // +-1:  e = -4 d2 + 17 d4 - 4 d6,  o = -4 d1 + 17 d3 - 4 d5
    const f32x4 e1 = fma4(-4.0f, d[2] + d[6], splat4(17.0f) * d[4]);
    const f32x4 o1 = fma4(-4.0f, d[1] + d[5], splat4(17.0f) * d[3]);
    v[0] = e1 + o1;
    v[1] = e1 - o1;
    // +-2:  e = d2 - 5 d4 + 4 d6,  o = 2 (d1 - 5 d3 + 4 d5)
    const f32x4 e2 = fma4(-5.0f, d[4], fma4(4.0f, d[6], d[2]));
    const f32x4 o2 = fma4(-5.0f, d[3], fma4(4.0f, d[5], d[1]));
    v[2] = fma4(2.0f, o2, e2);
    v[3] = fma4(-2.0f, o2, e2);
    // 0:  4 (d0 - d6) - 21 (d2 - d4)
    v[4] = fma4(-21.0f, d[2] - d[4], splat4(4.0f) * (d[0] - d[6]));
    // +-1/2:  e = 2 (4 d2 - 5 d4 + d6),  o = 4 d1 - 5 d3 + d5
    const f32x4 e3 = fma4(-5.0f, d[4], fma4(4.0f, d[2], d[6]));
    const f32x4 o3 = fma4(-5.0f, d[3], fma4(4.0f, d[1], d[5]));
    v[5] = fma4(2.0f, e3, o3);
    v[6] = fma4(2.0f, e3, -o3);
    // inf:  4 (d7 - d1) + 21 (d3 - d5)
    v[7] = fma4(21.0f, d[3] - d[5], splat4(4.0f) * (d[7] - d[1]));
}

// a column (group t of four positions of a chunk) as its index, the row of its first position in a [chunk][P rows] tensor and t,
// advanced by a fixed number of columns without dividing (P = pin for the rows fetched, pout for the rows stored)
struct WinoCol {
    int64_t col, row;
    int t;
    __device__ __forceinline__ void init(int64_t c, int tpc, int P) {
        const int64_t ch = c / tpc;
        col = c; t = (int)(c - ch * tpc); row = ch * P + 4 * t;
    }
    // dcol columns on: drow = (dcol / tpc) * P + 4 * (dcol % tpc), dr = dcol % tpc, wrap = P - 4 * tpc
    __device__ __forceinline__ void advance(int64_t dcol, int64_t drow, int dr, int tpc, int wrap) {
        col += dcol; row += drow; t += dr;
        if (t >= tpc) { t -= tpc; row += wrap; }
    }
};

constexpr int WINO_NC = 16;  // columns per iteration

template <int IC>
__global__ __launch_bounds__(512, 2) void wino_conv_kernel(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NC = WINO_NC;
    constexpr int G = IC / 16;                              // 16-channel groups = 16-byte slots of a column's row in one plane
    constexpr int RS = IC / 4;                              // floats per column per plane (unpadded; slots swizzled)
    constexpr int PL = NC * RS;                             // plane stride (a multiple of 64 floats)
    constexpr int XI = 4 * PL;                              // one x image
    constexpr int BUF = 8 * XI;                             // one V buffer
    constexpr int PBUF = 4 * 4 * 64 * 4;                    // one exchange buffer: [4 sums][4 w][64 lanes] x 16 B
    constexpr int S = 4 * IC / 4;                           // MFMA k-steps of a wave
    constexpr int CPG = 16 / G;                             // columns per 16-lane transform group = columns sharing a swizzle
    constexpr int ITEMS = NC * (IC / 4);                    // (column, channel quad) transform items per iteration
    static_assert(PL % 64 == 0 && (G == 8 || G == 4) && ITEMS <= 512, "layout");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, w = wv & 3, h = wv >> 2, q = lane >> 4, nn = lane & 15;
    float *pex = smem + 2 * BUF;

    float A[S];
    {
        const float *ap = a.wpack + ((size_t)w * 2 * S + (size_t)h * S) * 64 + lane;  // the wave's four x are consecutive
#pragma unroll
        for (int s = 0; s < S; ++s) A[s] = ap[(size_t)s * 64];
    }
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + 16 * w + 4 * q);
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};

    const int64_t n_iters = (a.ncols + NC - 1) / NC;
    const int64_t dcol = (int64_t)NC * gridDim.x, dq = dcol / a.tpc;
    const int dr = (int)(dcol - dq * a.tpc);
    const int64_t drow_in = dq * a.pin + 4 * dr, drow_out = dq * a.pout + 4 * dr;
    const int wrap_in = a.pin - 4 * a.tpc, wrap_out = a.pout - 4 * a.tpc;
    // This thread's transform item: column cl of the iteration, channels 16 tg + 4 q .. + 3 (threads beyond ITEMS - IC = 64: the
    // upper four waves - have none).  The 16 lanes of a ds_write_b128 service group share the plane (q) and cover CPG columns x
    // G slots: bank slot G (cl % CPG) + (tg ^ swizzle(cl)), sixteen distinct ones.  A wave's loads cover whole rows.
    const bool has_item = tid < ITEMS;
    const int tg = nn % G, cl = (CPG * wv + nn / G) % NC;
    float *const vdst = smem + q * PL + cl * RS + 4 * (tg ^ ((cl / CPG) % G));
    const float *const src0 = a.in + 16 * tg + 4 * q;
    const int64_t end_row = (a.ncols / a.tpc - 1) * a.pin + 4 * (a.tpc - 1);  // first row of the last column
    WinoCol fc;  // the column this thread fetches next
    fc.init((int64_t)blockIdx.x * NC + cl, a.tpc, a.pin);
    f32x4 d[8];
    // The eight input rows of the item.  A column beyond the end repeats the last one (results never stored).  The last group of a
    // pout that is no multiple of four reaches up to three rows behind its chunk: they enter as ZEROS - in exact arithmetic they
    // only reach the outputs that are not stored, in fp32 they shape the rounding of the stored ones, which must not depend on
    // the neighbouring chunk (and the last chunk has no neighbour).  Rows 0..4 of a group are always inside the chunk.
    auto fetch = [&]() {
        if (has_item) {
            const bool valid = fc.col < a.ncols;
            const int64_t row = valid ? fc.row : end_row;
            const int last = a.pin - 1 - 4 * (valid ? fc.t : a.tpc - 1);  // highest row offset inside the chunk (>= 4)
            const float *src = src0 + (size_t)row * IC;
#pragma unroll
            for (int j = 0; j < 5; ++j) d[j] = *reinterpret_cast<const f32x4 *>(src + j * IC);
#pragma unroll
            for (int j = 5; j < 8; ++j) d[j] = j <= last ? *reinterpret_cast<const f32x4 *>(src + j * IC) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            fc.advance(dcol, drow_in, dr, a.tpc, wrap_in);
        }
    };
    auto transform_to = [&](int buf) {
        if (has_item) {
            f32x4 v[8];
            wino_in_transform(d, v);
#pragma unroll
            for (int x = 0; x < 8; ++x) *reinterpret_cast<f32x4 *>(vdst + buf * BUF + x * XI) = v[x];
        }
    };

    int64_t it = blockIdx.x;
    if (it >= n_iters) return;
    WinoCol oc;  // the column this lane finishes (h == 0 waves): column nn of the iteration
    oc.init(it * NC + nn, a.tpc, a.pout);
    float *const dst0 = a.out + a.out_coff + 16 * w + 4 * q;
    fetch();
    transform_to(0);
    if (it + gridDim.x < n_iters) fetch();
    RMR_SYNC();
    int cur = 0;
    // B fragments of this lane: plane q, column nn, slot g ^ swizzle(nn)
    const float *const rbase = smem + 4 * h * XI + q * PL + nn * RS;
    const int sw = (nn / CPG) % G;
    for (; it < n_iters; it += gridDim.x) {
        const int64_t nxt = it + gridDim.x;
#ifdef RMR_TIMING_ABLATIONS
        if (!(a.abl & 1))
#endif
        if (nxt < n_iters) {
            transform_to(cur ^ 1);
            if (nxt + gridDim.x < n_iters) fetch();
        }
        // ---- four GEMMs of K = IC on this wave's 16 channels x 16 columns
        f32x4 acc[4];
        acc[0] = h == 1 ? b4 : zero;  // h = 1: x = 0 (AT[0][0] = 1: the bias of y0) ...
        acc[1] = zero;
        acc[2] = zero;
        acc[3] = h == 1 ? b4 : zero;  // ... and x = 7 (AT[3][7] = 1: the bias of y3); y1 and y2 take theirs in the partial sums below
        const float *r = rbase + cur * BUF;
        f32x4 x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = *reinterpret_cast<const f32x4 *>(r + k * XI + 4 * sw);
#ifdef RMR_TIMING_ABLATIONS
        if (!(a.abl & 2))
#endif
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f32x4 y[4];
            if (g + 1 < G) {
#pragma unroll
                for (int k = 0; k < 4; ++k) y[k] = *reinterpret_cast<const f32x4 *>(r + k * XI + 4 * ((g + 1) ^ sw));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(k * G + g) * 4 + j], x[k][j], acc[k], 0, 0, 0);
            if (g + 1 < G) {
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = y[k];
            }
        }
        // pin the software pipeline: the reads of step g + 1 are issued before the MFMAs of step g
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g + 1 < G) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
        // ---- AT m.  h = 0 holds m1..m4 (acc 0..3), h = 1 holds m0, m5, m6, m7:
        //   y0 = (m1 + m2) + (m3 + m4)     + [m0 + (m5 + m6)]
        //   y1 = (m1 - m2) + 2 (m3 - m4)   + [(m5 - m6) / 2 + b]
        //   y2 = (m1 + m2) + 4 (m3 + m4)   + [(m5 + m6) / 4 + b]
        //   y3 = (m1 - m2) + 8 (m3 - m4)   + [(m5 - m6) / 8 + m7]
        float *pb = pex + cur * PBUF + (w * 64 + lane) * 4;
#ifdef RMR_TIMING_ABLATIONS
        if ((a.abl & 4) && acc[0][0] != 123456.75f) { RMR_SYNC(); cur ^= 1; continue; }
#endif
        if (h == 1) {
            const f32x4 s56 = acc[1] + acc[2], d56 = acc[1] - acc[2];
            *reinterpret_cast<f32x4 *>(pb + 0 * 1024) = acc[0] + s56;
            *reinterpret_cast<f32x4 *>(pb + 1 * 1024) = fma4(0.5f, d56, b4);
            *reinterpret_cast<f32x4 *>(pb + 2 * 1024) = fma4(0.25f, s56, b4);
            *reinterpret_cast<f32x4 *>(pb + 3 * 1024) = fma4(0.125f, d56, acc[3]);
        }
        RMR_SYNC();  // V[next] complete, V[cur] free, the partial sums of this iteration visible
        if (h == 0) {
            if (oc.col < a.ncols) {
                const f32x4 s12 = acc[0] + acc[1], d12 = acc[0] - acc[1], s34 = acc[2] + acc[3], d34 = acc[2] - acc[3];
                f32x4 yv[4];
                yv[0] = (s12 + s34) + *reinterpret_cast<const f32x4 *>(pb + 0 * 1024);
                yv[1] = fma4(2.0f, d34, d12) + *reinterpret_cast<const f32x4 *>(pb + 1 * 1024);
                yv[2] = fma4(4.0f, s34, s12) + *reinterpret_cast<const f32x4 *>(pb + 2 * 1024);
                yv[3] = fma4(8.0f, d34, d12) + *reinterpret_cast<const f32x4 *>(pb + 3 * 1024);
                float *dst = dst0 + (size_t)oc.row * a.out_row;
                const int nvalid = a.pout - 4 * oc.t;  // positions of this group inside the chunk (>= 1)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < nvalid) {
                        f32x2 lo = f32x2{yv[i][0], yv[i][1]}, hi = f32x2{yv[i][2], yv[i][3]};
                        swish_pk(lo, hi);
                        *reinterpret_cast<f32x4 *>(dst + (size_t)i * a.out_row) = f32x4{lo.x, lo.y, hi.x, hi.y};
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
    const size_t lds = ((size_t)2 * 8 * 4 * WINO_NC * (IC / 4) + (size_t)2 * 4 * 4 * 64 * 4) * sizeof(float);
    WinoArgs a;
    a.in = in; a.out = out; a.wpack = c.wpack; a.bias = c.bias;
    a.pin = pin; a.pout = pout; a.tpc = (pout + 3) / 4;
    a.ncols = n * a.tpc;
    a.out_row = out_row; a.out_coff = out_coff;
    a.abl = abl_int("RMR_WINO_ABLATE", 0);  // ignored unless built with -DRMR_TIMING_ABLATIONS
    const int64_t iters = (a.ncols + WINO_NC - 1) / WINO_NC;
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

// =========================================================================================================================
// Stride 3: Conv_w_ref's seq_conv3 (32 -> 64 channels, 9 taps; models/Conv_w_ref.py:31-32,51) as a polyphase Winograd convolution.
//
//   out[p] = sum_tap w[tap] in[3 p + tap]  =  sum_{r = 0..2} sum_{m = 0..2} w[3 m + r] d_r[p + m],     d_r[i] = in[3 i + r]
//
// - three stride-1 convolutions of three taps on the phase signals d_r, summed.  Each in minimal form over groups of four outputs:
// F(4, 3) at the points 0, +-1, +-2, inf - 6 products per 4 outputs where the direct form has 12.  AT depends on the points only, so
// the three phases accumulate into the SAME x-domain accumulators: per x one GEMM with K = (phase, channel) = 96, half the direct
// form's MFMAs.  Structure of the stride-1 kernel: eight waves, wave (w, h) = 16 channels x three of the six x (h = 0: +1, -1, 0;
// h = 1: +2, -2, inf), 16 columns (groups of four outputs) per iteration, V double-buffered as [x][phase][plane q][group g][column] x 16 B
// (a column tile's 16 columns are 256 contiguous bytes: conflict-free reads and writes), the h = 1 half of AT m through LDS.
// (The same form for the 16-channel layers sig_conv3 / seq_conv2 - K = 48 per x - measured neutral against the kernels that fold
// their producers, k_conv_front.hip, and stayed a patch: profiles/NOTES_r06.md section 8.  This layer's input is in HBM anyway.)
struct WinoS3Args {
    const float *in;     // [n][pin][IC]
    float *out;
    const float *wpack;  // [oc/16][((x * 3 + phase) * G + g) * 4 + j][64 lanes], x in the order 1, 2, 0 | 3, 4, 5
    const float *bias;
    int64_t ncols;       // n * tpc
    int pin, pout, tpc;
    int out_row, out_coff;
};

// BT d of F(4, 3) at 0, 1, -1, 2, -2, inf (oracle/winograd.py), v in the kernel's x order (1, 2, 0, 3, 4, 5)
__device__ __forceinline__ void wino_in_transform6(const f32x4 (&d)[6], f32x4 (&v)[6]) {
    const f32x4 a = fma4(4.0f, d[2], -d[4]);           // 4 d2 - d4
    const f32x4 b = fma4(4.0f, d[1], -d[3]);           // 4 d1 - d3
    v[0] = a + b;
    v[1] = a - b;
    v[2] = fma4(-5.0f, d[2], fma4(4.0f, d[0], d[4]));  // 4 d0 - 5 d2 + d4
    const f32x4 c = d[4] - d[2], e = d[3] - d[1];
    v[3] = fma4(2.0f, e, c);
    v[4] = fma4(-2.0f, e, c);
    v[5] = fma4(-5.0f, d[3], fma4(4.0f, d[1], d[5]));  // 4 d1 - 5 d3 + d5
}

template <int IC>
__global__ __launch_bounds__(512, 2) void wino_s3_kernel(WinoS3Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NC = WINO_NC;
    constexpr int G = IC / 16;
    constexpr int PLC = NC * 4;             // one (x, phase, q, g) plane: 16 columns x 16 B
    constexpr int XI = 3 * 4 * G * PLC;     // one x image
    constexpr int BUF = 6 * XI;             // one V buffer
    constexpr int PBUF = 4 * 4 * 64 * 4;    // exchange buffer: [4 sums][4 w][64 lanes] x 16 B
    constexpr int KX = 3 * G * 4;           // MFMA k-steps per x
    constexpr int S = 3 * KX;               // ... of a wave
    constexpr int ITEMS = NC * 3 * 4 * G;   // (column, phase, plane, group) transform items per iteration
    static_assert(ITEMS <= 512, "one transform item per thread");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, w = wv & 3, h = wv >> 2, q = lane >> 4, nn = lane & 15;
    float *pex = smem + 2 * BUF;

    float A[S];
    {
        const float *ap = a.wpack + ((size_t)w * 2 * S + (size_t)h * S) * 64 + lane;
#pragma unroll
        for (int s = 0; s < S; ++s) A[s] = ap[(size_t)s * 64];
    }
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + 16 * w + 4 * q);
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    // The weight slice has arrived before the loop is entered: with its loads still "in flight" at the loop header hipcc puts a wait for
    // EVERY outstanding load behind the first MFMA of each iteration - the prefetched rows included (s_waitcnt vmcnt(0): 1.62 ms per
    // 250 k chunks instead of 1.2)
    __builtin_amdgcn_s_waitcnt(0x0F70);

    const int64_t n_iters = (a.ncols + NC - 1) / NC;
    const int64_t dcol = (int64_t)NC * gridDim.x, dq = dcol / a.tpc;
    const int dr = (int)(dcol - dq * a.tpc);
    const int64_t drow_in = dq * a.pin + 12 * dr, drow_out = dq * a.pout + 4 * dr;
    const int wrap_in = a.pin - 12 * a.tpc, wrap_out = a.pout - 4 * a.tpc;
    // transform item of wave wv < 3 G: phase wv / G, group wv % G, plane q, column nn: a wave's load covers 16 rows x 64 contiguous bytes
    const bool has_item = wv < 3 * G;
    const int ph = has_item ? wv / G : 0, tg = wv % G;
    float *const vdst = smem + ((ph * 4 + q) * G + tg) * PLC + nn * 4;
    const float *const src0 = a.in + ph * IC + 16 * tg + 4 * q;
    WinoCol fc;
    fc.col = (int64_t)blockIdx.x * NC + nn;
    {
        const int64_t ch = fc.col / a.tpc;
        fc.t = (int)(fc.col - ch * a.tpc);
        fc.row = ch * a.pin + 12 * fc.t;
    }
    const int64_t end_row = (a.ncols / a.tpc - 1) * a.pin + 12 * (a.tpc - 1);
    f32x4 d[6];
    int d_lim = 15;  // highest row offset 3 j of the fetched item that lies inside its chunk
    // Rows 12 t + 3 j + phase of the chunk, j = 0..5; rows behind the chunk enter as zeros (see the stride-1 kernel) - loaded from a
    // clamped address and zeroed in the transform: a load under divergent control flow makes hipcc wait for every load in flight.
    auto fetch = [&]() {
        if (has_item) {
            const bool valid = fc.col < a.ncols;
            const int64_t row = valid ? fc.row : end_row;
            d_lim = a.pin - 1 - ph - 12 * (valid ? fc.t : a.tpc - 1);
            const float *src = src0 + (size_t)row * IC;
#pragma unroll
            for (int j = 0; j < 6; ++j) d[j] = *reinterpret_cast<const f32x4 *>(src + (3 * j <= d_lim ? 3 * j : 0) * IC);
            fc.advance(dcol, drow_in, dr, a.tpc, wrap_in);
        }
    };
    auto transform_to = [&](int buf) {
        if (has_item) {
            if (d_lim < 15) {
#pragma unroll
                for (int j = 1; j < 6; ++j) d[j] = 3 * j <= d_lim ? d[j] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
            f32x4 v[6];
            wino_in_transform6(d, v);
#pragma unroll
            for (int x = 0; x < 6; ++x) *reinterpret_cast<f32x4 *>(vdst + buf * BUF + x * XI) = v[x];
        }
    };

    int64_t it = blockIdx.x;
    if (it >= n_iters) return;
    WinoCol oc;
    oc.init(it * NC + nn, a.tpc, a.pout);
    float *const dst0 = a.out + a.out_coff + 16 * w + 4 * q;
    fetch();
    transform_to(0);
    if (it + gridDim.x < n_iters) fetch();
    RMR_SYNC();
    int cur = 0;
    const float *const rbase = smem + 3 * h * XI + q * G * PLC + nn * 4;
    for (; it < n_iters; it += gridDim.x) {
        const int64_t nxt = it + gridDim.x;
        if (nxt < n_iters) {
            transform_to(cur ^ 1);
            if (nxt + gridDim.x < n_iters) fetch();
        }
        // ---- three GEMMs of K = 3 IC on this wave's 16 channels x 16 columns; a step = (phase, group g): three B fragments, 12 MFMAs
        f32x4 acc[3];
        acc[0] = zero;
        acc[1] = zero;
        acc[2] = b4;  // h = 0: x = 0 (AT[0][0] = 1: the bias of y0); h = 1: x = 5 (AT[3][5] = 1: the bias of y3)
        const float *r = rbase + cur * BUF;
        constexpr int NS = 3 * G;
        f32x4 xv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) xv[k] = *reinterpret_cast<const f32x4 *>(r + k * XI);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int p = st / G, g = st % G;
            f32x4 yv[3];
            if (st + 1 < NS) {
                const int p1 = (st + 1) / G, g1 = (st + 1) % G;
#pragma unroll
                for (int k = 0; k < 3; ++k) yv[k] = *reinterpret_cast<const f32x4 *>(r + k * XI + (p1 * 4 * G + g1) * PLC);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[k * KX + (p * G + g) * 4 + j], xv[k][j], acc[k], 0, 0, 0);
            if (st + 1 < NS) {
#pragma unroll
                for (int k = 0; k < 3; ++k) xv[k] = yv[k];
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            if (st + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        }
        // ---- AT m.  h = 0 holds m1, m2, m0 (acc 0..2), h = 1 holds m3, m4, m5:
        //   y0 = m0 + (m1 + m2)  + [m3 + m4]             y2 = (m1 + m2) + [4 (m3 + m4) + b]
        //   y1 = (m1 - m2)       + [2 (m3 - m4) + b]     y3 = (m1 - m2) + [8 (m3 - m4) + m5]
        float *pb = pex + cur * PBUF + (w * 64 + lane) * 4;
        if (h == 1) {
            const f32x4 s34 = acc[0] + acc[1], d34 = acc[0] - acc[1];
            *reinterpret_cast<f32x4 *>(pb + 0 * 1024) = s34;
            *reinterpret_cast<f32x4 *>(pb + 1 * 1024) = fma4(2.0f, d34, b4);
            *reinterpret_cast<f32x4 *>(pb + 2 * 1024) = fma4(4.0f, s34, b4);
            *reinterpret_cast<f32x4 *>(pb + 3 * 1024) = fma4(8.0f, d34, acc[2]);
        }
        RMR_SYNC();  // V[next] complete, V[cur] free, the partial sums of this iteration visible
        if (h == 0) {
            if (oc.col < a.ncols) {
                const f32x4 s12 = acc[0] + acc[1], d12 = acc[0] - acc[1];
                f32x4 yo[4];
                yo[0] = (acc[2] + s12) + *reinterpret_cast<const f32x4 *>(pb + 0 * 1024);
                yo[1] = d12 + *reinterpret_cast<const f32x4 *>(pb + 1 * 1024);
                yo[2] = s12 + *reinterpret_cast<const f32x4 *>(pb + 2 * 1024);
                yo[3] = d12 + *reinterpret_cast<const f32x4 *>(pb + 3 * 1024);
                float *dst = dst0 + (size_t)oc.row * a.out_row;
                const int nvalid = a.pout - 4 * oc.t;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < nvalid) {
                        f32x2 lo = f32x2{yo[i][0], yo[i][1]}, hi = f32x2{yo[i][2], yo[i][3]};
                        swish_pk(lo, hi);
                        *reinterpret_cast<f32x4 *>(dst + (size_t)i * a.out_row) = f32x4{lo.x, lo.y, hi.x, hi.y};
                    }
                }
            }
            oc.advance(dcol, drow_out, dr, a.tpc, wrap_out);
        }
        cur ^= 1;
    }
}

template <int IC>
static int launch_wino_s3_t(rmr_engine *e, const ConvLayer &c, const float *in, int pin, float *out, int out_row, int out_coff, int pout,
                            int64_t n) {
    const size_t lds = ((size_t)2 * 6 * 3 * 4 * (IC / 16) * WINO_NC * 4 + (size_t)2 * 4 * 4 * 64 * 4) * sizeof(float);
    WinoS3Args a;
    a.in = in; a.out = out; a.wpack = c.wpack; a.bias = c.bias;
    a.pin = pin; a.pout = pout; a.tpc = (pout + 3) / 4;
    a.ncols = n * a.tpc;
    a.out_row = out_row; a.out_coff = out_coff;
    const int64_t iters = (a.ncols + WINO_NC - 1) / WINO_NC;
    int64_t grid = (int64_t)e->num_cus;
    if (grid > iters) grid = iters;
    if (grid < 1) return 0;
    auto kern = wino_s3_kernel<IC>;
    RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(kern)));
    ProfScope ps(e, c.kid);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

bool conv_wino_s3_supported(const ConvLayer &c, int pin, int pout) {
    return c.wpack && c.stride == 3 && c.ic == 32 && c.oc == 64 && c.kw == 9 && pout >= 1 && pout == (pin - c.kw) / 3 + 1;
}

int launch_conv_wino_s3(rmr_engine *e, const ConvLayer &c, const float *in, int in_row, int pin, float *out, int out_row, int out_coff,
                        int pout, int64_t n) {
    if (in_row != c.ic) RMR_FAIL(RMR_ERR_INVALID, "conv input row %d != ic %d", in_row, c.ic);
    return launch_wino_s3_t<32>(e, c, in, pin, out, out_row, out_coff, pout, n);
}

bool conv_wino_supported(const ConvLayer &c, int pin, int pout) {
    return c.wpack && c.kw == 5 && c.stride == 1 && c.oc == 64 && (c.ic == 128 || c.ic == 64) && pin == pout + 4 && pout >= 4;
}

int launch_conv_wino(rmr_engine *e, const ConvLayer &c, const float *in, int in_row, int pin, float *out, int out_row, int out_coff,
                     int pout, int64_t n) {
    if (in_row != c.ic) RMR_FAIL(RMR_ERR_INVALID, "conv input row %d != ic %d", in_row, c.ic);
    if (c.ic == 128) return launch_wino_t<128>(e, c, in, pin, out, out_row, out_coff, pout, n);
    if (c.ic == 64) return launch_wino_t<64>(e, c, in, pin, out, out_row, out_coff, pout, n);
    RMR_FAIL(RMR_ERR_INVALID, "no Winograd kernel for ic=%d", c.ic);
}

}  // namespace rmr
