// k_conv.hip — 1-D convolution + folded BatchNorm + swish as an implicit GEMM on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32), channel-last activations.
//
// Replaces the nn.Conv1d/BatchNorm1d/swish triples of the reference networks that have
// >= 16 input channels: models/ConvLSTM_w_ref.py:43 (sig_conv3), :46 (seq_conv2),
// :50 (merge_conv1); models/Conv_w_ref.py:47,50-51,54-57.
//
// GEMM view:  out[oc][col] = sum_{tap, ic} W[oc][ic][tap] * in[chunk(col)][pos(col)*S + tap][ic]
//   M = oc  (one 16-row MFMA tile per wave: wave w owns channels 16w..16w+15; the whole
//            K-extent of its weight slice lives in VGPRs for the lifetime of the block)
//   N = (chunk, output position) flattened, 16 columns per MFMA tile
//   K = (tap, ic); within a 16-channel group g the four MFMA k-lanes q=lane>>4 take
//       channels 16g+4q+j for MFMA j=0..3, so ONE ds_read_b128 of 4 consecutive channels
//       feeds four MFMAs, and the D fragment (4 consecutive oc x 1 column per lane) is
//       ONE 16-byte store in the channel-last output.
//
// LDS: CB chunks of input staged as FOUR PLANES, plane q holding channels {16g+4q+j} of every
// row (IC/4 floats per row, padded so that rowstride/4 is odd), plane stride a multiple of
// 64 floats.  ds_read_b128 is serviced in 16-lane groups that contain all 16 columns n of a
// tile but mixed q; with the q-dependence a multiple of 256 B and an odd row stride the 16
// lanes of every group land on 16 distinct 16-byte bank slots: conflict-free (a row-padded
// [row][IC+4] image measured 54 % of LDS cycles lost to 2-way conflicts, profiles/r01).
#include "rmr_internal.h"
#include "rmr_math.h"

namespace rmr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float *in;
    float *out;
    const float *apack;
    const float *bias;
    int64_t n;      // chunks
    int in_row;     // floats per input position (== IC)
    int pin, pout;  // positions per chunk in / out
    int out_row;    // floats per output position
    int out_coff;   // channel offset of this layer's output inside out_row
    int cb;         // chunks per block iteration
    int plane;      // LDS plane stride in floats (multiple of 64)
    FastDiv div_pout;
    // Position windows (long chunk contexts: one chunk's input no longer fits a block's LDS share).  nwin > 1: an iteration is
    // ONE window of ONE chunk (cb == 1) - `pin` / `pout` above are then the rows staged / the columns computed per window, the
    // chunk's own extents are pin_total / pout_total, and window w covers output positions [w * pout, (w + 1) * pout).  The
    // reference's Conv1d knows no such limit (models/ConvLSTM_w_ref.py:39-58 is length-agnostic).
    int nwin, pin_total, pout_total;
};


// Waves per SIMD the register allocator is held to.  Left alone it takes 136-152 registers for layers that fit 128 without a
// spill (3 waves per SIMD instead of 4: merge_conv2 140, the 16-channel stride-3 layers 120-136); merge_conv1's 160-register
// weight slice needs two-wave occupancy, Conv_w_ref's seq_conv3 (72-register slice + 8 x 16-byte staging loads) three.
constexpr int conv_min_waves(int ic, int kw) { return kw * ic / 4 >= 160 ? 2 : (kw * ic / 4 >= 72 ? 3 : 4); }

// one (TWO = false) or two 16-column tiles of the implicit GEMM: independent accumulator chains, B fragments fetched one
// (tap, g) step ahead of the MFMAs that consume them, swish epilogue, 16-byte channel-last stores
// `store(chunk in the iteration, output position, swish(acc) as four consecutive output channels)` receives every valid column
template <int IC, int KW, int STRIDE, bool TWO, typename Store>
__device__ __forceinline__ void conv_tiles_to(const float *smem, int plane, int pin, int pout, FastDiv div_pout, const float (&A)[KW * IC / 4],
                                              const f32x4 b4, int ncols, int tile, int q, int nn, Store store) {
    constexpr int G = IC / 16;
    constexpr int RS = (G % 2 == 0) ? IC / 4 + 4 : IC / 4;
    constexpr int NS = KW * G;
    int col0 = tile * 16 + nn, col1 = col0 + 16;
    const bool v0 = col0 < ncols, v1 = TWO && col1 < ncols;
    col0 = v0 ? col0 : ncols - 1;
    col1 = v1 ? col1 : ncols - 1;
    const int ch0 = (int)(((float)col0 + 0.5f) * div_pout.inv);
    const int ch1 = (int)(((float)col1 + 0.5f) * div_pout.inv);
    const int p0 = col0 - ch0 * pout, p1 = col1 - ch1 * pout;
    const float *r0 = smem + (size_t)q * plane + (size_t)(ch0 * pin + p0 * STRIDE) * RS;
    const float *r1 = smem + (size_t)q * plane + (size_t)(ch1 * pin + p1 * STRIDE) * RS;
    f32x4 acc0 = b4, acc1 = b4;
    f32x4 x0 = *reinterpret_cast<const f32x4 *>(r0), x1 = x0;
    if (TWO) x1 = *reinterpret_cast<const f32x4 *>(r1);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        f32x4 y0 = x0, y1 = x1;
        if (st + 1 < NS) {
            const int tap = (st + 1) / G, g = (st + 1) % G;
            y0 = *reinterpret_cast<const f32x4 *>(r0 + tap * RS + 4 * g);
            if (TWO) y1 = *reinterpret_cast<const f32x4 *>(r1 + tap * RS + 4 * g);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[st * 4 + j], x0[j], acc0, 0, 0, 0);
            if (TWO) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[st * 4 + j], x1[j], acc1, 0, 0, 0);
        }
        x0 = y0;
        x1 = y1;
    }
    // pin the software pipeline: reads of step st+1 are issued before the MFMAs of step st
    constexpr int T = TWO ? 2 : 1;
    __builtin_amdgcn_sched_group_barrier(0x100, T, 0);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        if (st + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, T, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * T, 0);
    }
    if (v0) {
        f32x2 lo = f32x2{acc0[0], acc0[1]}, hi = f32x2{acc0[2], acc0[3]};
        swish_pk(lo, hi);  // same operations as swish_f, the plain ones two values per instruction
        store(ch0, p0, f32x4{lo.x, lo.y, hi.x, hi.y});
    }
    if (v1) {
        f32x2 lo = f32x2{acc1[0], acc1[1]}, hi = f32x2{acc1[2], acc1[3]};
        swish_pk(lo, hi);
        store(ch1, p1, f32x4{lo.x, lo.y, hi.x, hi.y});
    }
}

// the kernel below: columns leave as 16-byte channel-last stores to HBM
template <int IC, int KW, int STRIDE, bool TWO>
__device__ __forceinline__ void conv_tiles(const ConvArgs &a, const float *smem, const float (&A)[KW * IC / 4], const f32x4 b4,
                                           int64_t chunk0, int pbase, int ncols, int tile, int w, int q, int nn) {
    conv_tiles_to<IC, KW, STRIDE, TWO>(smem, a.plane, a.pin, a.pout, a.div_pout, A, b4, ncols, tile, q, nn,
                                       [&](int ch, int p, const f32x4 y) {
                                           float *dst = a.out + ((size_t)(chunk0 + ch) * a.pout_total + pbase + p) * a.out_row + a.out_coff + 16 * w + 4 * q;
                                           *reinterpret_cast<f32x4 *>(dst) = y;
                                       });
}

template <int IC, int KW, int STRIDE>
__global__ __launch_bounds__(256, conv_min_waves(IC, KW)) void conv_mfma_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int G = IC / 16;        // 16-channel groups
    constexpr int RS = (G % 2 == 0) ? IC / 4 + 4 : IC / 4;  // floats per row per plane, RS/4 odd
    constexpr int S = KW * IC / 4;    // MFMA k-steps
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15;

    // register-resident weight slice of this wave
    float A[S];
    {
        const float *ap = a.apack + (size_t)w * S * 64 + lane;
#pragma unroll
        for (int s = 0; s < S; ++s) A[s] = ap[(size_t)s * 64];
    }
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + 16 * w + 4 * q);

    const int64_t n_iters = a.nwin > 1 ? a.n * a.nwin : (a.n + a.cb - 1) / a.cb;
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        // whole chunks (nwin == 1): cb chunks from chunk0;  windows: window `win` of chunk `chunk0`, rows clipped to the chunk
        int64_t chunk0 = it * a.cb;
        int nch = (int)((a.n - chunk0) < a.cb ? (a.n - chunk0) : a.cb), pbase = 0, rows = nch * a.pin, ncols = nch * a.pout;
        size_t src_row = (size_t)chunk0 * a.pin;
        if (a.nwin > 1) {
            chunk0 = it / a.nwin;  // (64-bit: n * nwin may pass 2^31 only in theory, the division is per iteration anyway)
            const int win = (int)(it - chunk0 * a.nwin);
            pbase = win * a.pout;
            nch = 1;
            ncols = a.pout_total - pbase < a.pout ? a.pout_total - pbase : a.pout;
            rows = a.pin_total - pbase * STRIDE < a.pin ? a.pin_total - pbase * STRIDE : a.pin;
            src_row = (size_t)chunk0 * a.pin_total + (size_t)pbase * STRIDE;
        }
        RMR_SYNC();  // all reads of the previous iteration are done
        {                 // stage `rows` rows of IC floats into the 4 planes
            constexpr int R4 = IC / 4;
            constexpr int UNR = 8;  // loads in flight per thread before the first LDS write
            const int total4 = rows * R4;
            const float4 *src = reinterpret_cast<const float4 *>(a.in + src_row * a.in_row);
            for (int base = tid; base < total4; base += UNR * (int)blockDim.x) {
                float4 v[UNR];
                int dsto[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int i = base + u * (int)blockDim.x;
                    const int ii = i < total4 ? i : total4 - 1;
                    const int row = ii / R4, c = ii - row * R4;
                    const int qq = c / G, g = c - qq * G;  // consecutive lanes: same plane, consecutive g
                    v[u] = src[row * R4 + 4 * g + qq];
                    dsto[u] = i < total4 ? qq * a.plane + row * RS + 4 * g : 4 * a.plane;  // trash slot
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) *reinterpret_cast<float4 *>(smem + dsto[u]) = v[u];
            }
        }
        RMR_SYNC();
        const int ntiles = (ncols + 15) >> 4;
        for (int tile = 0; tile < ntiles; tile += 2) {
            // an odd last tile runs alone (wave-uniform): no MFMA is spent on a padding tile
            if (tile + 1 < ntiles) conv_tiles<IC, KW, STRIDE, true>(a, smem, A, b4, chunk0, pbase, ncols, tile, w, q, nn);
            else conv_tiles<IC, KW, STRIDE, false>(a, smem, A, b4, chunk0, pbase, ncols, tile, w, q, nn);
        }
    }
}

template <int IC, int KW, int STRIDE>
static int launch_conv_t(rmr_engine *e, const ConvLayer &c, const float *in, int in_row, int pin,
                         float *out, int out_row, int out_coff, int pout, int64_t n) {
    constexpr int G = IC / 16;
    constexpr int RS = (G % 2 == 0) ? IC / 4 + 4 : IC / 4;
    // chunks per iteration.  How many blocks a CU holds is set by the instantiation's registers (512 per SIMD lane, granule
    // 8: the 16-channel layers reach 3-4 waves per SIMD, merge_conv1's 160-register weight slice 2), so the LDS budget of a
    // block is its share of the 160 KB at that occupancy, capped at 72 KB (two blocks per CU).  Among the
    // chunk counts that fit, the one whose columns fill their 16-column tiles best wins (Conv_w_ref's merge_conv1: 4 x 20
    // columns = 5 tiles exactly, where 5 chunks would pad the 7th tile to 25 %); ties go to the larger count.
    const int threads_pb = 64 * (c.oc / 16);
    static int regs = 0;  // per instantiation; the same binary on every device
    if (regs == 0) {
        hipFuncAttributes attr;
        regs = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(conv_mfma_kernel<IC, KW, STRIDE>)) == hipSuccess ? attr.numRegs : 256;
        if (regs < 1) regs = 256;
    }
    int wps = 512 / ((regs + 7) & ~7);
    wps = wps < 1 ? 1 : (wps > 8 ? 8 : wps);
    int resident = wps * 4 / (threads_pb / 64);
    resident = resident < 1 ? 1 : (resident > 8 ? 8 : resident);
    const size_t row_bytes = (size_t)pin * RS * 4 * sizeof(float);  // all four planes
    size_t budget = (size_t)73728;
    const size_t share = (size_t)160 * 1024 / resident - 512;
    if (share < budget) budget = share;
    int cb_max = (int)(budget / row_bytes);
    if (cb_max < 1) cb_max = 1;
    if (cb_max > 8) cb_max = 8;
    int cb = cb_max;
    double best = -1.0;
    for (int k = cb_max; k >= (cb_max + 1) / 2; --k) {
        const int cols = k * pout;
        const double eff = (double)cols / (16.0 * ((cols + 15) / 16));
        if (eff > best + 1e-9) { best = eff; cb = k; }
    }
    // a small batch (one read per call: a few hundred chunks) spread over the CUs: fewer chunks per iteration until there is a block
    // for every CU - a chunk's columns are computed the same way whatever its neighbours in the iteration (same bits), and a
    // half-filled tile on an otherwise idle CU costs nothing
    while (cb > 1 && (n + cb - 1) / cb < e->num_cus) cb = (cb + 1) / 2;
    // a chunk whose rows do not fit a block's share goes through position windows: the most output positions whose input rows
    // ((win - 1) * STRIDE + KW of them) fit the share, one window of one chunk per iteration
    int nwin = 1, pin_w = pin, pout_w = pout;
    if (row_bytes > budget) {
        const int rows_fit = (int)(budget / ((size_t)RS * 4 * sizeof(float)));
        pout_w = (rows_fit - KW) / STRIDE + 1;
        if (pout_w < 16) RMR_FAIL(RMR_ERR_INVALID, "conv layer: not even a 16-column window fits %zu B of LDS", budget);
        pout_w &= ~15;  // whole column tiles
        nwin = (pout + pout_w - 1) / pout_w;
        pin_w = (pout_w - 1) * STRIDE + KW;
        cb = 1;
    }
    const int plane = ((cb * pin_w * RS) + 63) & ~63;
    const size_t lds = (size_t)plane * 4 * sizeof(float) + 64;  // + trash slot for masked staging writes
    if (lds > 160 * 1024) RMR_FAIL(RMR_ERR_INVALID, "conv layer needs %zu B of LDS", lds);
    ConvArgs a;
    a.in = in; a.out = out; a.apack = c.apack; a.bias = c.bias; a.n = n;
    a.in_row = in_row; a.pin = pin_w; a.pout = pout_w; a.out_row = out_row; a.out_coff = out_coff;
    a.cb = cb; a.plane = plane; a.div_pout = make_fastdiv(pout_w);
    a.nwin = nwin; a.pin_total = pin; a.pout_total = pout;
    const int64_t iters = nwin > 1 ? n * nwin : (n + cb - 1) / cb;
    const int threads = threads_pb;
    // persistent blocks: a grid of several times the resident count evens out the tail
    int64_t grid = (int64_t)e->num_cus * 8 * (c.oc >= 64 ? 1 : 64 / c.oc);
    if (grid > iters) grid = iters;
    if (grid < 1) return 0;
    auto kern = conv_mfma_kernel<IC, KW, STRIDE>;
    RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(kern)));
    ProfScope ps(e, c.kid);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

// (A one-launch tail for Conv_w_ref - merge_conv3 -> merge_conv4 -> flatten + fc with both intermediates in LDS - was built
//  on conv_tiles_to in round 3 and measured: bit-identical, 4.38 ns per chunk against 1.92 + 1.00 + 0.56 = 3.48 for the three
//  launches.  Two weight slices cost 160 VGPRs (three waves per SIMD instead of four) and an iteration of 6-8 chunks has
//  only 5 column tiles of work between its four barriers; removed.)

int launch_conv(rmr_engine *e, const ConvLayer &c, const float *in, int in_row, int pin,
                float *out, int out_row, int out_coff, int pout, int64_t n) {
    if (in_row != c.ic) RMR_FAIL(RMR_ERR_INVALID, "conv input row %d != ic %d", in_row, c.ic);
    if (c.apack4) return launch_conv_stream(e, c, in, in_row, pin, out, out_row, out_coff, pout, n);  // > 64 channels: k_stream.hip
    // 5 taps, stride 1, 64 output channels: Winograd F(4, 5), 0.4 of the MFMAs (k_wino.hip).  RMR_WINOGRAD=0: the direct form below
    // (its comparand, tests/test_gpu_wino.py).  Every batch size takes it: the bits of a chunk do not depend on the batch it arrives in
    if (conv_wino_supported(c, pin, pout) && tune_int("RMR_WINOGRAD", 1)) return launch_conv_wino(e, c, in, in_row, pin, out, out_row, out_coff, pout, n);
    if (conv_wino_s3_supported(c, pin, pout) && tune_int("RMR_WINOGRAD", 1)) return launch_conv_wino_s3(e, c, in, in_row, pin, out, out_row, out_coff, pout, n);
#define RMR_CONV_CASE(IC_, KW_, ST_)                                  \
    if (c.ic == IC_ && c.kw == KW_ && c.stride == ST_)                \
        return launch_conv_t<IC_, KW_, ST_>(e, c, in, in_row, pin, out, out_row, out_coff, pout, n);
    // ConvLSTM_w_ref (size 64 / 32 / 16)
    RMR_CONV_CASE(16, 9, 3)    // sig_conv3
    RMR_CONV_CASE(16, 13, 3)   // seq_conv2
    RMR_CONV_CASE(128, 5, 1)   // merge_conv1, size 64
    RMR_CONV_CASE(64, 5, 1)    // merge_conv1 size 32; Conv_w_ref merge_conv2 size 64
    RMR_CONV_CASE(32, 5, 1)    // merge_conv1 size 16; merge_conv2 size 32
    // Conv_w_ref
    RMR_CONV_CASE(16, 11, 1)   // seq_conv2
    RMR_CONV_CASE(32, 9, 3)    // seq_conv3
    RMR_CONV_CASE(64, 3, 2)    // merge_conv3/4 size 64
    RMR_CONV_CASE(32, 3, 2)
    RMR_CONV_CASE(16, 5, 1)
    RMR_CONV_CASE(16, 3, 2)
#undef RMR_CONV_CASE
    RMR_FAIL(RMR_ERR_INVALID, "no conv kernel for ic=%d kw=%d stride=%d", c.ic, c.kw, c.stride);
}

}  // namespace rmr
