// k_conv_front.hip — fp32 ConvLSTM_w_ref, size 64: sig_conv3 and seq_conv2 with their producers folded in.
//
//   sig3_front_kernel : signal -> sig_conv1 -> sig_conv2 (VALU, packed fp32) -> LDS planes -> sig_conv3 (fp32 MFMA)
//   seq2_front_kernel : (sequence, mapping, length) -> seq_conv1 as the two-level gather-sum -> LDS planes
//                       -> seq_conv2 (fp32 MFMA)
//
// Replaces, for this model shape, the pairs front_sig_kernel + conv_mfma_kernel<16,9,3> and front_seq_kernel +
// conv_mfma_kernel<16,13,3> (k_front.hip, k_conv.hip; models/ConvLSTM_w_ref.py:41-46 and
// src/remora/encoded_kmers.pyx:13-45).  The standalone front kernels were bound by HBM, not by their arithmetic:
// 12 KB per chunk of fp32 sig2 / seq1 written (3.9 and 2.2 TB/s) only to be read back by the next launch, 10 % of
// the fp32 step.  Here the 16-channel rows are produced straight into the four-plane LDS image the MFMA tiles read
// (k_conv.hip: plane q = channels 4q..4q+3 of every row), so they never exist in HBM.
//
// Block = 4 waves.  Producer phase: wave w makes the rows of chunks w, w+4, ... of the iteration on its own (every
// per-chunk LDS region is private to one wave: wave-level ordering only, as in k_front.hip).  After one block
// barrier the matrix phase is conv_mfma_kernel's: wave w owns output channels 16w..16w+15 with its weight slice in
// registers.  The arithmetic is that of the separate kernels, operation for operation (same results bit for bit).
//
// Measured (1 M chunks, MI355X; ms per 131 k chunks): sig 0.20 + 0.60 -> 0.68, seq 0.365 + 0.776 -> 1.00; headline
// 22.5 -> 24.1 M chunks/s together with the larger sub-batch.  Timing ablations (make abl, RMR_CONV_FRONT_ABLATE):
// producer phase alone 0.18 / 0.31, matrix phase alone 0.53 / 0.75 - the two add up.  Starting every second block of a
// CU half an iteration late (per-CU arrival counters) changed nothing, nor did four accumulator chains per wave: a
// producer phase running under the other block's matrix phase is slowed by what it saves (shared fp32 datapath).
// What did pay inside the producers: the K gathers of an item batched before the first add (1.22 -> 1.12), exact tile
// groups + the next iteration's rows requested under the matrix phase (-> 1.00; sig -3 %).
#include <algorithm>

#include "rmr_internal.h"
#include "rmr_math.h"

namespace rmr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ void wave_sync() {
    RMR_JITTER_POINT();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct ConvFrontArgs {
    // producer side
    const float *signal;                      // [n][L]                     (sig)
    const float *w_sig1, *b_sig1, *w_sig2, *b_sig2;
    const int8_t *seqs;                       // [n][seq_w]                 (seq)
    const int16_t *maps;                      // [n][map_w]
    const int16_t *lens;                      // [n]
    const float *wt5;                         // [5][K][5][16] gather table (row 4 of a slot = zeros)
    const float *b_seq1;                      // [16]
    int L, P1, seq_w, map_w, K, maxlen;
    int o_front;                              // LDS offset (floats) of the producer scratch
    int per_chunk;                            // producer scratch per chunk (floats)
    int o_map, o_seq, o_code, o_pidx, o_u;    // seq: offsets inside a chunk's scratch (floats)
    // matrix side (as ConvArgs)
    float *out;
    const float *apack, *bias;
    int64_t n;
    int pin, pout, out_row, out_coff, cb, plane;
    FastDiv div_pout;
    int abl;  // experiment builds only (-DRMR_TIMING_ABLATIONS): 1 = skip the producer phase, 2 = skip the matrix phase
    // sig3_front_wino_kernel: sig_conv3 in polyphase Winograd form (k_wino.hip has the algebra)
    const float *wpack;   // [oc/16][(x * 3 + phase) * 4 + j][64 lanes], natural point order
    int o_v, vplane;      // LDS offset (floats) of the x-domain image V, floats per (x, phase, q) plane = padded columns * 4
    int ngrp;             // groups of four output positions per chunk
    FastDiv div_ngrp;
};

#ifdef RMR_TIMING_ABLATIONS
#define CF_ABL(bit) (a.abl & (bit))
#else
#define CF_ABL(bit) 0
#endif

// the matrix phase of conv_mfma_kernel<16, KW, 3> on the staged planes (RS = 4 floats per row and plane): groups of up
// to four column tiles with one accumulator chain each; the last group of an iteration holds what is left (1-4 tiles,
// wave-uniform) so that no MFMA is spent on padding tiles
template <int KW, int NT>
__device__ __forceinline__ void mfma_group(const ConvFrontArgs &a, const float *smem, const float (&A)[KW * 4], f32x4 b4,
                                           int64_t chunk0, int ncols, int tile, int w, int q, int nn) {
    constexpr int RS = 4, STRIDE = 3, NS = KW;
    const float *r[NT];
    int chn[NT], pp[NT];
    bool valid[NT];
    f32x4 acc[NT], x[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        int col = (tile + k) * 16 + nn;
        valid[k] = col < ncols;
        col = valid[k] ? col : ncols - 1;
        chn[k] = (int)(((float)col + 0.5f) * a.div_pout.inv);
        pp[k] = col - chn[k] * a.pout;
        r[k] = smem + (size_t)q * a.plane + (size_t)(chn[k] * a.pin + pp[k] * STRIDE) * RS;
        acc[k] = b4;
        x[k] = *reinterpret_cast<const f32x4 *>(r[k]);
    }
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        f32x4 y[NT];  // B fragments are fetched one tap ahead of the MFMAs that consume them
#pragma unroll
        for (int k = 0; k < NT; ++k) y[k] = st + 1 < NS ? *reinterpret_cast<const f32x4 *>(r[k] + (st + 1) * RS) : x[k];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < NT; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[st * 4 + j], x[k][j], acc[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NT; ++k) x[k] = y[k];
    }
    __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        if (st + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        if (valid[k]) {
            f32x2 lo = f32x2{acc[k][0], acc[k][1]}, hi = f32x2{acc[k][2], acc[k][3]};
            swish_pk(lo, hi);  // same operations as swish_f, the plain ones two values per instruction
            const f32x4 y = {lo.x, lo.y, hi.x, hi.y};
            float *dst = a.out + ((size_t)(chunk0 + chn[k]) * a.pout + pp[k]) * a.out_row + a.out_coff + 16 * w + 4 * q;
            *reinterpret_cast<f32x4 *>(dst) = y;
        }
    }
}

template <int KW>
__device__ __forceinline__ void mfma_phase(const ConvFrontArgs &a, const float *smem, const float (&A)[KW * 4], f32x4 b4,
                                           int64_t chunk0, int nch, int w, int q, int nn) {
    const int ncols = nch * a.pout;
    const int ntiles = (ncols + 15) >> 4;
    int tile = 0;
    for (; tile + 4 <= ntiles; tile += 4) mfma_group<KW, 4>(a, smem, A, b4, chunk0, ncols, tile, w, q, nn);
    const int left = ntiles - tile;
    if (left == 3) mfma_group<KW, 3>(a, smem, A, b4, chunk0, ncols, tile, w, q, nn);
    else if (left == 2) mfma_group<KW, 2>(a, smem, A, b4, chunk0, ncols, tile, w, q, nn);
    else if (left == 1) mfma_group<KW, 1>(a, smem, A, b4, chunk0, ncols, tile, w, q, nn);
}

// ---------------------------------------------------------------------------------------
// signal branch
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sig3_front_kernel(ConvFrontArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KW = 9, KW1 = 5, S = KW * 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15, quad = lane & 3;

    float A[S];
    {
        const float *ap = a.apack + (size_t)w * S * 64 + lane;
#pragma unroll
        for (int s = 0; s < S; ++s) A[s] = ap[(size_t)s * 64];
    }
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + 16 * w + 4 * q);
    // producer weights as channel pairs (front_sig_kernel)
    f32x2 w1[KW1][2];
#pragma unroll
    for (int t = 0; t < KW1; ++t)
#pragma unroll
        for (int o = 0; o < 2; ++o) w1[t][o] = f32x2{a.w_sig1[t * 4 + 2 * o], a.w_sig1[t * 4 + 2 * o + 1]};
    const f32x2 b1lo = f32x2{a.b_sig1[0], a.b_sig1[1]}, b1hi = f32x2{a.b_sig1[2], a.b_sig1[3]};
    f32x2 w2[KW1][4][2];
#pragma unroll
    for (int t = 0; t < KW1; ++t)
#pragma unroll
        for (int ic = 0; ic < 4; ++ic) {
            const float4 v = *reinterpret_cast<const float4 *>(a.w_sig2 + (t * 4 + ic) * 16 + 4 * quad);
            w2[t][ic][0] = f32x2{v.x, v.y};
            w2[t][ic][1] = f32x2{v.z, v.w};
        }
    const f32x2 b2lo = f32x2{a.b_sig2[4 * quad], a.b_sig2[4 * quad + 1]}, b2hi = f32x2{a.b_sig2[4 * quad + 2], a.b_sig2[4 * quad + 3]};

    const int64_t n_iters = (a.n + a.cb - 1) / a.cb;
    // the signal rows of the wave's (up to two) chunks of an iteration are requested an iteration ahead, up to four
    // samples per lane and chunk, and wait in registers under the matrix phase
    const bool pre_ok = a.L <= 256 && a.cb <= 8;
    float pre[2][4];
    auto prefetch = [&](int64_t it) {
        if (!pre_ok || it >= n_iters) return;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int64_t chunk = it * a.cb + w + 4 * k;
            if (w + 4 * k >= a.cb || chunk >= a.n) continue;
            const float *src = a.signal + (size_t)chunk * a.L;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (lane + 64 * j < a.L) pre[k][j] = src[lane + 64 * j];
        }
    };
    prefetch(blockIdx.x);
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        const int64_t chunk0 = it * a.cb;
        const int nch = (int)((a.n - chunk0) < a.cb ? (a.n - chunk0) : a.cb);
        RMR_SYNC();  // the matrix phase of the previous iteration has read the planes
        for (int c = w; c < (CF_ABL(1) ? 0 : nch); c += 4) {
            float *s_sig = smem + a.o_front + (size_t)c * a.per_chunk;
            float *s_sig1 = s_sig + ((a.L + 3) & ~3);
            if (pre_ok) {
                const int k = c >> 2;  // c = w + 4 k
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (lane + 64 * j < a.L) s_sig[lane + 64 * j] = k == 0 ? pre[0][j] : pre[1][j];
            } else {
                const float *src = a.signal + (size_t)(chunk0 + c) * a.L;
                for (int s = lane; s < a.L; s += 64) s_sig[s] = src[s];
            }
            wave_sync();
            for (int pos = lane; pos < a.P1; pos += 64) {
                f32x2 lo = b1lo, hi = b1hi;
#pragma unroll
                for (int t = 0; t < KW1; ++t) {
                    const f32x2 xv = pk_splat(s_sig[pos + t]);
                    lo = pk_fma(w1[t][0], xv, lo);
                    hi = pk_fma(w1[t][1], xv, hi);
                }
                swish_pk(lo, hi);
                *reinterpret_cast<float4 *>(s_sig1 + pos * 4) = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
            wave_sync();
            float *row0 = smem + (size_t)quad * a.plane + (size_t)c * a.pin * 4;  // plane of this lane's channel quad
            // two positions per pass (i and i + 64, same channel quad): two independent multiply-add chains per lane, so
            // that a wave alone covers the latency of its own LDS reads and dependent FMAs
            const int n_items = a.pin * 4;
            for (int i = lane; i < n_items; i += 128) {  // i & 3 == quad
                const bool two = i + 64 < n_items;
                const int pos0 = i >> 2, pos1 = two ? (i + 64) >> 2 : pos0;
                f32x2 lo0 = b2lo, hi0 = b2hi, lo1 = b2lo, hi1 = b2hi;
#pragma unroll
                for (int t = 0; t < KW1; ++t) {
                    const float4 xa = *reinterpret_cast<const float4 *>(s_sig1 + (pos0 + t) * 4);
                    const float4 xb = *reinterpret_cast<const float4 *>(s_sig1 + (pos1 + t) * 4);
                    const float a4[4] = {xa.x, xa.y, xa.z, xa.w}, b4v[4] = {xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                    for (int ic = 0; ic < 4; ++ic) {
                        const f32x2 sa = pk_splat(a4[ic]), sb = pk_splat(b4v[ic]);
                        lo0 = pk_fma(w2[t][ic][0], sa, lo0);
                        hi0 = pk_fma(w2[t][ic][1], sa, hi0);
                        lo1 = pk_fma(w2[t][ic][0], sb, lo1);
                        hi1 = pk_fma(w2[t][ic][1], sb, hi1);
                    }
                }
                swish_pk(lo0, hi0);
                swish_pk(lo1, hi1);
                *reinterpret_cast<float4 *>(row0 + pos0 * 4) = make_float4(lo0.x, lo0.y, hi0.x, hi0.y);
                if (two) *reinterpret_cast<float4 *>(row0 + pos1 * 4) = make_float4(lo1.x, lo1.y, hi1.x, hi1.y);
            }
        }
        prefetch(it + gridDim.x);
        RMR_SYNC();
        if (!CF_ABL(2)) mfma_phase<KW>(a, smem, A, b4, chunk0, nch, w, q, nn);
    }
}

// ---------------------------------------------------------------------------------------
// signal branch with the producer's sig_conv2 on the matrix cores (round 3): sig_conv1 -> sig_conv2 -> LDS planes -> sig_conv3.
// The VALU producer above keeps a lane's 4-channel slice of the sig_conv2 weights in registers (KW1 x 4 x 4 floats: 80 at
// 5 taps - 248 VGPRs for the kernel, two waves per SIMD -, 176 at Conv_w_ref's 11 taps: does not fit next to sig_conv3's
// slice at all).  As an implicit GEMM (M = 16 channels, N = positions, K = (tap, input channel)) one tap is ONE
// v_mfma_f32_16x16x4_f32: A of lane (q, m) = w_sig2[tap][ic = q][oc = m] (KW1 VGPRs, read from the VALU kernel's table), B of
// lane (q, n) = sig1[ic = q][pos + tap] from a channel-planar row in the wave's scratch.  The D fragment - four consecutive
// output channels of one position - is one float4 store into plane q of the image sig_conv3 reads.  An fp32 MFMA is a
// k-ordered fmaf chain (bit-for-bit the VALU loop: same order tap-major, input channel minor), so the activations equal the
// VALU producer's bit for bit.  One wave produces a chunk at a time, all its column tiles advancing together.
// ---------------------------------------------------------------------------------------
template <int KW1, int MAXT>
__global__ __launch_bounds__(256, (KW1 <= 5 ? 3 : 2)) void sig3_front_mfma_kernel(ConvFrontArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KW = 9, S = KW * 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15;

    float A[S];
    {
        const float *ap = a.apack + (size_t)w * S * 64 + lane;
#pragma unroll
        for (int s = 0; s < S; ++s) A[s] = ap[(size_t)s * 64];
    }
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + 16 * w + 4 * q);
    float A2[KW1];
#pragma unroll
    for (int t = 0; t < KW1; ++t) A2[t] = a.w_sig2[(t * 4 + q) * 16 + nn];
    const f32x4 b2 = *reinterpret_cast<const f32x4 *>(a.b_sig2 + 4 * q);
    float w1[KW1][4];
#pragma unroll
    for (int t = 0; t < KW1; ++t)
#pragma unroll
        for (int o = 0; o < 4; ++o) w1[t][o] = a.w_sig1[t * 4 + o];
    const float b1[4] = {a.b_sig1[0], a.b_sig1[1], a.b_sig1[2], a.b_sig1[3]};

    const int Lp = (a.L + 3) & ~3, Pst = (a.P1 + 16 + 3) & ~3;  // planar row stride: a padded last tile reads inside the wave's scratch
    float *s_sig = smem + a.o_front + (size_t)w * a.per_chunk;   // per WAVE here: [Lp] signal, [4][Pst] sig1
    float *s_sig1 = s_sig + Lp;
    for (int i = lane; i < 4 * Pst; i += 64) s_sig1[i] = 0.0f;   // padding positions stay finite
    const int ntiles = (a.pin + 15) >> 4;                        // pin = P2 = positions of sig_conv2's output

    const int64_t n_iters = (a.n + a.cb - 1) / a.cb;
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        const int64_t chunk0 = it * a.cb;
        const int nch = (int)((a.n - chunk0) < a.cb ? (a.n - chunk0) : a.cb);
        RMR_SYNC();  // the matrix phase of the previous iteration has read the planes
        for (int c = w; c < (CF_ABL(1) ? 0 : nch); c += 4) {
            wave_sync();
            const float *src = a.signal + (size_t)(chunk0 + c) * a.L;
            for (int s = lane; s < a.L; s += 64) s_sig[s] = src[s];
            wave_sync();
            for (int pos = lane; pos < a.P1; pos += 64) {
                float acc[4] = {b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
                for (int t = 0; t < KW1; ++t) {
                    const float xv = s_sig[pos + t];
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[o] = fmaf(w1[t][o], xv, acc[o]);
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) s_sig1[o * Pst + pos] = swish_f(acc[o]);
            }
            wave_sync();
            float *row0 = smem + (size_t)q * a.plane + (size_t)c * a.pin * 4;  // plane q = channels 4q..4q+3 of every row
            for (int t0 = 0; t0 < ntiles; t0 += MAXT) {
                f32x4 acc[MAXT];
#pragma unroll
                for (int k = 0; k < MAXT; ++k) acc[k] = b2;
                const float *row = s_sig1 + q * Pst + 16 * t0 + nn;
#pragma unroll
                for (int t = 0; t < KW1; ++t) {
#pragma unroll
                    for (int k = 0; k < MAXT; ++k)
                        if (t0 + k < ntiles)  // wave-uniform
                            acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[t], row[16 * k + t], acc[k], 0, 0, 0);
                }
#pragma unroll
                for (int k = 0; k < MAXT; ++k) {
                    const int pos = 16 * (t0 + k) + nn;
                    if (t0 + k < ntiles && pos < a.pin) {
                        f32x2 lo = f32x2{acc[k][0], acc[k][1]}, hi = f32x2{acc[k][2], acc[k][3]};
                        swish_pk(lo, hi);
                        *reinterpret_cast<float4 *>(row0 + pos * 4) = make_float4(lo.x, lo.y, hi.x, hi.y);
                    }
                }
            }
        }
        RMR_SYNC();
        if (!CF_ABL(2)) mfma_phase<KW>(a, smem, A, b4, chunk0, nch, w, q, nn);
    }
}

// ---------------------------------------------------------------------------------------
// The same kernel with sig_conv3 in minimal form (round 6; k_wino.hip has the algebra and the stride-1 kernels).  Stride 3 = three
// stride-1 phase filters of three taps on d_r[i] = sig2[3 i + r]; each over groups of four outputs as F(4, 3) at 0, +-1, +-2, inf:
// 6 products per 4 outputs where the direct form has 12, the three phases accumulating into the same x-domain accumulators
// (K = 3 x 16 = 48 per point): 126 MFMAs per chunk at C100 instead of 252.  The producer is the one above, untouched; between
// it and the matrix phase the block turns the staged rows into the x-domain image V[x][phase][plane q][column] (column = (chunk,
// group), 16 B each, the columns of a 16-column tile rotated by 4 x phase so that the three phases of a column - written by
// neighbouring lanes - sit on different bank slots): 1.5 x the rows' size, which still fits two blocks per CU at four chunks per
// iteration.  Rows behind a chunk's last enter as zeros.  (seq_conv2's 5-tap phases need F(4, 5): V is twice the rows and does
// not fit beside its producer's gather tables; profiles/NOTES_r06.md section 8.)
// ---------------------------------------------------------------------------------------
template <int KW1, int MAXT>
__global__ __launch_bounds__(256, 2) void sig3_front_wino_kernel(ConvFrontArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NX = 6, S = NX * 12;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15;

    float A[S];
    {
        const float *ap = a.wpack + (size_t)w * S * 64 + lane;
#pragma unroll
        for (int s = 0; s < S; ++s) A[s] = ap[(size_t)s * 64];
    }
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + 16 * w + 4 * q);
    float A2[KW1];
#pragma unroll
    for (int t = 0; t < KW1; ++t) A2[t] = a.w_sig2[(t * 4 + q) * 16 + nn];
    const f32x4 b2 = *reinterpret_cast<const f32x4 *>(a.b_sig2 + 4 * q);
    float w1[KW1][4];
#pragma unroll
    for (int t = 0; t < KW1; ++t)
#pragma unroll
        for (int o = 0; o < 4; ++o) w1[t][o] = a.w_sig1[t * 4 + o];
    const float b1[4] = {a.b_sig1[0], a.b_sig1[1], a.b_sig1[2], a.b_sig1[3]};

    const int Lp = (a.L + 3) & ~3, Pst = (a.P1 + 16 + 3) & ~3;
    float *s_sig = smem + a.o_front + (size_t)w * a.per_chunk;
    float *s_sig1 = s_sig + Lp;
    for (int i = lane; i < 4 * Pst; i += 64) s_sig1[i] = 0.0f;
    const int ntiles = (a.pin + 15) >> 4;
    float *const V = smem + a.o_v;
    const int XIV = 12 * a.vplane;  // one x image of V

    const int64_t n_iters = (a.n + a.cb - 1) / a.cb;
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        const int64_t chunk0 = it * a.cb;
        const int nch = (int)((a.n - chunk0) < a.cb ? (a.n - chunk0) : a.cb);
        RMR_SYNC();  // the matrix phase of the previous iteration has read V
        for (int c = w; c < nch; c += 4) {  // ---- producer: sig_conv1 -> sig_conv2 (matrix cores) -> the rows of chunk c (as in sig3_front_mfma_kernel)
            wave_sync();
            const float *src = a.signal + (size_t)(chunk0 + c) * a.L;
            for (int s = lane; s < a.L; s += 64) s_sig[s] = src[s];
            wave_sync();
            for (int pos = lane; pos < a.P1; pos += 64) {
                float acc[4] = {b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
                for (int t = 0; t < KW1; ++t) {
                    const float xv = s_sig[pos + t];
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[o] = fmaf(w1[t][o], xv, acc[o]);
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) s_sig1[o * Pst + pos] = swish_f(acc[o]);
            }
            wave_sync();
            float *row0 = smem + (size_t)q * a.plane + (size_t)c * a.pin * 4;
            for (int t0 = 0; t0 < ntiles; t0 += MAXT) {
                f32x4 acc[MAXT];
#pragma unroll
                for (int k = 0; k < MAXT; ++k) acc[k] = b2;
                const float *row = s_sig1 + q * Pst + 16 * t0 + nn;
#pragma unroll
                for (int t = 0; t < KW1; ++t) {
#pragma unroll
                    for (int k = 0; k < MAXT; ++k)
                        if (t0 + k < ntiles) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(A2[t], row[16 * k + t], acc[k], 0, 0, 0);
                }
#pragma unroll
                for (int k = 0; k < MAXT; ++k) {
                    const int pos = 16 * (t0 + k) + nn;
                    if (t0 + k < ntiles && pos < a.pin) {
                        f32x2 lo = f32x2{acc[k][0], acc[k][1]}, hi = f32x2{acc[k][2], acc[k][3]};
                        swish_pk(lo, hi);
                        *reinterpret_cast<float4 *>(row0 + pos * 4) = make_float4(lo.x, lo.y, hi.x, hi.y);
                    }
                }
            }
        }
        RMR_SYNC();
        // ---- rows -> V: wave w takes plane w; an item = (column, phase): six rows 12 t + 3 j + phase of the column's chunk
        const int ncols = nch * a.ngrp;
        {
            const float *img = smem + (size_t)w * a.plane;
            float *vq = V + (size_t)w * a.vplane;
            for (int i = lane; i < 3 * ncols; i += 64) {
                const int col = i / 3, ph = i - 3 * col;
                const int c = (int)(((float)col + 0.5f) * a.div_ngrp.inv), t = col - c * a.ngrp;
                const int lim = a.pin - 1 - ph - 12 * t;  // highest row offset 3 j inside the chunk (>= 0)
                const float *rp = img + (size_t)(c * a.pin + 12 * t + ph) * 4;
                f32x4 d[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    d[j] = *reinterpret_cast<const f32x4 *>(rp + (3 * j <= lim ? 3 * j : 0) * 4);
                    if (3 * j > lim) d[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
                // BT d of F(4, 3), natural point order (0, 1, -1, 2, -2, inf)
                f32x4 v[6];
                const f32x4 c4 = {4.0f, 4.0f, 4.0f, 4.0f}, m5 = {-5.0f, -5.0f, -5.0f, -5.0f}, c2 = {2.0f, 2.0f, 2.0f, 2.0f};
                v[0] = __builtin_elementwise_fma(m5, d[2], __builtin_elementwise_fma(c4, d[0], d[4]));
                const f32x4 ea = __builtin_elementwise_fma(c4, d[2], -d[4]), eb = __builtin_elementwise_fma(c4, d[1], -d[3]);
                v[1] = ea + eb;
                v[2] = ea - eb;
                const f32x4 ec = d[4] - d[2], ee = d[3] - d[1];
                v[3] = __builtin_elementwise_fma(c2, ee, ec);
                v[4] = __builtin_elementwise_fma(-c2, ee, ec);
                v[5] = __builtin_elementwise_fma(m5, d[3], __builtin_elementwise_fma(c4, d[1], d[5]));
                float *dst = vq + (size_t)ph * 4 * a.vplane + (size_t)((col & ~15) + ((col + 4 * ph) & 15)) * 4;
#pragma unroll
                for (int x = 0; x < 6; ++x) *reinterpret_cast<f32x4 *>(dst + (size_t)x * XIV) = v[x];
            }
        }
        RMR_SYNC();
        // ---- six GEMMs of K = 48 per column tile; steps = (phase, half of the points); AT m, bias, swish, four stores per column
        if (CF_ABL(2)) continue;
        const int ntl = (ncols + 15) >> 4;
        for (int tile = 0; tile < ntl; ++tile) {
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            f32x4 acc[6] = {b4, zero, zero, zero, zero, b4};  // AT[0][0] = AT[3][5] = 1: the bias of y0 and y3
            const float *r = V + (size_t)q * a.vplane + (size_t)tile * 64;
            f32x4 xv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) xv[k] = *reinterpret_cast<const f32x4 *>(r + (size_t)k * XIV + nn * 4);
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                const int p = st >> 1, x0 = (st & 1) * 3;
                f32x4 yv[3];
                if (st + 1 < 6) {
                    const int p1 = (st + 1) >> 1, x1 = ((st + 1) & 1) * 3;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        yv[k] = *reinterpret_cast<const f32x4 *>(r + (size_t)(x1 + k) * XIV + (size_t)p1 * 4 * a.vplane + ((nn + 4 * p1) & 15) * 4);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        acc[x0 + k] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[((x0 + k) * 3 + p) * 4 + j], xv[k][j], acc[x0 + k], 0, 0, 0);
                if (st + 1 < 6) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) xv[k] = yv[k];
                }
            }
            // pin the software pipeline: the reads of step st + 1 are issued before the MFMAs of step st
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                if (st + 1 < 6) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            }
            const int col = tile * 16 + nn;
            if (col < ncols) {
                const int c = (int)(((float)col + 0.5f) * a.div_ngrp.inv), t = col - c * a.ngrp;
                const f32x4 s12 = acc[1] + acc[2], d12 = acc[1] - acc[2], s34 = acc[3] + acc[4], d34 = acc[3] - acc[4];
                const f32x4 c2 = {2.0f, 2.0f, 2.0f, 2.0f}, c4 = {4.0f, 4.0f, 4.0f, 4.0f}, c8 = {8.0f, 8.0f, 8.0f, 8.0f};
                f32x4 yo[4];
                yo[0] = (acc[0] + s12) + s34;
                yo[1] = __builtin_elementwise_fma(c2, d34, d12) + b4;
                yo[2] = __builtin_elementwise_fma(c4, s34, s12) + b4;
                yo[3] = __builtin_elementwise_fma(c8, d34, d12) + acc[5];
                float *dst = a.out + ((size_t)(chunk0 + c) * a.pout + 4 * t) * a.out_row + a.out_coff + 16 * w + 4 * q;
                const int nvalid = a.pout - 4 * t;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < nvalid) {
                        f32x2 lo = f32x2{yo[i][0], yo[i][1]}, hi = f32x2{yo[i][2], yo[i][3]};
                        swish_pk(lo, hi);
                        *reinterpret_cast<f32x4 *>(dst + (size_t)i * a.out_row) = f32x4{lo.x, lo.y, hi.x, hi.y};
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// sequence branch (two-level gather of k_front.hip's front_seq_kernel<5, false>, 64 lanes per chunk)
// ---------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void seq2_front_kernel(ConvFrontArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KW = 13, KW1 = 5, S = KW * 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15, quad = lane & 3;

    float A[S];
    {
        const float *ap = a.apack + (size_t)w * S * 64 + lane;
#pragma unroll
        for (int s = 0; s < S; ++s) A[s] = ap[(size_t)s * 64];
    }
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + 16 * w + 4 * q);
    constexpr int wt_words = KW1 * K * 80;
    float *s_wt = smem + a.o_front;  // [KW1][K][5][16]
    for (int i = tid; i < wt_words; i += 256) s_wt[i] = a.wt5[i];
    const f32x2 bq_lo = f32x2{a.b_seq1[4 * quad], a.b_seq1[4 * quad + 1]}, bq_hi = f32x2{a.b_seq1[4 * quad + 2], a.b_seq1[4 * quad + 3]};

    const int64_t n_iters = (a.n + a.cb - 1) / a.cb;
    // the mapping / sequence row and the length of the wave's first chunk of an iteration travel one element per lane
    // (rows of up to 64 elements) and are requested an iteration ahead
    const bool pre_ok = a.map_w <= 64 && a.seq_w <= 64;
    int16_t pre_map = 0;
    int8_t pre_seq = 0;
    int pre_len = 0;
    auto prefetch = [&](int64_t it) {
        const int64_t chunk = it * a.cb + w;
        if (!pre_ok || it >= n_iters || chunk >= a.n) return;
        if (lane < a.map_w) pre_map = a.maps[(size_t)chunk * a.map_w + lane];
        if (lane < a.seq_w) pre_seq = a.seqs[(size_t)chunk * a.seq_w + lane];
        pre_len = a.lens[chunk];
    };
    prefetch(blockIdx.x);
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        const int64_t chunk0 = it * a.cb;
        const int nch = (int)((a.n - chunk0) < a.cb ? (a.n - chunk0) : a.cb);
        RMR_SYNC();  // planes free again; (first iteration) gather table visible
        for (int c = w; c < (CF_ABL(1) ? 0 : nch); c += 4) {
            const int64_t chunk = chunk0 + c;
            float *cbase = smem + a.o_front + wt_words + (size_t)c * a.per_chunk;
            int16_t *s_map = reinterpret_cast<int16_t *>(cbase + a.o_map);
            int8_t *s_seq = reinterpret_cast<int8_t *>(cbase + a.o_seq);
            unsigned long long *s_code = reinterpret_cast<unsigned long long *>(cbase + a.o_code);
            int16_t *s_pidx = reinterpret_cast<int16_t *>(cbase + a.o_pidx);
            float *s_u = cbase + a.o_u;  // [(maxlen+1)][KW1][16], row `maxlen` = zeros
            int len;
            if (c == w && pre_ok) {  // the wave's first chunk: its rows left HBM during the previous matrix phase
                len = pre_len;
                if (lane < a.map_w) s_map[lane] = pre_map;
                if (lane < a.seq_w) s_seq[lane] = pre_seq;
            } else {
                len = a.lens[chunk];
                const int16_t *mp = a.maps + (size_t)chunk * a.map_w;
                for (int j = lane; j < a.map_w; j += 64) s_map[j] = mp[j];
                const int8_t *sq = a.seqs + (size_t)chunk * a.seq_w;
                for (int j = lane; j < a.seq_w; j += 64) s_seq[j] = sq[j];
            }
            len = len < 0 ? 0 : (len > a.maxlen ? a.maxlen : len);
            for (int i = lane; i < KW1 * 16; i += 64) s_u[(size_t)a.maxlen * KW1 * 16 + i] = 0.0f;
            for (int s = lane; s < a.L; s += 64) s_pidx[s] = (int16_t)a.maxlen;
            wave_sync();
            // base covering every signal position, written as runs (base p owns [map[p], map[p+1])); positions no base
            // owns keep the zero row `maxlen` (the gather form of the reference's scatter loops, encoded_kmers.pyx:33-44)
            for (int p = lane; p < len; p += 64) {
                const int s0 = max((int)s_map[p], 0), s1 = min((int)s_map[p + 1], a.L);
                for (int s = s0; s < s1; ++s) s_pidx[s] = (int16_t)p;
                unsigned long long wv = 0;
#pragma unroll
                for (int kp = 0; kp < K; ++kp) {
                    const int b = s_seq[p + kp];
                    wv |= (unsigned long long)((b >= 0 && b < 4) ? b : 4) << (3 * kp);
                }
                s_code[p] = wv;
            }
            wave_sync();
            // U[p][tap][oc] = sum over the K k-mer slots of the table rows of base p's k-mer
            const int items = len * KW1 * 4;
            for (int i = lane; i < items; i += 128) {  // i & 3 == quad; two (base, tap) items per pass
                const bool two = i + 64 < items;
                const int pt0 = i >> 2, pt1 = two ? (i + 64) >> 2 : pt0;
                const int p0 = pt0 / KW1, t0 = pt0 - p0 * KW1, p1 = pt1 / KW1, t1 = pt1 - p1 * KW1;
                const unsigned long long wv0 = s_code[p0], wv1 = s_code[p1];
                const float *wt0 = s_wt + (size_t)t0 * K * 80 + 4 * quad, *wt1 = s_wt + (size_t)t1 * K * 80 + 4 * quad;
                // K known at compile time: the 2 K gathers of a pass are all in flight before the first add (two or three
                // waves per SIMD here, not the eight of the standalone front kernel, so the loop must not serialise them)
                float4 va[K], vb[K];
#pragma unroll
                for (int kp = 0; kp < K; ++kp) {
                    va[kp] = *reinterpret_cast<const float4 *>(wt0 + (kp * 5 + (int)((wv0 >> (3 * kp)) & 7ull)) * 16);
                    vb[kp] = *reinterpret_cast<const float4 *>(wt1 + (kp * 5 + (int)((wv1 >> (3 * kp)) & 7ull)) * 16);
                }
                f32x2 lo0 = pk_splat(0.f), hi0 = pk_splat(0.f), lo1 = pk_splat(0.f), hi1 = pk_splat(0.f);
#pragma unroll
                for (int kp = 0; kp < K; ++kp) {
                    lo0 += f32x2{va[kp].x, va[kp].y};
                    hi0 += f32x2{va[kp].z, va[kp].w};
                    lo1 += f32x2{vb[kp].x, vb[kp].y};
                    hi1 += f32x2{vb[kp].z, vb[kp].w};
                }
                *reinterpret_cast<float4 *>(s_u + (size_t)pt0 * 16 + 4 * quad) = make_float4(lo0.x, lo0.y, hi0.x, hi0.y);
                if (two) *reinterpret_cast<float4 *>(s_u + (size_t)pt1 * 16 + 4 * quad) = make_float4(lo1.x, lo1.y, hi1.x, hi1.y);
            }
            wave_sync();
            float *row0 = smem + (size_t)quad * a.plane + (size_t)c * a.pin * 4;
            const int n_pos_items = a.pin * 4;
            for (int i = lane; i < n_pos_items; i += 128) {  // two positions per pass (same channel quad): see sig3_front_kernel
                const bool two = i + 64 < n_pos_items;
                const int pos0 = i >> 2, pos1 = two ? (i + 64) >> 2 : pos0;
                f32x2 lo0 = bq_lo, hi0 = bq_hi, lo1 = bq_lo, hi1 = bq_hi;
                int pa[KW1], pb[KW1];
#pragma unroll
                for (int t = 0; t < KW1; ++t) {
                    pa[t] = s_pidx[pos0 + t];
                    pb[t] = s_pidx[pos1 + t];
                }
                float4 va[KW1], vb[KW1];
#pragma unroll
                for (int t = 0; t < KW1; ++t) {
                    va[t] = *reinterpret_cast<const float4 *>(s_u + ((size_t)pa[t] * KW1 + t) * 16 + 4 * quad);
                    vb[t] = *reinterpret_cast<const float4 *>(s_u + ((size_t)pb[t] * KW1 + t) * 16 + 4 * quad);
                }
#pragma unroll
                for (int t = 0; t < KW1; ++t) {
                    lo0 += f32x2{va[t].x, va[t].y};
                    hi0 += f32x2{va[t].z, va[t].w};
                    lo1 += f32x2{vb[t].x, vb[t].y};
                    hi1 += f32x2{vb[t].z, vb[t].w};
                }
                swish_pk(lo0, hi0);
                swish_pk(lo1, hi1);
                *reinterpret_cast<float4 *>(row0 + pos0 * 4) = make_float4(lo0.x, lo0.y, hi0.x, hi0.y);
                if (two) *reinterpret_cast<float4 *>(row0 + pos1 * 4) = make_float4(lo1.x, lo1.y, hi1.x, hi1.y);
            }
        }
        prefetch(it + gridDim.x);
        RMR_SYNC();
        if (!CF_ABL(2)) mfma_phase<KW>(a, smem, A, b4, chunk0, nch, w, q, nn);
    }
}

// ---------------------------------------------------------------------------------------
// The sequence branch with seq_conv2 in minimal form (round 6; see sig3_front_wino_kernel and k_wino.hip).  13 taps at stride 3 = phase
// filters of 5, 4 and 4 taps, all three taken as F(4, 5) at 0, +-1, +-2, +-1/2, inf (the 4-tap ones with a zero fifth tap): 8 points x
// K = 48, 168 MFMAs per chunk at C100 instead of 364.  V is TWICE the rows here; it fits a half CU at four chunks per iteration only
// because it takes the place of the producer's gather tables and scratch, which are dead once the rows are staged - the 14 KB
// seq_conv1 table is therefore re-read from L2 every iteration (into registers under the matrix phase, into LDS behind the barrier that
// frees V).
// ---------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256, 2) void seq2_front_wino_kernel(ConvFrontArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KW1 = 5, NX = 8, S = NX * 12;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15, quad = lane & 3;

    float A[S];
    {
        const float *ap = a.wpack + (size_t)w * S * 64 + lane;
#pragma unroll
        for (int s = 0; s < S; ++s) A[s] = ap[(size_t)s * 64];
    }
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + 16 * w + 4 * q);
    constexpr int wt_words = KW1 * K * 80;
    float *s_wt = smem + a.o_front;  // [KW1][K][5][16]
    constexpr int WT_PER = (wt_words + 255) / 256;
    float wt_pre[WT_PER];  // this thread's share of the table, fetched under the matrix phase for the next iteration
#pragma unroll
    for (int u = 0; u < WT_PER; ++u) wt_pre[u] = tid + 256 * u < wt_words ? a.wt5[tid + 256 * u] : 0.0f;
    float *const V = smem + a.o_v;
    const int XIV = 12 * a.vplane;
    const f32x2 bq_lo = f32x2{a.b_seq1[4 * quad], a.b_seq1[4 * quad + 1]}, bq_hi = f32x2{a.b_seq1[4 * quad + 2], a.b_seq1[4 * quad + 3]};

    const int64_t n_iters = (a.n + a.cb - 1) / a.cb;
    // the mapping / sequence row and the length of the wave's first chunk of an iteration travel one element per lane
    // (rows of up to 64 elements) and are requested an iteration ahead
    const bool pre_ok = a.map_w <= 64 && a.seq_w <= 64;
    int16_t pre_map = 0;
    int8_t pre_seq = 0;
    int pre_len = 0;
    auto prefetch = [&](int64_t it) {
        const int64_t chunk = it * a.cb + w;
        if (!pre_ok || it >= n_iters || chunk >= a.n) return;
        if (lane < a.map_w) pre_map = a.maps[(size_t)chunk * a.map_w + lane];
        if (lane < a.seq_w) pre_seq = a.seqs[(size_t)chunk * a.seq_w + lane];
        pre_len = a.lens[chunk];
    };
    prefetch(blockIdx.x);
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        const int64_t chunk0 = it * a.cb;
        const int nch = (int)((a.n - chunk0) < a.cb ? (a.n - chunk0) : a.cb);
        RMR_SYNC();  // the matrix phase of the previous iteration has read V: its place is the table's and the scratch's again
#pragma unroll
        for (int u = 0; u < WT_PER; ++u)
            if (tid + 256 * u < wt_words) s_wt[tid + 256 * u] = wt_pre[u];
        RMR_SYNC();  // gather table visible
        for (int c = w; c < nch; c += 4) {
            const int64_t chunk = chunk0 + c;
            float *cbase = smem + a.o_front + wt_words + (size_t)c * a.per_chunk;
            int16_t *s_map = reinterpret_cast<int16_t *>(cbase + a.o_map);
            int8_t *s_seq = reinterpret_cast<int8_t *>(cbase + a.o_seq);
            unsigned long long *s_code = reinterpret_cast<unsigned long long *>(cbase + a.o_code);
            int16_t *s_pidx = reinterpret_cast<int16_t *>(cbase + a.o_pidx);
            float *s_u = cbase + a.o_u;  // [(maxlen+1)][KW1][16], row `maxlen` = zeros
            int len;
            if (c == w && pre_ok) {  // the wave's first chunk: its rows left HBM during the previous matrix phase
                len = pre_len;
                if (lane < a.map_w) s_map[lane] = pre_map;
                if (lane < a.seq_w) s_seq[lane] = pre_seq;
            } else {
                len = a.lens[chunk];
                const int16_t *mp = a.maps + (size_t)chunk * a.map_w;
                for (int j = lane; j < a.map_w; j += 64) s_map[j] = mp[j];
                const int8_t *sq = a.seqs + (size_t)chunk * a.seq_w;
                for (int j = lane; j < a.seq_w; j += 64) s_seq[j] = sq[j];
            }
            len = len < 0 ? 0 : (len > a.maxlen ? a.maxlen : len);
            for (int i = lane; i < KW1 * 16; i += 64) s_u[(size_t)a.maxlen * KW1 * 16 + i] = 0.0f;
            for (int s = lane; s < a.L; s += 64) s_pidx[s] = (int16_t)a.maxlen;
            wave_sync();
            // base covering every signal position, written as runs (base p owns [map[p], map[p+1])); positions no base
            // owns keep the zero row `maxlen` (the gather form of the reference's scatter loops, encoded_kmers.pyx:33-44)
            for (int p = lane; p < len; p += 64) {
                const int s0 = max((int)s_map[p], 0), s1 = min((int)s_map[p + 1], a.L);
                for (int s = s0; s < s1; ++s) s_pidx[s] = (int16_t)p;
                unsigned long long wv = 0;
#pragma unroll
                for (int kp = 0; kp < K; ++kp) {
                    const int b = s_seq[p + kp];
                    wv |= (unsigned long long)((b >= 0 && b < 4) ? b : 4) << (3 * kp);
                }
                s_code[p] = wv;
            }
            wave_sync();
            // U[p][tap][oc] = sum over the K k-mer slots of the table rows of base p's k-mer
            const int items = len * KW1 * 4;
            for (int i = lane; i < items; i += 128) {  // i & 3 == quad; two (base, tap) items per pass
                const bool two = i + 64 < items;
                const int pt0 = i >> 2, pt1 = two ? (i + 64) >> 2 : pt0;
                const int p0 = pt0 / KW1, t0 = pt0 - p0 * KW1, p1 = pt1 / KW1, t1 = pt1 - p1 * KW1;
                const unsigned long long wv0 = s_code[p0], wv1 = s_code[p1];
                const float *wt0 = s_wt + (size_t)t0 * K * 80 + 4 * quad, *wt1 = s_wt + (size_t)t1 * K * 80 + 4 * quad;
                // K known at compile time: the 2 K gathers of a pass are all in flight before the first add (two or three
                // waves per SIMD here, not the eight of the standalone front kernel, so the loop must not serialise them)
                float4 va[K], vb[K];
#pragma unroll
                for (int kp = 0; kp < K; ++kp) {
                    va[kp] = *reinterpret_cast<const float4 *>(wt0 + (kp * 5 + (int)((wv0 >> (3 * kp)) & 7ull)) * 16);
                    vb[kp] = *reinterpret_cast<const float4 *>(wt1 + (kp * 5 + (int)((wv1 >> (3 * kp)) & 7ull)) * 16);
                }
                f32x2 lo0 = pk_splat(0.f), hi0 = pk_splat(0.f), lo1 = pk_splat(0.f), hi1 = pk_splat(0.f);
#pragma unroll
                for (int kp = 0; kp < K; ++kp) {
                    lo0 += f32x2{va[kp].x, va[kp].y};
                    hi0 += f32x2{va[kp].z, va[kp].w};
                    lo1 += f32x2{vb[kp].x, vb[kp].y};
                    hi1 += f32x2{vb[kp].z, vb[kp].w};
                }
                *reinterpret_cast<float4 *>(s_u + (size_t)pt0 * 16 + 4 * quad) = make_float4(lo0.x, lo0.y, hi0.x, hi0.y);
                if (two) *reinterpret_cast<float4 *>(s_u + (size_t)pt1 * 16 + 4 * quad) = make_float4(lo1.x, lo1.y, hi1.x, hi1.y);
            }
            wave_sync();
            float *row0 = smem + (size_t)quad * a.plane + (size_t)c * a.pin * 4;
            const int n_pos_items = a.pin * 4;
            for (int i = lane; i < n_pos_items; i += 128) {  // two positions per pass (same channel quad): see sig3_front_kernel
                const bool two = i + 64 < n_pos_items;
                const int pos0 = i >> 2, pos1 = two ? (i + 64) >> 2 : pos0;
                f32x2 lo0 = bq_lo, hi0 = bq_hi, lo1 = bq_lo, hi1 = bq_hi;
                int pa[KW1], pb[KW1];
#pragma unroll
                for (int t = 0; t < KW1; ++t) {
                    pa[t] = s_pidx[pos0 + t];
                    pb[t] = s_pidx[pos1 + t];
                }
                float4 va[KW1], vb[KW1];
#pragma unroll
                for (int t = 0; t < KW1; ++t) {
                    va[t] = *reinterpret_cast<const float4 *>(s_u + ((size_t)pa[t] * KW1 + t) * 16 + 4 * quad);
                    vb[t] = *reinterpret_cast<const float4 *>(s_u + ((size_t)pb[t] * KW1 + t) * 16 + 4 * quad);
                }
#pragma unroll
                for (int t = 0; t < KW1; ++t) {
                    lo0 += f32x2{va[t].x, va[t].y};
                    hi0 += f32x2{va[t].z, va[t].w};
                    lo1 += f32x2{vb[t].x, vb[t].y};
                    hi1 += f32x2{vb[t].z, vb[t].w};
                }
                swish_pk(lo0, hi0);
                swish_pk(lo1, hi1);
                *reinterpret_cast<float4 *>(row0 + pos0 * 4) = make_float4(lo0.x, lo0.y, hi0.x, hi0.y);
                if (two) *reinterpret_cast<float4 *>(row0 + pos1 * 4) = make_float4(lo1.x, lo1.y, hi1.x, hi1.y);
            }
        }
        prefetch(it + gridDim.x);
        RMR_SYNC();
        // ---- rows -> V (over the table and the scratch): wave w takes plane w; an item = (column, phase): eight rows 12 t + 3 j + phase
        const int ncols = nch * a.ngrp;
        {
            const float *img = smem + (size_t)w * a.plane;
            float *vq = V + (size_t)w * a.vplane;
            for (int i = lane; i < 3 * ncols; i += 64) {
                const int col = i / 3, ph = i - 3 * col;
                const int c = (int)(((float)col + 0.5f) * a.div_ngrp.inv), t = col - c * a.ngrp;
                const int lim = a.pin - 1 - ph - 12 * t;  // highest row offset 3 j inside the chunk (>= 0)
                const float *rp = img + (size_t)(c * a.pin + 12 * t + ph) * 4;
                f32x4 d[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    d[j] = *reinterpret_cast<const f32x4 *>(rp + (3 * j <= lim ? 3 * j : 0) * 4);
                    if (3 * j > lim) d[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
                // BT d of F(4, 5), natural point order (0, 1, -1, 2, -2, 1/2, -1/2, inf); k_wino.hip wino_in_transform has the derivation
                auto f4 = [](float cst, f32x4 x, f32x4 y) { return __builtin_elementwise_fma(f32x4{cst, cst, cst, cst}, x, y); };
                f32x4 v[8];
                const f32x4 e1 = f4(-4.0f, d[2] + d[6], f32x4{17.0f, 17.0f, 17.0f, 17.0f} * d[4]);
                const f32x4 o1 = f4(-4.0f, d[1] + d[5], f32x4{17.0f, 17.0f, 17.0f, 17.0f} * d[3]);
                v[1] = e1 + o1;
                v[2] = e1 - o1;
                const f32x4 e2 = f4(-5.0f, d[4], f4(4.0f, d[6], d[2]));
                const f32x4 o2 = f4(-5.0f, d[3], f4(4.0f, d[5], d[1]));
                v[3] = f4(2.0f, o2, e2);
                v[4] = f4(-2.0f, o2, e2);
                v[0] = f4(-21.0f, d[2] - d[4], f32x4{4.0f, 4.0f, 4.0f, 4.0f} * (d[0] - d[6]));
                const f32x4 e3 = f4(-5.0f, d[4], f4(4.0f, d[2], d[6]));
                const f32x4 o3 = f4(-5.0f, d[3], f4(4.0f, d[1], d[5]));
                v[5] = f4(2.0f, e3, o3);
                v[6] = f4(2.0f, e3, -o3);
                v[7] = f4(21.0f, d[3] - d[5], f32x4{4.0f, 4.0f, 4.0f, 4.0f} * (d[7] - d[1]));
                float *dst = vq + (size_t)ph * 4 * a.vplane + (size_t)((col & ~15) + ((col + 4 * ph) & 15)) * 4;
#pragma unroll
                for (int x = 0; x < 8; ++x) *reinterpret_cast<f32x4 *>(dst + (size_t)x * XIV) = v[x];
            }
        }
        // the table for the next iteration leaves L2 now and lands under the matrix phase
#pragma unroll
        for (int u = 0; u < WT_PER; ++u) wt_pre[u] = tid + 256 * u < wt_words ? a.wt5[tid + 256 * u] : 0.0f;
        RMR_SYNC();
        // ---- eight GEMMs of K = 48 per column tile; steps = (phase, half of the points); AT m, bias, swish, four stores per column
        const int ntl = (ncols + 15) >> 4;
        for (int tile = 0; tile < ntl; ++tile) {
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            f32x4 acc[8] = {b4, zero, zero, zero, zero, zero, zero, b4};  // AT[0][0] = AT[3][7] = 1: the bias of y0 and y3
            const float *r = V + (size_t)q * a.vplane + (size_t)tile * 64;
            f32x4 xv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) xv[k] = *reinterpret_cast<const f32x4 *>(r + (size_t)k * XIV + nn * 4);
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                const int p = st >> 1, x0 = (st & 1) * 4;
                f32x4 yv[4];
                if (st + 1 < 6) {
                    const int p1 = (st + 1) >> 1, x1 = ((st + 1) & 1) * 4;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        yv[k] = *reinterpret_cast<const f32x4 *>(r + (size_t)(x1 + k) * XIV + (size_t)p1 * 4 * a.vplane + ((nn + 4 * p1) & 15) * 4);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        acc[x0 + k] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[((x0 + k) * 3 + p) * 4 + j], xv[k][j], acc[x0 + k], 0, 0, 0);
                if (st + 1 < 6) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) xv[k] = yv[k];
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                if (st + 1 < 6) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
            const int col = tile * 16 + nn;
            if (col < ncols) {
                const int c = (int)(((float)col + 0.5f) * a.div_ngrp.inv), t = col - c * a.ngrp;
                auto f4 = [](float cst, f32x4 x, f32x4 y) { return __builtin_elementwise_fma(f32x4{cst, cst, cst, cst}, x, y); };
                const f32x4 s12 = acc[1] + acc[2], d12 = acc[1] - acc[2], s34 = acc[3] + acc[4], d34 = acc[3] - acc[4];
                const f32x4 s56 = acc[5] + acc[6], d56 = acc[5] - acc[6];
                f32x4 yo[4];
                yo[0] = ((acc[0] + s12) + s34) + s56;
                yo[1] = f4(0.5f, d56, f4(2.0f, d34, d12)) + b4;
                yo[2] = f4(0.25f, s56, f4(4.0f, s34, s12)) + b4;
                yo[3] = f4(0.125f, d56, f4(8.0f, d34, d12)) + acc[7];
                float *dst = a.out + ((size_t)(chunk0 + c) * a.pout + 4 * t) * a.out_row + a.out_coff + 16 * w + 4 * q;
                const int nvalid = a.pout - 4 * t;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < nvalid) {
                        f32x2 lo = f32x2{yo[i][0], yo[i][1]}, hi = f32x2{yo[i][2], yo[i][3]};
                        swish_pk(lo, hi);
                        *reinterpret_cast<f32x4 *>(dst + (size_t)i * a.out_row) = f32x4{lo.x, lo.y, hi.x, hi.y};
                    }
                }
            }
        }
    }
}

}  // namespace

static constexpr size_t CONV_FRONT_MAX_LDS = 156 * 1024;  // of the CU's 160 KB

bool conv_front_supported(const rmr_model *m, int kb, int ka, int seq_w, int map_w) {
    if (m->desc.arch != RMR_ARCH_CONV_LSTM || m->desc.size != 64 || m->nparts != 0 || m->front.kw1 != 5) return false;
    if (m->sig3.ic != 16 || m->sig3.kw != 9 || m->sig3.stride != 3 || m->sig3.oc != 64) return false;
    if (m->seq2.ic != 16 || m->seq2.kw != 13 || m->seq2.stride != 3 || m->seq2.oc != 64) return false;
    if (kb + ka + 1 != m->desc.kmer_len || m->desc.kmer_len != 9) return false;  // the instantiated k-mer length
    if (map_w < 2 || seq_w < map_w - 1 + m->desc.kmer_len - 1) return false;
    // one chunk per block iteration must fit a CU's LDS in both kernels (long chunk contexts / sequences otherwise go
    // through the separate front + conv_mfma kernels): the same sizes launch_conv_front computes
    auto up4 = [](int words) { return (words + 3) & ~3; };
    const size_t sig1 = ((size_t)(4 * (((m->P2 * 4) + 63) & ~63) + 16) + up4(((m->L + 3) & ~3) + m->P1 * 4)) * sizeof(float);
    const int maxlen = map_w - 1;
    const int per_seq = up4(up4((map_w * 2 + 3) / 4) + up4((seq_w + 3) / 4) + up4(maxlen * 2) + up4((m->L * 2 + 3) / 4) + (maxlen + 1) * 5 * 16);
    const size_t seq1 = ((size_t)(4 * (((m->P1 * 4) + 63) & ~63) + 16) + 5 * m->desc.kmer_len * 80 + per_seq) * sizeof(float);
    return sig1 <= CONV_FRONT_MAX_LDS && seq1 <= CONV_FRONT_MAX_LDS;
}

bool sig3_front_mfma_supported(const rmr_model *m) {
    if (m->desc.size != 64 || m->nparts != 0 || (m->front.kw1 != 5 && m->front.kw1 != 11)) return false;
    if (m->sig3.ic != 16 || m->sig3.kw != 9 || m->sig3.stride != 3 || m->sig3.oc != 64) return false;
    const int Lp = (m->L + 3) & ~3, Pst = (m->P1 + 16 + 3) & ~3;
    const size_t one = ((size_t)(4 * (((m->P2 * 4) + 63) & ~63) + 16) + 4 * (size_t)(Lp + 4 * Pst)) * sizeof(float);
    return one <= CONV_FRONT_MAX_LDS && tune_int("RMR_SIG3_MFMA", 1) != 0;
}

// sig_conv1 -> sig_conv2 (matrix cores) -> sig_conv3 of `n` chunks into channels [0, 64) of cat [n][P3][out_row]
int launch_sig3_front_mfma(rmr_model *m, const float *signal, int64_t n, float *cat) {
    rmr_engine *e = m->eng;
    if (n <= 0) return 0;
    const int sz = m->desc.size, kw1 = m->front.kw1;
    ConvFrontArgs a{};
    a.signal = signal; a.w_sig1 = m->front.w_sig1; a.b_sig1 = m->front.b_sig1; a.w_sig2 = m->front.w_sig2; a.b_sig2 = m->front.b_sig2;
    a.L = m->L; a.P1 = m->P1;
    a.out = cat; a.apack = m->sig3.apack; a.bias = m->sig3.bias; a.n = n;
    a.pin = m->P2; a.pout = m->P3; a.out_row = 2 * sz; a.out_coff = 0; a.div_pout = make_fastdiv(m->P3);
    const int Lp = (m->L + 3) & ~3, Pst = (m->P1 + 16 + 3) & ~3;
    a.per_chunk = Lp + 4 * Pst;  // scratch per WAVE
    // sig_conv3 in polyphase Winograd form (sig3_front_wino_kernel) where its x-domain image fits beside the rows; RMR_WINOGRAD=0 and
    // long chunk contexts: the direct form
    bool wino = m->sig3.wpack && tune_int("RMR_WINOGRAD", 1);
    a.wpack = m->sig3.wpack; a.ngrp = (m->P3 + 3) / 4; a.div_ngrp = make_fastdiv(a.ngrp);
    for (int attempt = 0; attempt < 2; ++attempt, wino = false) {
        void (*kern)(ConvFrontArgs) = wino ? (kw1 == 5 ? sig3_front_wino_kernel<5, 6> : sig3_front_wino_kernel<11, 5>)
                                           : (kw1 == 5 ? sig3_front_mfma_kernel<5, 6> : sig3_front_mfma_kernel<11, 5>);
        // blocks per CU by registers, LDS share accordingly; among the chunk counts that fit, the one that fills its tiles best
        static int regs_tab[2][2] = {{0, 0}, {0, 0}};
        int &regs = regs_tab[wino ? 1 : 0][kw1 == 5 ? 0 : 1];
        if (regs == 0) {
            hipFuncAttributes attr;
            regs = hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(kern)) == hipSuccess && attr.numRegs > 0 ? attr.numRegs : 256;
        }
        int resident = 512 / ((regs + 7) & ~7);
        resident = resident < 1 ? 1 : (resident > 4 ? 4 : resident);
        if (wino && resident > 2) resident = 2;  // V is 1.5 x the rows: four chunks per iteration need a half CU's LDS
        size_t budget = (size_t)73728;
        const size_t share = (size_t)160 * 1024 / resident - 512;
        if (share < budget) budget = share;
        // LDS of k chunks per iteration: the four row planes, (Winograd) V = 6 points x 12 (phase, plane) planes of the padded columns, the waves' scratch
        auto plan = [&](int k, int *plane, int *vplane, size_t *need) {
            *plane = ((k * a.pin * 4) + 63) & ~63;
            *vplane = wino ? ((k * a.ngrp + 15) & ~15) * 4 : 0;
            *need = ((size_t)4 * *plane + 16 + (size_t)72 * *vplane + 4 * (size_t)a.per_chunk) * sizeof(float);
        };
        // score of a chunk count = tile fill of the matrix phase x balance of the producer phase (4 waves, one chunk at a time)
        int cb = 0;
        size_t lds = 0;
        double best = -1.0;
        for (int k = 8; k >= 1; --k) {
            int plane, vplane;
            size_t need;
            plan(k, &plane, &vplane, &need);
            if (need > budget && !(k == 1 && !wino && need <= CONV_FRONT_MAX_LDS)) continue;
            if (cb == 0) cb = k;            // the largest count that fits
            if (2 * k < cb) break;          // never below half of it
            const int cols = wino ? k * a.ngrp : k * a.pout;
            const double score = (double)cols / (16.0 * ((cols + 15) / 16)) * (double)k / (4.0 * ((k + 3) / 4));
            if (score > best + 1e-9) { best = score; a.cb = k; lds = need; }
        }
        if (cb == 0 || (wino && a.cb < 3)) {
            // Winograd with fewer than three chunks per iteration (C200: rows + V of ONE chunk fill a half CU) leaves three of the four
            // producing waves idle - 4.5 against the direct form's 3.1 ms per 250 k chunks; the direct form has its own plan for long chunks
            if (wino) continue;
            RMR_FAIL(RMR_ERR_INVALID, "sig3_front: one chunk of %d samples does not fit the LDS", m->L);
        }
        while (a.cb > 1 && (n + a.cb - 1) / a.cb < e->num_cus) a.cb = (a.cb + 1) / 2;  // a small batch spread over the CUs (same bits for any count)
        plan(a.cb, &a.plane, &a.vplane, &lds);
        a.o_v = 4 * a.plane + 16;
        a.o_front = a.o_v + 72 * a.vplane;
        a.abl = abl_int("RMR_CONV_FRONT_ABLATE", 0);  // ignored unless built with -DRMR_TIMING_ABLATIONS
        const int64_t iters = (n + a.cb - 1) / a.cb;
        int64_t grid = (int64_t)e->num_cus * 8;
        if (grid > iters) grid = iters;
        RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(kern)));
        ProfScope ps(e, K_SIG3_FRONT);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, e->stream, a);
        RMR_HIP(hipGetLastError());
        return 0;
    }
    RMR_FAIL(RMR_ERR_INVALID, "sig3_front: no launch plan");
}

// sig_conv3 (+ sig_conv1/2) and seq_conv2 (+ seq_conv1) of `n` chunks into the two halves of cat [n][P3][128]
int launch_conv_front(rmr_model *m, const float *signal, const int8_t *seqs, int seq_w, const int16_t *maps, int map_w,
                      const int16_t *lens, int64_t n, float *cat) {
    rmr_engine *e = m->eng;
    if (n <= 0) return 0;
    const int sz = m->desc.size, K = m->desc.kmer_len;
    const int budget = 73728;
    auto up4 = [](int words) { return (words + 3) & ~3; };
    // The matrix-core producer is the default since round 4: bit-identical to the VALU one, 5.39 against 5.47 ns per chunk at
    // C100 (12.57 against 12.03 at C200), and it is the one that stayed exact next to foreign processes on the same GPU in
    // every run (rmr_math.h, pk_fma).  RMR_SIG3_MFMA=0 selects the VALU producers (the comparand of tests/test_gpu_conv_front.py).
    if (sig3_front_mfma_supported(m)) {
        RMR_TRY(launch_sig3_front_mfma(m, signal, n, cat));
    } else {   // ---- signal branch, VALU producer ----
        ConvFrontArgs a{};
        a.signal = signal; a.w_sig1 = m->front.w_sig1; a.b_sig1 = m->front.b_sig1; a.w_sig2 = m->front.w_sig2; a.b_sig2 = m->front.b_sig2;
        a.L = m->L; a.P1 = m->P1;
        a.out = cat; a.apack = m->sig3.apack; a.bias = m->sig3.bias; a.n = n;
        a.pin = m->P2; a.pout = m->P3; a.out_row = 2 * sz; a.out_coff = 0; a.div_pout = make_fastdiv(m->P3);
        a.per_chunk = up4(((m->L + 3) & ~3) + m->P1 * 4);
        int cb = 8;
        size_t lds = 0;
        for (; cb >= 1; --cb) {
            a.plane = ((cb * a.pin * 4) + 63) & ~63;
            a.o_front = 4 * a.plane + 16;
            lds = ((size_t)a.o_front + (size_t)cb * a.per_chunk) * sizeof(float);
            if (lds <= (size_t)budget) break;
        }
        if (cb < 1) {  // a long chunk context: one chunk per iteration, one block per CU
            cb = 1;
            if (lds > CONV_FRONT_MAX_LDS) RMR_FAIL(RMR_ERR_INVALID, "sig3_front: one chunk of %d samples needs %zu B of LDS", m->L, lds);
        }
        a.cb = cb;
        a.abl = abl_int("RMR_CONV_FRONT_ABLATE", 0);  // ignored unless built with -DRMR_TIMING_ABLATIONS
        const int64_t iters = (n + cb - 1) / cb;
        int64_t grid = (int64_t)e->num_cus * 8;
        if (grid > iters) grid = iters;
        RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(sig3_front_kernel)));
        ProfScope ps(e, K_SIG3_FRONT);
        hipLaunchKernelGGL(sig3_front_kernel, dim3((unsigned)grid), dim3(256), lds, e->stream, a);
        RMR_HIP(hipGetLastError());
    }
    {   // ---- sequence branch ----
        ConvFrontArgs a{};
        a.seqs = seqs; a.maps = maps; a.lens = lens; a.wt5 = m->front.wt5_seq1; a.b_seq1 = m->front.b_seq1;
        a.L = m->L; a.P1 = m->P1; a.seq_w = seq_w; a.map_w = map_w; a.K = K; a.maxlen = map_w - 1;
        a.out = cat; a.apack = m->seq2.apack; a.bias = m->seq2.bias; a.n = n;
        a.pin = m->P1; a.pout = m->P3; a.out_row = 2 * sz; a.out_coff = sz; a.div_pout = make_fastdiv(m->P3);
        int off = 0;
        a.o_map = off; off += up4((map_w * 2 + 3) / 4);
        a.o_seq = off; off += up4((seq_w + 3) / 4);
        a.o_code = off; off += up4(a.maxlen * 2);
        a.o_pidx = off; off += up4((m->L * 2 + 3) / 4);
        a.o_u = off; off += (a.maxlen + 1) * 5 * 16;
        a.per_chunk = up4(off);
        const int wt_words = 5 * K * 80;
        int cb = 8;
        size_t lds = 0;
        // seq_conv2 in polyphase Winograd form (seq2_front_wino_kernel) where at least three chunks per iteration fit a half CU with V in
        // the place of the gather table and the scratch; RMR_WINOGRAD=0, long chunks and small batches: the direct form
        bool wino = false;
        if (m->seq2.wpack && tune_int("RMR_WINOGRAD", 1)) {
            a.wpack = m->seq2.wpack; a.ngrp = (m->P3 + 3) / 4; a.div_ngrp = make_fastdiv(a.ngrp);
            auto plan = [&](int k) {
                a.plane = ((k * a.pin * 4) + 63) & ~63;
                a.vplane = ((k * a.ngrp + 15) & ~15) * 4;
                a.o_front = 4 * a.plane + 16;
                a.o_v = a.o_front;
                return ((size_t)a.o_front + std::max((size_t)wt_words + (size_t)k * a.per_chunk, (size_t)96 * a.vplane)) * sizeof(float);
            };
            for (int k = 8; k >= 3 && !wino; --k)
                if (plan(k) <= (size_t)80 * 1024 - 512) { wino = true; cb = k; }
            if (wino) {
                while (cb > 1 && (n + cb - 1) / cb < e->num_cus) cb = (cb + 1) / 2;  // a small batch spread over the CUs (same bits for any count)
                lds = plan(cb);
            }
        }
        if (wino) {
            a.cb = cb;
            a.abl = 0;
            const int64_t iters = (n + cb - 1) / cb;
            int64_t grid = (int64_t)e->num_cus * 8;
            if (grid > iters) grid = iters;
            RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(seq2_front_wino_kernel<9>)));
            ProfScope ps(e, K_SEQ2_FRONT);
            hipLaunchKernelGGL(seq2_front_wino_kernel<9>, dim3((unsigned)grid), dim3(256), lds, e->stream, a);
            RMR_HIP(hipGetLastError());
            return 0;
        }
        for (; cb >= 1; --cb) {
            a.plane = ((cb * a.pin * 4) + 63) & ~63;
            a.o_front = 4 * a.plane + 16;
            lds = ((size_t)a.o_front + wt_words + (size_t)cb * a.per_chunk) * sizeof(float);
            if (lds <= (size_t)budget) break;
        }
        if (cb < 1) {
            cb = 1;
            if (lds > CONV_FRONT_MAX_LDS) RMR_FAIL(RMR_ERR_INVALID, "seq2_front: max_seq_len %d needs %zu B of LDS", a.maxlen, lds);
        }
        while (cb > 1 && (n + cb - 1) / cb < e->num_cus) {  // a small batch spread over the CUs (same bits for any count)
            cb = (cb + 1) / 2;
            a.plane = ((cb * a.pin * 4) + 63) & ~63;
            a.o_front = 4 * a.plane + 16;
            lds = ((size_t)a.o_front + wt_words + (size_t)cb * a.per_chunk) * sizeof(float);
        }
        a.cb = cb;
        a.abl = abl_int("RMR_CONV_FRONT_ABLATE", 0);  // ignored unless built with -DRMR_TIMING_ABLATIONS
        const int64_t iters = (n + cb - 1) / cb;
        int64_t grid = (int64_t)e->num_cus * 8;
        if (grid > iters) grid = iters;
        RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(seq2_front_kernel<9>)));
        ProfScope ps(e, K_SEQ2_FRONT);
        hipLaunchKernelGGL(seq2_front_kernel<9>, dim3((unsigned)grid), dim3(256), lds, e->stream, a);
        RMR_HIP(hipGetLastError());
    }
    return 0;
}

}  // namespace rmr
