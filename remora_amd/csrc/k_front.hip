// k_front.hip — input side of both networks, straight from the chunk arrays:
//   signal  -> sig_conv1 -> sig_conv2                 (1->4->16 channels, VALU)
//   (sequence, mapping, length) -> seq_conv1          (one-hot k-mer conv as a gather-sum)
// plus the dense variant of seq_conv1 for callers that hand over a materialised tensor.
//
// Replaces, fused: encoded_kmers.compute_encoded_kmer_batch
// (src/remora/encoded_kmers.pyx:13-45) + models/ConvLSTM_w_ref.py:41-42,45
// (models/Conv_w_ref.py:45-46,49).  The one-hot tensor f32[n,4K,L] never exists: for signal
// position s the encode puts a single 1.0 per k-mer slot kp in row 4kp+base, with
// base = seqs[c, p(s)+kp] and p(s) = the base whose [map[p], map[p+1]) contains s, so
//   seq_conv1(onehot)[oc, pos] = b[oc] + sum_{tap,kp} W[oc][4kp + base(pos+tap, kp)][tap]
// (terms with base == -1, or s outside every base, contribute nothing — exactly the
// zero columns of the reference encode).  BatchNorm (eval) is folded into W and b.
//
// Outputs are channel-last: sig2 f32[n][P2][16], seq1 f32[n][P1][16].
#include "rmr_internal.h"

namespace rmr {

__device__ __forceinline__ float swish_ff(float x) { return x * __frcp_rn(1.0f + __expf(-x)); }

struct FrontArgs {
    const float *signal;   // [n][L]
    const int8_t *seqs;    // [n][seq_w]
    const int16_t *maps;   // [n][map_w]
    const int16_t *lens;   // [n]
    const float *w_sig1, *b_sig1, *w_sig2, *b_sig2, *wt_seq1, *b_seq1;
    float *sig2, *seq1;
    int64_t n;
    int L, seq_w, map_w, K, P1, P2, cb;
    int seq_w_pad;  // LDS row of the sequence bytes (multiple of 4)
};

// LDS carve (floats unless noted), all offsets multiples of 4 floats:
//   w_sig2 [KW][4][16] | wt_seq1 [KW][K][4][16] | sig [cb][Lp] | sig1 [cb][P1][4] |
//   pidx (int16) [cb][Lp] | seqrow (int8) [cb][seq_w_pad]
template <int KW>
__global__ __launch_bounds__(256) void front_kernel(FrontArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int Lp = (a.L + 3) & ~3;
    float *s_w2 = smem;
    float *s_wt = s_w2 + KW * 64;
    float *s_sig = s_wt + KW * a.K * 64;
    float *s_sig1 = s_sig + a.cb * Lp;
    int16_t *s_pidx = reinterpret_cast<int16_t *>(s_sig1 + a.cb * a.P1 * 4);
    int8_t *s_seq = reinterpret_cast<int8_t *>(s_pidx + a.cb * Lp);

    for (int i = tid; i < KW * 64; i += blockDim.x) s_w2[i] = a.w_sig2[i];
    if (a.seq1)
        for (int i = tid; i < KW * a.K * 64; i += blockDim.x) s_wt[i] = a.wt_seq1[i];
    float w1[KW][4];
#pragma unroll
    for (int t = 0; t < KW; ++t)
#pragma unroll
        for (int o = 0; o < 4; ++o) w1[t][o] = a.w_sig1[t * 4 + o];
    const float4 b1 = *reinterpret_cast<const float4 *>(a.b_sig1);

    const int64_t n_iters = (a.n + a.cb - 1) / a.cb;
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        const int64_t chunk0 = it * a.cb;
        const int nch = (int)((a.n - chunk0) < a.cb ? (a.n - chunk0) : a.cb);
        __syncthreads();
        // ---- phase 1: stage signal + sequence bytes; p(s) by upper_bound on the mapping --
        for (int i = tid; i < nch * a.L; i += blockDim.x) {
            const int c = i / a.L, s = i - c * a.L;
            s_sig[c * Lp + s] = a.signal[(size_t)(chunk0 + c) * a.L + s];
            if (a.seq1) {
                const int16_t *mp = a.maps + (size_t)(chunk0 + c) * a.map_w;
                const int len = a.lens[chunk0 + c];
                // first index in [0, len] with map[idx] > s
                int lo = 0, hi = len + 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (mp[mid] <= s) lo = mid + 1; else hi = mid;
                }
                const int p = lo - 1;
                s_pidx[c * Lp + s] = (int16_t)((p >= 0 && p < len) ? p : -1);
            }
        }
        if (a.seq1)
            for (int i = tid; i < nch * a.seq_w; i += blockDim.x) {
                const int c = i / a.seq_w, j = i - c * a.seq_w;
                s_seq[c * a.seq_w_pad + j] = a.seqs[(size_t)(chunk0 + c) * a.seq_w + j];
            }
        __syncthreads();
        // ---- phase 2: sig_conv1 (1 -> 4) into LDS ---------------------------------------
        for (int i = tid; i < nch * a.P1; i += blockDim.x) {
            const int c = i / a.P1, pos = i - c * a.P1;
            const float *x = s_sig + c * Lp + pos;
            float4 acc = b1;
#pragma unroll
            for (int t = 0; t < KW; ++t) {
                const float xv = x[t];
                acc.x += w1[t][0] * xv; acc.y += w1[t][1] * xv;
                acc.z += w1[t][2] * xv; acc.w += w1[t][3] * xv;
            }
            acc.x = swish_ff(acc.x); acc.y = swish_ff(acc.y);
            acc.z = swish_ff(acc.z); acc.w = swish_ff(acc.w);
            *reinterpret_cast<float4 *>(s_sig1 + (size_t)(c * a.P1 + pos) * 4) = acc;
        }
        // ---- phase 3a: seq_conv1 as gather-sum (independent of phase 2) -------------------
        if (a.seq1) {
            for (int i = tid; i < nch * a.P1 * 4; i += blockDim.x) {
                const int quad = i & 3, cp = i >> 2;
                const int c = cp / a.P1, pos = cp - c * a.P1;
                float4 acc = *reinterpret_cast<const float4 *>(a.b_seq1 + 4 * quad);
                const int16_t *pp = s_pidx + c * Lp + pos;
                const int8_t *sq = s_seq + c * a.seq_w_pad;
#pragma unroll
                for (int t = 0; t < KW; ++t) {
                    const int p = pp[t];
                    if (p >= 0) {
                        const float *wt = s_wt + (size_t)t * a.K * 64 + 4 * quad;
                        for (int kp = 0; kp < a.K; ++kp) {
                            const int b = sq[p + kp];
                            if (b >= 0) {
                                const float4 wv = *reinterpret_cast<const float4 *>(wt + (kp * 4 + b) * 16);
                                acc.x += wv.x; acc.y += wv.y; acc.z += wv.z; acc.w += wv.w;
                            }
                        }
                    }
                }
                acc.x = swish_ff(acc.x); acc.y = swish_ff(acc.y);
                acc.z = swish_ff(acc.z); acc.w = swish_ff(acc.w);
                *reinterpret_cast<float4 *>(a.seq1 + ((size_t)(chunk0 + c) * a.P1 + pos) * 16 + 4 * quad) = acc;
            }
        }
        __syncthreads();
        // ---- phase 3b: sig_conv2 (4 -> 16) -------------------------------------------------
        for (int i = tid; i < nch * a.P2 * 4; i += blockDim.x) {
            const int quad = i & 3, cp = i >> 2;
            const int c = cp / a.P2, pos = cp - c * a.P2;
            float4 acc = *reinterpret_cast<const float4 *>(a.b_sig2 + 4 * quad);
            const float *x = s_sig1 + (size_t)(c * a.P1 + pos) * 4;
#pragma unroll
            for (int t = 0; t < KW; ++t) {
                const float4 xv = *reinterpret_cast<const float4 *>(x + 4 * t);
                const float *wv = s_w2 + t * 64 + 4 * quad;
                const float4 w0 = *reinterpret_cast<const float4 *>(wv);
                const float4 w1v = *reinterpret_cast<const float4 *>(wv + 16);
                const float4 w2v = *reinterpret_cast<const float4 *>(wv + 32);
                const float4 w3v = *reinterpret_cast<const float4 *>(wv + 48);
                acc.x += w0.x * xv.x + w1v.x * xv.y + w2v.x * xv.z + w3v.x * xv.w;
                acc.y += w0.y * xv.x + w1v.y * xv.y + w2v.y * xv.z + w3v.y * xv.w;
                acc.z += w0.z * xv.x + w1v.z * xv.y + w2v.z * xv.z + w3v.z * xv.w;
                acc.w += w0.w * xv.x + w1v.w * xv.y + w2v.w * xv.z + w3v.w * xv.w;
            }
            acc.x = swish_ff(acc.x); acc.y = swish_ff(acc.y);
            acc.z = swish_ff(acc.z); acc.w = swish_ff(acc.w);
            *reinterpret_cast<float4 *>(a.sig2 + ((size_t)(chunk0 + c) * a.P2 + pos) * 16 + 4 * quad) = acc;
        }
    }
}

int launch_front(rmr_model *m, const float *signal, const int8_t *seqs, int seq_w,
                 const int16_t *maps, int map_w, const int16_t *lens, int kb, int ka, int64_t n,
                 float *sig2, float *seq1) {
    rmr_engine *e = m->eng;
    if (n <= 0) return 0;
    const int K = m->desc.kmer_len;
    if (seq1 && kb + ka + 1 != K) RMR_FAIL(RMR_ERR_INVALID, "kmer context (%d,%d) != model kmer_len %d", kb, ka, K);
    const int kw = m->front.kw1;
    FrontArgs a;
    a.signal = signal; a.seqs = seqs; a.maps = maps; a.lens = lens;
    a.w_sig1 = m->front.w_sig1; a.b_sig1 = m->front.b_sig1; a.w_sig2 = m->front.w_sig2;
    a.b_sig2 = m->front.b_sig2; a.wt_seq1 = m->front.wt_seq1; a.b_seq1 = m->front.b_seq1;
    a.sig2 = sig2; a.seq1 = seq1; a.n = n; a.L = m->L; a.seq_w = seq_w; a.map_w = map_w;
    a.K = K; a.P1 = m->P1; a.P2 = m->P2;
    a.seq_w_pad = (seq_w + 3) & ~3;
    const int Lp = (m->L + 3) & ~3;
    const size_t fixed = (size_t)(kw * 64 + kw * K * 64) * 4;
    const size_t per_chunk = (size_t)Lp * 4 + (size_t)m->P1 * 16 + (size_t)Lp * 2 + a.seq_w_pad;
    int cb = (int)((65536 - fixed) / per_chunk);
    if (cb > 16) cb = 16;
    if (cb < 1) RMR_FAIL(RMR_ERR_INVALID, "front kernel: chunk too large for LDS");
    a.cb = cb;
    const size_t lds = fixed + per_chunk * cb + 64;
    const int64_t iters = (n + cb - 1) / cb;
    int64_t grid = (int64_t)e->num_cus * 4;
    if (grid > iters) grid = iters;
    ProfScope ps(e, K_FRONT);
    if (kw == 5)
        hipLaunchKernelGGL(front_kernel<5>, dim3((unsigned)grid), dim3(256), lds, e->stream, a);
    else if (kw == 11)
        hipLaunchKernelGGL(front_kernel<11>, dim3((unsigned)grid), dim3(256), lds, e->stream, a);
    else
        RMR_FAIL(RMR_ERR_INVALID, "front kernel width %d unsupported", kw);
    RMR_HIP(hipGetLastError());
    return 0;
}

// ---- dense seq_conv1: seqs f32[n][4K][L] (arbitrary values) -> seq1 channel-last ---------
// Replaces models/ConvLSTM_w_ref.py:45 / models/Conv_w_ref.py:49 when the caller passes a
// materialised tensor (the `model(sigs, enc_kmers)` contract, data_chunks.py:528-533).
struct DenseArgs {
    const float *enc;   // [n][EC][L]
    const float *wd;    // [KW][EC][16]  folded weights, oc fastest
    const float *bias;  // [16]
    float *seq1;        // [n][P1][16]
    int64_t n;
    int EC, L, P1, kw;
};

__global__ __launch_bounds__(256) void seq1_dense_kernel(DenseArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_w = smem;                      // [kw][EC][16]
    float *s_x = smem + a.kw * a.EC * 16;   // [EC][L]
    const int tid = threadIdx.x;
    for (int i = tid; i < a.kw * a.EC * 16; i += blockDim.x) s_w[i] = a.wd[i];
    for (int64_t c = blockIdx.x; c < a.n; c += gridDim.x) {
        __syncthreads();
        const float *src = a.enc + (size_t)c * a.EC * a.L;
        for (int i = tid; i < a.EC * a.L; i += blockDim.x) s_x[i] = src[i];
        __syncthreads();
        for (int i = tid; i < a.P1 * 4; i += blockDim.x) {
            const int quad = i & 3, pos = i >> 2;
            float4 acc = *reinterpret_cast<const float4 *>(a.bias + 4 * quad);
            for (int ic = 0; ic < a.EC; ++ic) {
                const float *x = s_x + ic * a.L + pos;
                for (int t = 0; t < a.kw; ++t) {
                    const float xv = x[t];
                    const float4 wv = *reinterpret_cast<const float4 *>(s_w + ((size_t)t * a.EC + ic) * 16 + 4 * quad);
                    acc.x += wv.x * xv; acc.y += wv.y * xv; acc.z += wv.z * xv; acc.w += wv.w * xv;
                }
            }
            acc.x = swish_ff(acc.x); acc.y = swish_ff(acc.y);
            acc.z = swish_ff(acc.z); acc.w = swish_ff(acc.w);
            *reinterpret_cast<float4 *>(a.seq1 + ((size_t)c * a.P1 + pos) * 16 + 4 * quad) = acc;
        }
    }
}

int launch_seq1_dense(rmr_model *m, const float *enc, int64_t n, float *seq1) {
    rmr_engine *e = m->eng;
    if (n <= 0) return 0;
    DenseArgs a;
    a.enc = enc; a.wd = m->front.wt_seq1; a.bias = m->front.b_seq1; a.seq1 = seq1; a.n = n;
    a.EC = 4 * m->desc.kmer_len; a.L = m->L; a.P1 = m->P1; a.kw = m->front.kw1;
    // wt_seq1 layout [kw][K][4][16] == [kw][EC][16] with ic = 4*kp + base: same table
    const size_t lds = ((size_t)a.kw * a.EC * 16 + (size_t)a.EC * a.L) * 4;
    if (lds > 160 * 1024) RMR_FAIL(RMR_ERR_INVALID, "dense seq_conv1 needs %zu B LDS", lds);
    static bool attr_done = false;
    if (!attr_done) {
        RMR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(seq1_dense_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    int64_t grid = (int64_t)e->num_cus * 2;
    if (grid > n) grid = n;
    ProfScope ps(e, K_SEQ1_DENSE);
    hipLaunchKernelGGL(seq1_dense_kernel, dim3((unsigned)grid), dim3(256), lds, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

}  // namespace rmr
