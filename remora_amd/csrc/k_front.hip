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
#include "rmr_math.h"

namespace rmr {

// The 32 threads that own a chunk live in ONE wavefront and every per-chunk LDS region is
// private to them, so phases only need ordering inside the wave: LDS operations of a wave are
// processed in issue order; the fence keeps the compiler from reordering across the phase
// boundary.  No block-wide barrier -> the 4 waves of a block drift apart and hide each
// other's LDS / global latency.
__device__ __forceinline__ void wave_sync() {
    RMR_JITTER_POINT();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// ---------------------------------------------------------------------------------------
// signal branch: sig_conv1 (1->4) -> LDS -> sig_conv2 (4->16).  32 threads per chunk, the
// thread's output-channel quad (tid & 3) is fixed, so its 4-channel slice of the sig_conv2
// weights (KW*4*4 floats) lives in VGPRs.
// ---------------------------------------------------------------------------------------
struct FrontSigArgs {
    const float *signal;  // [n][L]
    const float *w_sig1, *b_sig1, *w_sig2, *b_sig2;
    float *sig2;          // [n][P2][16]
    int64_t n;
    int L, P1, P2, cb;
};

template <int KW>
__global__ __launch_bounds__(256) void front_sig_kernel(FrontSigArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int c = tid >> 5, sub = tid & 31, quad = sub & 3;
    const int Lp = (a.L + 3) & ~3;
    float *s_sig = smem + (size_t)c * Lp;
    float *s_sig1 = smem + (size_t)a.cb * Lp + (size_t)c * a.P1 * 4;

    // weights as channel PAIRS: every multiply-add below is a v_pk_fma_f32 (two output channels per instruction)
    f32x2 w1[KW][2];
#pragma unroll
    for (int t = 0; t < KW; ++t)
#pragma unroll
        for (int o = 0; o < 2; ++o) w1[t][o] = f32x2{a.w_sig1[t * 4 + 2 * o], a.w_sig1[t * 4 + 2 * o + 1]};
    const f32x2 b1lo = f32x2{a.b_sig1[0], a.b_sig1[1]}, b1hi = f32x2{a.b_sig1[2], a.b_sig1[3]};
    f32x2 w2[KW][4][2];  // [tap][ic] -> the 4 oc of this thread's quad as two pairs
#pragma unroll
    for (int t = 0; t < KW; ++t)
#pragma unroll
        for (int ic = 0; ic < 4; ++ic) {
            const float4 v = *reinterpret_cast<const float4 *>(a.w_sig2 + (t * 4 + ic) * 16 + 4 * quad);
            w2[t][ic][0] = f32x2{v.x, v.y};
            w2[t][ic][1] = f32x2{v.z, v.w};
        }
    const f32x2 b2lo = f32x2{a.b_sig2[4 * quad], a.b_sig2[4 * quad + 1]}, b2hi = f32x2{a.b_sig2[4 * quad + 2], a.b_sig2[4 * quad + 3]};

    const int64_t n_iters = (a.n + a.cb - 1) / a.cb;
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        const int64_t chunk = it * a.cb + c;
        const bool live = chunk < a.n;
        wave_sync();
        if (live) {
            const float *src = a.signal + (size_t)chunk * a.L;
            for (int s = sub; s < a.L; s += 32) s_sig[s] = src[s];
        }
        wave_sync();
        if (live) {
            for (int pos = sub; pos < a.P1; pos += 32) {
                f32x2 lo = b1lo, hi = b1hi;
#pragma unroll
                for (int t = 0; t < KW; ++t) {
                    const f32x2 xv = pk_splat(s_sig[pos + t]);
                    lo = pk_fma(w1[t][0], xv, lo);
                    hi = pk_fma(w1[t][1], xv, hi);
                }
                swish_pk(lo, hi);
                *reinterpret_cast<float4 *>(s_sig1 + pos * 4) = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        }
        wave_sync();
        if (live) {
            float *dst = a.sig2 + (size_t)chunk * a.P2 * 16;
            for (int i = sub; i < a.P2 * 4; i += 32) {  // i & 3 == quad
                const int pos = i >> 2;
                f32x2 lo = b2lo, hi = b2hi;
#pragma unroll
                for (int t = 0; t < KW; ++t) {
                    const float4 xv = *reinterpret_cast<const float4 *>(s_sig1 + (pos + t) * 4);
                    const float x4[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                    for (int ic = 0; ic < 4; ++ic) {
                        const f32x2 xs = pk_splat(x4[ic]);
                        lo = pk_fma(w2[t][ic][0], xs, lo);
                        hi = pk_fma(w2[t][ic][1], xs, hi);
                    }
                }
                swish_pk(lo, hi);
                *reinterpret_cast<float4 *>(dst + (size_t)i * 4) = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// sequence branch, two-level gather.  The k-mer at signal position s depends only on the
// base p(s) covering it, so the sum over the K k-mer slots is done once per BASE and tap:
//     U[p][tap][oc] = sum_kp W[oc][4kp + seq[p+kp]][tap]            (<= max_seq_len bases)
//     seq1[pos][oc] = swish(b[oc] + sum_tap U[p(pos+tap)][tap][oc])  (P1 positions)
// ~3x fewer LDS gathers than summing K*KW table rows per position.  The K bases of a window
// are packed 3 bits each into one 64-bit word (value 4 = missing base -> an all-zero table
// row), so the gather is branch-free; p(s) comes from a binary search on the LDS-resident
// mapping row (upper_bound: the gather form of the reference's scatter loops,
// src/remora/encoded_kmers.pyx:33-44); positions outside every base use a zero U row.
// ---------------------------------------------------------------------------------------
struct FrontSeqArgs {
    const int8_t *seqs;    // [n][seq_w]
    const int16_t *maps;   // [n][map_w]
    const int16_t *lens;   // [n]
    const float *wt5;      // [KW][K][5][16]  (row 4 of each slot = zeros)
    const float *b_seq1;   // [16]
    float *seq1;           // [n][P1][16]
    int64_t n;
    int L, seq_w, map_w, K, P1, cb, maxlen;
    int o_map, o_seq, o_code, o_pidx, o_u, per_chunk;  // LDS offsets (in 4-byte words) per chunk
};

// DIRECT = true skips the per-base table U and sums the KW*K gather rows per output position
// (3x the LDS gathers, but no U buffer: with KW = 11 the U rows would cost 15 KB of LDS per chunk
// and leave two waves per CU; the direct form keeps 16+ waves resident).
template <int KW, bool DIRECT>
__global__ __launch_bounds__(256) void front_seq_kernel(FrontSeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int c = tid >> 5, sub = tid & 31, quad = sub & 3;
    float *s_wt = smem;  // [KW][K][5][16]
    const int wt_words = KW * a.K * 80;
    float *cbase = smem + wt_words + (size_t)c * a.per_chunk;
    int16_t *s_map = reinterpret_cast<int16_t *>(cbase + a.o_map);
    int8_t *s_seq = reinterpret_cast<int8_t *>(cbase + a.o_seq);
    unsigned long long *s_code = reinterpret_cast<unsigned long long *>(cbase + a.o_code);
    int16_t *s_pidx = reinterpret_cast<int16_t *>(cbase + a.o_pidx);
    float *s_u = cbase + a.o_u;  // [(maxlen+1)][KW][16], row `maxlen` = zeros

    for (int i = tid; i < wt_words; i += blockDim.x) s_wt[i] = a.wt5[i];
    if (!DIRECT)
        for (int i = sub; i < KW * 16; i += 32) s_u[(size_t)a.maxlen * KW * 16 + i] = 0.0f;
    const f32x2 bq_lo = f32x2{a.b_seq1[4 * quad], a.b_seq1[4 * quad + 1]}, bq_hi = f32x2{a.b_seq1[4 * quad + 2], a.b_seq1[4 * quad + 3]};
    RMR_SYNC();  // gather table visible to every wave

    const int64_t n_iters = (a.n + a.cb - 1) / a.cb;
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        const int64_t chunk = it * a.cb + c;
        const bool live = chunk < a.n;
        int len = 0;
        wave_sync();
        if (live) {
            len = a.lens[chunk];
            len = len < 0 ? 0 : (len > a.maxlen ? a.maxlen : len);
            const int16_t *mp = a.maps + (size_t)chunk * a.map_w;
            for (int j = sub; j < a.map_w; j += 32) s_map[j] = mp[j];
            const int8_t *sq = a.seqs + (size_t)chunk * a.seq_w;
            for (int j = sub; j < a.seq_w; j += 32) s_seq[j] = sq[j];
        }
        wave_sync();
        if (live) {
            // base covering every signal position: p(s) = last p with map[p] <= s < map[len] (the gather form of the
            // reference's scatter loops).  Written as runs - base p owns [map[p], map[p+1]) - which needs no
            // dependent LDS probes; positions no base owns keep the zero row `maxlen`
            for (int s = sub; s < a.L; s += 32) s_pidx[s] = (int16_t)a.maxlen;
            wave_sync();
            for (int p = sub; p < len; p += 32) {
                const int s0 = max((int)s_map[p], 0), s1 = min((int)s_map[p + 1], a.L);
                for (int s = s0; s < s1; ++s) s_pidx[s] = (int16_t)p;
            }
            for (int p = sub; p < len; p += 32) {
                unsigned long long wv = 0;
                for (int kp = 0; kp < a.K; ++kp) {
                    const int b = s_seq[p + kp];
                    wv |= (unsigned long long)((b >= 0 && b < 4) ? b : 4) << (3 * kp);
                }
                s_code[p] = wv;
            }
        }
        wave_sync();
        if (live && !DIRECT) {
            const int items = len * KW * 4;
            for (int i = sub; i < items; i += 32) {  // i & 3 == quad
                const int pt = i >> 2;
                const int p = pt / KW, t = pt - p * KW;
                unsigned long long wv = s_code[p];
                const float *wt = s_wt + (size_t)t * a.K * 80 + 4 * quad;
                f32x2 lo = pk_splat(0.f), hi = pk_splat(0.f);  // gather-adds as v_pk_add_f32: two channels per instruction
                for (int kp = 0; kp < a.K; ++kp) {
                    const int b = (int)(wv & 7ull);
                    wv >>= 3;
                    const float4 v = *reinterpret_cast<const float4 *>(wt + (kp * 5 + b) * 16);
                    lo += f32x2{v.x, v.y};
                    hi += f32x2{v.z, v.w};
                }
                *reinterpret_cast<float4 *>(s_u + (size_t)pt * 16 + 4 * quad) = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        }
        if (!DIRECT) wave_sync();
        if (live) {
            float *dst = a.seq1 + (size_t)chunk * a.P1 * 16;
            for (int i = sub; i < a.P1 * 4; i += 32) {
                const int pos = i >> 2;
                f32x2 lo = bq_lo, hi = bq_hi;
#pragma unroll
                for (int t = 0; t < KW; ++t) {
                    const int p = s_pidx[pos + t];
                    if (DIRECT) {
                        if (p < a.maxlen) {
                            unsigned long long wv = s_code[p];
                            const float *wt = s_wt + (size_t)t * a.K * 80 + 4 * quad;
                            for (int kp = 0; kp < a.K; ++kp) {
                                const int b = (int)(wv & 7ull);
                                wv >>= 3;
                                const float4 v = *reinterpret_cast<const float4 *>(wt + (kp * 5 + b) * 16);
                                lo += f32x2{v.x, v.y};
                                hi += f32x2{v.z, v.w};
                            }
                        }
                        continue;
                    }
                    const float4 v = *reinterpret_cast<const float4 *>(s_u + ((size_t)p * KW + t) * 16 + 4 * quad);
                    lo += f32x2{v.x, v.y};
                    hi += f32x2{v.z, v.w};
                }
                swish_pk(lo, hi);
                *reinterpret_cast<float4 *>(dst + (size_t)i * 4) = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// sequence branch, two-level gather walked TAP BY TAP (Conv_w_ref's 11-tap seq_conv1).  The per-base table
// U[p][tap][oc] of the two-level form costs (max_seq_len + 1) x KW x 64 B of LDS per chunk - 14.8 KB at KW = 11, which left
// four waves per CU and made the kernel latency-bound, so round 1 shipped the DIRECT form for this shape: KW x K = 99
// table gathers per output position, 35.6 k float4 gathers per chunk, LDS-bound at 9.5 ns per chunk (17 % of the
// Conv_w_ref step).  Here the taps are the OUTER loop: for tap t a wave builds only U_t[p][oc] (max_seq_len x 64 B = 1.3 KB),
// every lane adds U_t[p(pos + t)] to the accumulators of its output positions, and the buffer is reused for tap t + 1.
// Same 12 k gathers per chunk as the two-level form, 3.4 KB of private LDS per wave, one wave per chunk, 16 waves per CU.
// Sum order: over the K slots inside a (base, tap), then over the taps in ascending order - the order of
// front_seq_kernel<KW, false>.
// ---------------------------------------------------------------------------------------
template <int KW, int K>
__global__ __launch_bounds__(512, 4) void front_seq_tap_kernel(FrontSeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int MAXJ = 6;  // output items (position, channel quad) per lane: P1 * 4 <= 64 * MAXJ (Conv_w_ref: L = 100, P1 = 90)
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6, quad = lane & 3;
    float *s_wt = smem;  // [KW][K][5][16]
    constexpr int wt_words = KW * K * 80;
    float *cbase = smem + wt_words + (size_t)w * a.per_chunk;
    int16_t *s_map = reinterpret_cast<int16_t *>(cbase + a.o_map);
    int8_t *s_seq = reinterpret_cast<int8_t *>(cbase + a.o_seq);
    unsigned *s_code = reinterpret_cast<unsigned *>(cbase + a.o_code);
    int16_t *s_pidx = reinterpret_cast<int16_t *>(cbase + a.o_pidx);
    float *s_ut = cbase + a.o_u;  // [(maxlen + 1)][16], row `maxlen` = zeros (positions no base owns)
    static_assert(K <= 10, "base codes of a k-mer are packed 3 bits each into 32 bits");

    for (int i = tid; i < wt_words; i += blockDim.x) s_wt[i] = a.wt5[i];
    if (lane < 16) s_ut[(size_t)a.maxlen * 16 + lane] = 0.0f;
    const f32x2 bq_lo = f32x2{a.b_seq1[4 * quad], a.b_seq1[4 * quad + 1]}, bq_hi = f32x2{a.b_seq1[4 * quad + 2], a.b_seq1[4 * quad + 3]};
    RMR_SYNC();  // gather table visible to every wave

    const int n_items = a.P1 * 4;  // (position, quad) pairs of a chunk; item i = lane + 64 j has quad = lane & 3
    for (int64_t chunk = (int64_t)blockIdx.x * nw + w; chunk < a.n; chunk += (int64_t)gridDim.x * nw) {
        int len = a.lens[chunk];
        len = len < 0 ? 0 : (len > a.maxlen ? a.maxlen : len);
        wave_sync();  // the previous chunk's reads of the row buffers are done
        {
            const int16_t *mp = a.maps + (size_t)chunk * a.map_w;
            for (int j = lane; j < a.map_w; j += 64) s_map[j] = mp[j];
            const int8_t *sq = a.seqs + (size_t)chunk * a.seq_w;
            for (int j = lane; j < a.seq_w; j += 64) s_seq[j] = sq[j];
            for (int s = lane; s < a.L; s += 64) s_pidx[s] = (int16_t)a.maxlen;
        }
        wave_sync();
        // base covering every signal position, written as runs (base p owns [map[p], map[p+1])): the gather form of the
        // reference's scatter loops (src/remora/encoded_kmers.pyx:33-44); the K bases of a window as 3-bit codes, 4 = missing
        for (int p = lane; p < len; p += 64) {
            const int s0 = max((int)s_map[p], 0), s1 = min((int)s_map[p + 1], a.L);
            for (int s = s0; s < s1; ++s) s_pidx[s] = (int16_t)p;
            unsigned code = 0;
#pragma unroll
            for (int kp = 0; kp < K; ++kp) {
                const int b = s_seq[p + kp];
                code |= (unsigned)((b >= 0 && b < 4) ? b : 4) << (3 * kp);
            }
            s_code[p] = code;
        }
        wave_sync();
        f32x2 lo[MAXJ], hi[MAXJ];
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) { lo[j] = bq_lo; hi[j] = bq_hi; }
        const int u_items = len * 4;  // (base, quad) pairs
#pragma unroll 1
        for (int t = 0; t < KW; ++t) {
            const float *wt = s_wt + (size_t)t * K * 80 + 4 * quad;
            for (int i = lane; i < u_items; i += 64) {  // i & 3 == quad
                const int p = i >> 2;
                const unsigned code = s_code[p];
                float4 v[K];  // the K gathers of an item are all in flight before the first add
#pragma unroll
                for (int kp = 0; kp < K; ++kp) v[kp] = *reinterpret_cast<const float4 *>(wt + (kp * 5 + (int)((code >> (3 * kp)) & 7u)) * 16);
                f32x2 ul = pk_splat(0.f), uh = pk_splat(0.f);
#pragma unroll
                for (int kp = 0; kp < K; ++kp) {
                    ul += f32x2{v[kp].x, v[kp].y};
                    uh += f32x2{v[kp].z, v[kp].w};
                }
                *reinterpret_cast<float4 *>(s_ut + (size_t)p * 16 + 4 * quad) = make_float4(ul.x, ul.y, uh.x, uh.y);
            }
            wave_sync();  // U_t complete
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                const int i = lane + 64 * j;
                if (i < n_items) {
                    const int p = s_pidx[(i >> 2) + t];
                    const float4 v = *reinterpret_cast<const float4 *>(s_ut + (size_t)p * 16 + 4 * quad);
                    lo[j] += f32x2{v.x, v.y};
                    hi[j] += f32x2{v.z, v.w};
                }
            }
            wave_sync();  // U_t consumed: the buffer takes tap t + 1
        }
        float *dst = a.seq1 + (size_t)chunk * a.P1 * 16;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int i = lane + 64 * j;
            if (i < n_items) {
                swish_pk(lo[j], hi[j]);
                *reinterpret_cast<float4 *>(dst + (size_t)i * 4) = make_float4(lo[j].x, lo[j].y, hi[j].x, hi[j].y);
            }
        }
    }
}

int launch_front(rmr_model *m, hipStream_t st, const float *signal, const int8_t *seqs, int seq_w,
                 const int16_t *maps, int map_w, const int16_t *lens, int kb, int ka, int64_t n,
                 float *sig2, float *seq1) {
    rmr_engine *e = m->eng;
    if (n <= 0) return 0;
    const int K = m->desc.kmer_len;
    const int kw = m->front.kw1;
    if (kw != 5 && kw != 11) RMR_FAIL(RMR_ERR_INVALID, "front kernel width %d unsupported", kw);
    if (sig2) {   // ---- signal branch (skipped when the caller folds it into sig_conv3: launch_sig3_front_mfma) ----
        FrontSigArgs a;
        a.signal = signal; a.w_sig1 = m->front.w_sig1; a.b_sig1 = m->front.b_sig1;
        a.w_sig2 = m->front.w_sig2; a.b_sig2 = m->front.b_sig2; a.sig2 = sig2; a.n = n;
        a.L = m->L; a.P1 = m->P1; a.P2 = m->P2; a.cb = 8;
        const int Lp = (m->L + 3) & ~3;
        const size_t lds = (size_t)a.cb * (Lp + m->P1 * 4) * 4;
        const int64_t iters = (n + a.cb - 1) / a.cb;
        int64_t grid = (int64_t)e->num_cus * 8;
        if (grid > iters) grid = iters;
        ProfScope ps(e, K_FRONT_SIG, st, true);
        if (kw == 5) hipLaunchKernelGGL(front_sig_kernel<5>, dim3((unsigned)grid), dim3(256), lds, st, a);
        else hipLaunchKernelGGL(front_sig_kernel<11>, dim3((unsigned)grid), dim3(256), lds, st, a);
        RMR_HIP(hipGetLastError());
    }
    if (!seq1) return 0;
    if (kb + ka + 1 != K) RMR_FAIL(RMR_ERR_INVALID, "kmer context (%d,%d) != model kmer_len %d", kb, ka, K);
    if (K > 21) RMR_FAIL(RMR_ERR_INVALID, "kmer_len %d > 21 not supported by the fused encode", K);
    FrontSeqArgs a;
    a.seqs = seqs; a.maps = maps; a.lens = lens; a.wt5 = m->front.wt5_seq1; a.b_seq1 = m->front.b_seq1;
    a.seq1 = seq1; a.n = n; a.L = m->L; a.seq_w = seq_w; a.map_w = map_w; a.K = K; a.P1 = m->P1;
    a.maxlen = map_w - 1;
    if (seq_w < a.maxlen + K - 1) RMR_FAIL(RMR_ERR_INVALID, "sequence width %d too small for mapping width %d", seq_w, map_w);
    // per-chunk LDS carve in 4-byte words, every region 16-byte aligned
    auto up4 = [](int words) { return (words + 3) & ~3; };
    int off = 0;
    a.o_map = off; off += up4((map_w * 2 + 3) / 4);
    a.o_seq = off; off += up4((seq_w + 3) / 4);
    a.o_code = off; off += up4(a.maxlen * 2);
    a.o_pidx = off; off += up4((m->L * 2 + 3) / 4);
    // Conv_w_ref's released shape (11 taps, k-mer length 9): the tap-by-tap two-level kernel, one wave per chunk
    if (kw == 11 && K == 9 && m->P1 * 4 <= 64 * 6 && a.maxlen <= 1024) {
        int off = 0;
        a.o_map = off; off += up4((map_w * 2 + 3) / 4);
        a.o_seq = off; off += up4((seq_w + 3) / 4);
        a.o_code = off; off += up4(a.maxlen);
        a.o_pidx = off; off += up4((m->L * 2 + 3) / 4);
        a.o_u = off; off += (a.maxlen + 1) * 16;
        a.per_chunk = up4(off);
        a.cb = 1;
        const int waves = 8;
        const size_t lds = ((size_t)kw * K * 80 + (size_t)waves * a.per_chunk) * 4;
        if (lds <= 80 * 1024) {  // two blocks (16 waves) per CU; longer rows fall through to the forms below
            auto kern = front_seq_tap_kernel<11, 9>;
            RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(kern)));
            int64_t grid = (int64_t)e->num_cus * 4;
            const int64_t need = (n + waves - 1) / waves;
            if (grid > need) grid = need;
            ProfScope ps(e, K_FRONT_SEQ, st, true);
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * waves), lds, st, a);
            RMR_HIP(hipGetLastError());
            return 0;
        }
    }
    const bool direct = kw == 11;
    a.o_u = off; if (!direct) off += (a.maxlen + 1) * kw * 16;
    a.per_chunk = up4(off);
    const size_t fixed = (size_t)kw * K * 80 * 4;
    int cb = 8;
    const size_t lds_cap = (size_t)78 * 1024;
    while (cb > 1 && fixed + (size_t)cb * a.per_chunk * 4 > lds_cap) cb >>= 1;
    const size_t lds = fixed + (size_t)cb * a.per_chunk * 4;
    if (lds > 160 * 1024) RMR_FAIL(RMR_ERR_INVALID, "front_seq: max_seq_len %d needs %zu B of LDS", a.maxlen, lds);
    a.cb = cb;
    auto kern = (kw == 5) ? front_seq_kernel<5, false> : front_seq_kernel<11, true>;
    RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(kern)));
    const int64_t iters = (n + cb - 1) / cb;
    int64_t grid = (int64_t)e->num_cus * (direct ? 8 : 4);
    if (grid > iters) grid = iters;
    ProfScope ps(e, K_FRONT_SEQ, st, true);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(32 * cb), lds, st, a);
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
        RMR_SYNC();
        const float *src = a.enc + (size_t)c * a.EC * a.L;
        for (int i = tid; i < a.EC * a.L; i += blockDim.x) s_x[i] = src[i];
        RMR_SYNC();
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
            acc.x = swish_f(acc.x); acc.y = swish_f(acc.y);
            acc.z = swish_f(acc.z); acc.w = swish_f(acc.w);
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
    RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(seq1_dense_kernel)));
    int64_t grid = (int64_t)e->num_cus * 2;
    if (grid > n) grid = n;
    ProfScope ps(e, K_SEQ1_DENSE);
    hipLaunchKernelGGL(seq1_dense_kernel, dim3((unsigned)grid), dim3(256), lds, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

}  // namespace rmr
