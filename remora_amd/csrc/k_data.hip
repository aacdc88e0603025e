// k_data.hip — the integer / byte side of the hot path (all HBM-bound):
//   E1 encode_kmers   src/remora/encoded_kmers.pyx:13-45
//   T1 trim           src/remora/data_chunks_core.pyx:10-45
//   M1 parse_moves    src/remora/io.py:394-407
//   X1 normalise      src/remora/data_chunks.py:191-197
//   X2/X3 geometry + fill   src/remora/data_chunks.py:425-466, :331-423, :1376-1418
//   label tally       src/remora/validate.py:42-45 (argmax), data_chunks.py:1074-1082
#include "rmr_internal.h"
#include "rmr_geometry.h"

namespace rmr {

// ======================================================================================
// E1: out f32[n][4K][L].  One block iteration = one chunk: the mapping row is expanded to
// p(s) once in LDS (upper_bound, the gather form of the reference's scatter loops), then
// the chunk's 4K*L floats are produced as coalesced 16-byte stores, each lane deriving
// (row, s) from its flat element index.  Algorithmic traffic: 72 B in, 14,400 B out per
// C100 chunk -> pure HBM-write roofline.
// ======================================================================================
struct EncodeArgs {
    const int8_t *seqs;
    const int16_t *maps;
    const int16_t *lens;
    float *out;
    int64_t n;
    int seq_w, map_w, K, L;
    float inv_L;
};

// PF = the wave carries the mapping and sequence rows of its NEXT chunk from HBM in registers (map_w, seq_w <= 64 * PF
// elements): the loads leave before the current chunk's stores, so the `s_waitcnt vmcnt` in front of their use counts
// past the 4K*L/256 stores issued after them - the loop never waits for a store to land.  (The first version bisected
// the mapping row in HBM: every step a dependent global load whose vmcnt(0) also drained the previous chunk's stores.)
// PF = 0: rows of any width, read where they are needed.
// NST = the 16-byte stores a lane issues per chunk, ceil(K L / 64), as a compile-time count: gfx950 has ONE counter for
// loads and stores, retired in order, and only with the store loop unrolled can the compiler wait for "all but the NST
// newest" instead of "all" (NST = 0: any shape, a plain loop, and with PF the conservative wait).
#ifndef RMR_ENCODE_NT
#define RMR_ENCODE_NT 1  // non-temporal stores: 5.05 against 4.97 TB/s in this kernel, although a pure store stream of the same shape
                         // prefers plain ones (5.55 against 5.28 TB/s, tools/ubench/write_bw.hip): the kernel is within 4 % of that stream
#endif
#ifndef RMR_ENCODE_CODE_FORM
#define RMR_ENCODE_CODE_FORM 1  // 0: the per-element select form everywhere (A/B builds)
#endif
template <int PF, int NST>
__global__ __launch_bounds__(256) void encode_kernel(EncodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) int smem_i[];
    // one WAVE per chunk: the wave's LDS slice is private, so only wave-level ordering is needed
    // and the four waves of a block stream their 14 KB of stores independently
    const int Lp = (a.L + 7) & ~7;
    const int mapb = PF ? ((a.map_w * 2 + 15) & ~15) : 0;
    const int per_wave = (Lp * 4 + ((a.seq_w + 15) & ~15) + mapb + 15) & ~15;  // bytes
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    char *base = reinterpret_cast<char *>(smem_i) + (size_t)wv * per_wave;
    int16_t *s_pidx = reinterpret_cast<int16_t *>(base);   // [Lp] covering base of a position (general form)
    unsigned *s_code = reinterpret_cast<unsigned *>(base);  // [Lp] or: its k-mer as 3-bit base codes, 4 = none (code form)
    int8_t *s_seq = reinterpret_cast<int8_t *>(base + Lp * 4);  // [seq_w]
    int16_t *s_map = reinterpret_cast<int16_t *>(base + Lp * 4 + ((a.seq_w + 15) & ~15));  // [map_w] (PF only)
    // code form (chunk lengths that are multiples of 4, k-mers of up to 10 bases - the unrolled shapes always): a
    // float4 of the output is four positions of ONE row (k-mer slot kp, base b), and each element is
    // ((code >> 3 kp) & 7) == b worked out in arithmetic - on gfx950 a v_cndmask_b32 that reads VCC costs 16-19 cycles
    // against 4.5 for plain VALU (tools/ubench/valu_cycles.hip), and the general form below selects per element
    const bool code_form = RMR_ENCODE_CODE_FORM && (NST > 0 || ((a.L & 3) == 0 && a.K <= 10));
    const int total = 4 * a.K * a.L;       // floats per chunk (multiple of 4)
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + wv, n_waves = (int64_t)gridDim.x * 4;
    constexpr int NP = PF ? PF : 1;
    int16_t r_map[NP];
    int8_t r_seq[NP];
    int r_len = 0;
    auto prefetch = [&](int64_t c) {  // (clamped indices instead of predicates: no branch between the loads)
        if (!PF || c >= a.n) return;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int j = lane + 64 * k;
            r_map[k] = a.maps[(size_t)c * a.map_w + (j < a.map_w ? j : a.map_w - 1)];
            r_seq[k] = a.seqs[(size_t)c * a.seq_w + (j < a.seq_w ? j : a.seq_w - 1)];
        }
        r_len = a.lens[c];
    };
    auto to_lds = [&]() {  // the prefetched rows -> this wave's LDS slice
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int j = lane + 64 * k;
            if (j < a.map_w) s_map[j] = r_map[k];
            if (j < a.seq_w) s_seq[j] = r_seq[k];
        }
    };
    int len = 0;
    if (PF) {
        prefetch(wave_id);
        to_lds();
        len = r_len;
    }
    for (int64_t c = wave_id; c < a.n; c += n_waves) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int16_t *mp;
        if (PF) {
            mp = s_map;
            prefetch(c + n_waves);  // consumed at the END of this iteration, behind exactly NST stores
        } else {
            len = a.lens[c];
            mp = a.maps + (size_t)c * a.map_w;
            for (int j = lane; j < a.seq_w; j += 64) s_seq[j] = a.seqs[(size_t)c * a.seq_w + j];
        }
        for (int s = lane; s < a.L; s += 64) {
            int lo = 0, hi = len + 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (mp[mid] <= s) lo = mid + 1; else hi = mid;
            }
            const int p = lo - 1;
            const bool valid = p >= 0 && p < len;
            if (code_form) {
                const unsigned char *sq = reinterpret_cast<const unsigned char *>(s_seq) + (valid ? p : 0);
                unsigned code = 0;
                for (int kp = 0; kp < a.K; ++kp) {
                    const unsigned bb = sq[kp];  // 0..3, or 0xFF (-1: beyond the sequence) and anything else -> 4
                    code |= (bb < 4u ? bb : 4u) << (3 * kp);
                }
                s_code[s] = valid ? code : 0x24924924u;
            } else {
                s_pidx[s] = (int16_t)(valid ? p : -1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float4 *dst = reinterpret_cast<float4 *>(a.out + (size_t)c * total);
        auto put = [&](int f) {
            float v[4];
            const int e0 = 4 * f;
            int row = (int)(((float)e0 + 0.5f) * a.inv_L);
            int s = e0 - row * a.L;
            if (code_form) {
                const uint4 cd = *reinterpret_cast<const uint4 *>(s_code + s);
                const unsigned sh = 3u * (unsigned)(row >> 2), b = (unsigned)row & 3u;
                auto one = [&](unsigned c) {  // 1.0f where the slot holds base b: (x == 0) as ((min(x, 1) - 1) & bits of 1.0f)
                    const unsigned x = ((c >> sh) & 7u) ^ b;
                    unsigned t;  // (in assembly: written in C++ the optimiser turns the expression back into compare + select)
                    asm("v_min_u32 %0, 1, %1" : "=v"(t) : "v"(x));
                    return (t - 1u) & 0x3F800000u;
                };
                typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
                const nt_u32x4 ov = {one(cd.x), one(cd.y), one(cd.z), one(cd.w)};
                if (RMR_ENCODE_NT) __builtin_nontemporal_store(ov, reinterpret_cast<nt_u32x4 *>(dst) + f);
                else reinterpret_cast<nt_u32x4 *>(dst)[f] = ov;
                return;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = s_pidx[s];
                const int kp = row >> 2, b = row & 3;
                v[k] = (p >= 0 && s_seq[p + kp] == b) ? 1.0f : 0.0f;
                if (++s == a.L) { s = 0; ++row; }
            }
            typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
            const nt_f32x4 ov = {v[0], v[1], v[2], v[3]};
            if (RMR_ENCODE_NT) __builtin_nontemporal_store(ov, reinterpret_cast<nt_f32x4 *>(dst) + f);
            else reinterpret_cast<nt_f32x4 *>(dst)[f] = ov;
        };
        if (NST) {
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int f = lane + 64 * i;
                if (i + 1 < NST || f < total / 4) put(f);  // only the last round is partial
            }
        } else {
            for (int f = lane; f < total / 4; f += 64) put(f);
        }
        if (PF) {  // the stores above read LDS when they were built: the slice can take the next chunk's rows
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            to_lds();
            len = r_len;
        }
    }
}

int launch_encode(rmr_engine *e, int kb, int ka, const int8_t *seqs, int seq_w,
                  const int16_t *maps, int map_w, const int16_t *lens, int64_t n, int sig_len,
                  float *out) {
    if (n <= 0) return 0;
    EncodeArgs a;
    a.seqs = seqs; a.maps = maps; a.lens = lens; a.out = out; a.n = n;
    a.seq_w = seq_w; a.map_w = map_w; a.K = kb + ka + 1; a.L = sig_len;
    a.inv_L = 1.0f / (float)sig_len;
    if ((size_t)4 * a.K * sig_len >= (1u << 21)) RMR_FAIL(RMR_ERR_INVALID, "encode: chunk too large");
    const int Lp = (sig_len + 7) & ~7;
    const int wide = seq_w > map_w ? seq_w : map_w;
    const int form = tune_int("RMR_ENCODE_FORM", 2);  // 2: shipped; 1: no unrolled store loop; 0: no row prefetch either (round 2's kernel)
    const int pf = form >= 1 ? (wide <= 64 ? 1 : wide <= 128 ? 2 : wide <= 256 ? 4 : 0) : 0;
    const size_t mapb = pf ? (((size_t)map_w * 2 + 15) & ~(size_t)15) : 0;
    const size_t per_wave = ((size_t)Lp * 4 + ((seq_w + 15) & ~15) + mapb + 15) & ~(size_t)15;
    const size_t lds = per_wave * 4;
    int64_t grid = (int64_t)e->num_cus * 8;
    if (grid > (n + 3) / 4) grid = (n + 3) / 4;
    ProfScope ps(e, K_ENCODE);
    // the shapes of the shipped models' chunk contexts get an unrolled store loop; anything else the plain one
    const int nst = (a.K * sig_len + 63) / 64;
    void (*kern)(EncodeArgs) = pf == 1 ? encode_kernel<1, 0> : pf == 2 ? encode_kernel<2, 0> : pf == 4 ? encode_kernel<4, 0> : encode_kernel<0, 0>;
    if (pf == 1 && form >= 2) {
        if (nst == 15) kern = encode_kernel<1, 15>;       // 9-mer, 100 samples
        else if (nst == 29) kern = encode_kernel<1, 29>;  // 9-mer, 200 samples
        else if (nst == 10) kern = encode_kernel<1, 10>;  // 6-mer, 100 samples
        else if (nst == 19) kern = encode_kernel<1, 19>;  // 6-mer, 200 samples
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

// ======================================================================================
// T1: one thread per chunk, in-row shifts exactly as the reference's loops (rows are a
// few tens of bytes; the work is a byte shuffle bounded by HBM).
// ======================================================================================
__global__ void trim_kernel(int sb, int sa, int cb, int ca, int tsc, int8_t *seqs, int seq_w,
                            int16_t *maps, int map_w, int16_t *lens, int64_t n) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    int16_t *cm = maps + (size_t)c * map_w;
    int8_t *cs = seqs + (size_t)c * seq_w;
    const int16_t cc_width = (int16_t)(cb + ca);
    int sl = lens[c];
    if (sb > cb) {
        int st_clip = 0;
        while (st_clip + 1 < map_w && cm[st_clip + 1] <= 0) st_clip++;
        for (int i = 0; i < sl + 1 - st_clip; ++i) cm[i] = cm[st_clip + i];
        for (int i = 0; i < sl + tsc - st_clip; ++i) cs[i] = cs[i + st_clip];
        sl -= st_clip;
        cm[0] = 0;
    }
    if (sa > ca) {
        while (sl > 1 && cm[sl - 1] >= cc_width) sl--;
        cm[sl] = cc_width;
    }
    lens[c] = (int16_t)sl;
}

int launch_trim(rmr_engine *e, int sb, int sa, int cb, int ca, int tsc, int8_t *seqs, int seq_w,
                int16_t *maps, int map_w, int16_t *lens, int64_t n) {
    if (n <= 0) return 0;
    ProfScope ps(e, K_TRIM);
    hipLaunchKernelGGL(trim_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, sb,
                       sa, cb, ca, tsc, seqs, seq_w, maps, map_w, lens, n);
    RMR_HIP(hipGetLastError());
    return 0;
}

// ======================================================================================
// M1: move table -> query_to_signal.  Stream compaction of the non-zero moves with a
// block-wide ballot/popcount prefix (64-wide wavefronts), one block per table.
// ======================================================================================
__global__ __launch_bounds__(1024) void moves_kernel(const int8_t *mv_tag, int64_t mv_tag_len,
                                                      int64_t sig_len, int reverse, int64_t *q2s,
                                                      int64_t *d_count) {
    __shared__ int wave_tot[16];
    __shared__ long long base_sh;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t stride = mv_tag[0];
    const int64_t nmv = mv_tag_len - 1;
    if (tid == 0) base_sh = 0;
    __syncthreads();
    // pass 1 (only when reversing): total count is needed to place entries from the end
    int64_t total = 0;
    if (reverse) {
        int cnt = 0;
        for (int64_t i = tid; i < nmv; i += blockDim.x) cnt += (mv_tag[1 + i] != 0);
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (lane == 0) wave_tot[wv] = cnt;
        __syncthreads();
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) total += wave_tot[k];
        __syncthreads();
    }
    for (int64_t start = 0; start < nmv; start += blockDim.x) {
        const int64_t i = start + tid;
        const bool nz = (i < nmv) && (mv_tag[1 + i] != 0);
        const unsigned long long bal = __ballot(nz);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(bal);
        __syncthreads();
        int wave_off = 0, blk = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) {
            if (k < wv) wave_off += wave_tot[k];
            blk += wave_tot[k];
        }
        const long long base = base_sh;
        if (nz) {
            const int64_t k = base + wave_off + before;
            if (!reverse) q2s[k] = i * stride;
            else q2s[total - k] = sig_len - i * stride;  // reversed: [0]=sig_len-sig_len .. see below
        }
        __syncthreads();
        if (tid == 0) base_sh = base + blk;
        __syncthreads();
    }
    if (tid == 0) {
        const int64_t cnt = base_sh;
        if (!reverse) q2s[cnt] = sig_len;
        else q2s[0] = 0;  // sig_len - sig_len
        *d_count = cnt + 1;
    }
}

int launch_moves(rmr_engine *e, const int8_t *mv_tag, int64_t mv_tag_len, int64_t sig_len,
                 int reverse, int64_t *q2s, int64_t *d_count) {
    ProfScope ps(e, K_MOVES);
    hipLaunchKernelGGL(moves_kernel, dim3(1), dim3(1024), 0, e->stream, mv_tag, mv_tag_len, sig_len,
                       reverse, q2s, d_count);
    RMR_HIP(hipGetLastError());
    return 0;
}

// batch form: block b expands table b of a concatenation (tables at mv_off[b]..mv_off[b+1], outputs at the
// same offsets: a table of m entries yields at most m coordinates); per-table status as rmr_parse_moves returns it
__global__ __launch_bounds__(1024) void moves_batch_kernel(const int8_t *mv_tags, const int64_t *mv_off,
                                                            const int64_t *sig_len, const int64_t *seq_len, int check,
                                                            int reverse, int64_t *q2s, int64_t *counts, int32_t *status) {
    __shared__ int wave_tot[16];
    __shared__ long long base_sh;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int8_t *mv_tag = mv_tags + mv_off[b];
    const int64_t mv_tag_len = mv_off[b + 1] - mv_off[b];
    int64_t *out = q2s + mv_off[b];
    if (mv_tag_len < 1) {
        if (tid == 0) { counts[b] = 0; status[b] = RMR_ERR_INVALID; }
        return;
    }
    const int64_t stride = mv_tag[0], nmv = mv_tag_len - 1, slen = sig_len[b];
    if (tid == 0) base_sh = 0;
    __syncthreads();
    int64_t total = 0;
    if (reverse) {
        int cnt = 0;
        for (int64_t i = tid; i < nmv; i += blockDim.x) cnt += (mv_tag[1 + i] != 0);
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (lane == 0) wave_tot[wv] = cnt;
        __syncthreads();
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) total += wave_tot[k];
        __syncthreads();
    }
    for (int64_t start = 0; start < nmv; start += blockDim.x) {
        const int64_t i = start + tid;
        const bool nz = (i < nmv) && (mv_tag[1 + i] != 0);
        const unsigned long long bal = __ballot(nz);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(bal);
        __syncthreads();
        int wave_off = 0, blk = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) {
            if (k < wv) wave_off += wave_tot[k];
            blk += wave_tot[k];
        }
        const long long base = base_sh;
        if (nz) {
            const int64_t k = base + wave_off + before;
            if (!reverse) out[k] = i * stride;
            else out[total - k] = slen - i * stride;
        }
        __syncthreads();
        if (tid == 0) base_sh = base + blk;
        __syncthreads();
    }
    if (tid == 0) {
        const int64_t cnt = base_sh;
        if (!reverse) out[cnt] = slen;
        else out[0] = 0;
        counts[b] = cnt + 1;
        int st = 0;
        if (stride <= 0) st = RMR_ERR_INVALID;
        else if (check && seq_len[b] >= 0 && cnt != seq_len[b]) st = RMR_ERR_DISCORDANT_SEQ;
        else if (check && nmv != slen / stride) st = RMR_ERR_DISCORDANT_SIG;
        status[b] = st;
    }
}

int launch_moves_batch(rmr_engine *e, const int8_t *mv_tags, const int64_t *mv_off, const int64_t *sig_len,
                       const int64_t *seq_len, int64_t n, int check, int reverse, int64_t *q2s, int64_t *counts,
                       int32_t *status) {
    if (n <= 0) return 0;
    ProfScope ps(e, K_MOVES);
    hipLaunchKernelGGL(moves_batch_kernel, dim3((unsigned)n), dim3(1024), 0, e->stream, mv_tags, mv_off, sig_len,
                       seq_len, check, reverse, q2s, counts, status);
    RMR_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// Batch tail of io.Read.add_alignment + into_remora_read (src/remora/io.py:2003-2012, 2123-2177): read i keeps the samples
// its move table maps, signal[src_start[i] + q2s_i[0] .. src_start[i] + q2s_i[last]), and its mapping re-based to 0.
// Pass 1 (one thread per read): the kept length; pass 2 (grid = reads x segments): the copies, coalesced.
// ---------------------------------------------------------------------------------------
__global__ void assemble_lengths_kernel(const int64_t *q2s, const int64_t *q2s_off, const int64_t *seq_len, int64_t n,
                                        int64_t *len_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t *q = q2s + q2s_off[i];
    len_out[i] = q[seq_len[i]] - q[0];
}

__global__ __launch_bounds__(256) void assemble_reads_kernel(const int16_t *signal, const int64_t *src_start, const int64_t *q2s,
                                                             const int64_t *q2s_off, const int64_t *sig_off, const int64_t *seq_off,
                                                             int16_t *dacs, int64_t *s2s) {
    const int64_t i = blockIdx.x;
    const int64_t *q = q2s + q2s_off[i];
    const int64_t first = q[0], n_sig = sig_off[i + 1] - sig_off[i], n_map = seq_off[i + 1] - seq_off[i] + 1;
    const int16_t *src = signal + src_start[i] + first;
    int16_t *dst = dacs + sig_off[i];
    const int64_t step = (int64_t)gridDim.y * blockDim.x, t0 = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
    for (int64_t k = t0; k < n_sig; k += step) dst[k] = src[k];
    int64_t *m = s2s + seq_off[i] + i;
    for (int64_t k = t0; k < n_map; k += step) m[k] = q[k] - first;
}

// ---------------------------------------------------------------------------------------
// Reads without sm / sd tags are scaled by the median and the median absolute deviation of their trimmed signal
// (io.Read.compute_pa_to_norm_scaling, src/remora/io.py:1851-1856).  Both are order statistics of a function of the int16
// samples, so the GPU only has to count: per span of the decoded signal its smallest and largest sample, then the histogram
// over that range (a few hundred to a few thousand bins); the float64 arithmetic on the occupied bins stays on the host,
// operation for operation what numpy does on the samples.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void signal_range_kernel(const int16_t *signal, const int64_t *start, const int64_t *len, int32_t *lo,
                                                           int32_t *hi) {
    const int64_t r = blockIdx.x, n = len[r];
    const int16_t *s = signal + start[r];
    int mn = 32767, mx = -32768;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int v = s[i];
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const int a = __shfl_xor(mn, d), b = __shfl_xor(mx, d);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    __shared__ int w_mn[4], w_mx[4];
    if ((threadIdx.x & 63) == 0) { w_mn[threadIdx.x >> 6] = mn; w_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            mn = w_mn[w] < mn ? w_mn[w] : mn;
            mx = w_mx[w] > mx ? w_mx[w] : mx;
        }
        lo[r] = mn;  // an empty span: lo = 32767 > hi = -32768
        hi[r] = mx;
    }
}

constexpr int HIST_LDS_BINS = 8192;

// grid (spans, parts): every block counts its share of the span in LDS when the range fits, then adds its non-empty bins
__global__ __launch_bounds__(256) void signal_hist_kernel(const int16_t *signal, const int64_t *start, const int64_t *len, const int32_t *lo,
                                                          const int64_t *hist_off, unsigned int *hist) {
    __shared__ unsigned int bins[HIST_LDS_BINS];
    const int64_t r = blockIdx.x, n = len[r];
    const int64_t width = hist_off[r + 1] - hist_off[r];
    if (width <= 0 || n <= 0) return;
    const int16_t *s = signal + start[r];
    const int base = lo[r];
    unsigned int *out = hist + hist_off[r];
    const int64_t step = (int64_t)gridDim.y * blockDim.x, t0 = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
    if (width <= HIST_LDS_BINS) {
        for (int k = threadIdx.x; k < (int)width; k += blockDim.x) bins[k] = 0u;
        __syncthreads();
        for (int64_t i = t0; i < n; i += step) atomicAdd(&bins[(int)s[i] - base], 1u);
        __syncthreads();
        for (int k = threadIdx.x; k < (int)width; k += blockDim.x)
            if (bins[k]) atomicAdd(&out[k], bins[k]);
    } else {
        for (int64_t i = t0; i < n; i += step) atomicAdd(&out[(int)s[i] - base], 1u);
    }
}

int launch_signal_range(rmr_engine *e, const int16_t *signal, const int64_t *start, const int64_t *len, int64_t n, int32_t *lo, int32_t *hi) {
    hipLaunchKernelGGL(signal_range_kernel, dim3((unsigned)n), dim3(256), 0, e->stream, signal, start, len, lo, hi);
    RMR_HIP(hipGetLastError());
    return 0;
}

int launch_signal_hist(rmr_engine *e, const int16_t *signal, const int64_t *start, const int64_t *len, const int32_t *lo,
                       const int64_t *hist_off, int64_t n, unsigned int *hist) {
    hipLaunchKernelGGL(signal_hist_kernel, dim3((unsigned)n, 8), dim3(256), 0, e->stream, signal, start, len, lo, hist_off, hist);
    RMR_HIP(hipGetLastError());
    return 0;
}

int launch_assemble_lengths(rmr_engine *e, const int64_t *q2s, const int64_t *q2s_off, const int64_t *seq_len, int64_t n, int64_t *len_out) {
    hipLaunchKernelGGL(assemble_lengths_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, q2s, q2s_off, seq_len, n, len_out);
    RMR_HIP(hipGetLastError());
    return 0;
}

int launch_assemble_reads(rmr_engine *e, const int16_t *signal, const int64_t *src_start, const int64_t *q2s, const int64_t *q2s_off,
                          const int64_t *sig_off, const int64_t *seq_off, int64_t n, int16_t *dacs, int64_t *s2s) {
    hipLaunchKernelGGL(assemble_reads_kernel, dim3((unsigned)n, 16), dim3(256), 0, e->stream, signal, src_start, q2s, q2s_off, sig_off,
                       seq_off, dacs, s2s);
    RMR_HIP(hipGetLastError());
    return 0;
}

// ======================================================================================
// X1 + X2/X3 geometry.  Signal normalisation in float64 then one rounding to float32
// (bit-exact with numpy); per chunk: focus clip, focus signal index, window with clipping,
// the two binary searches on the read's seq_to_signal map.
// ======================================================================================
// 8 consecutive samples per thread: one bisection of sig_off per thread (the read can only move forward inside
// the 8), one 16-byte load and two 16-byte stores when the span is aligned and inside one read.
__global__ __launch_bounds__(256) void normalise_kernel(const int16_t *__restrict__ dacs, const int64_t *__restrict__ sig_off,
                                                        int64_t n_reads, const double *__restrict__ shift,
                                                        const double *__restrict__ scale, float *__restrict__ sig, int64_t total) {
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i0 >= total) return;
    // owning read of the first sample: last r with sig_off[r] <= i0 (reads are concatenated; offsets ascend)
    int64_t lo = 0, hi = n_reads;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (sig_off[mid] <= i0) lo = mid; else hi = mid;
    }
    int64_t r = lo, r_end = sig_off[r + 1];
    double sh = shift[r], sc = scale[r];
    const int nv = (int)min((int64_t)8, total - i0);
    if (nv == 8 && i0 + 8 <= r_end) {  // aligned (i0 is a multiple of 8; the buffers are 16-byte aligned) and one read
        const uint4 raw = *reinterpret_cast<const uint4 *>(dacs + i0);
        const int16_t *d = reinterpret_cast<const int16_t *>(&raw);
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (float)(((double)d[k] - sh) / sc);
        reinterpret_cast<float4 *>(sig + i0)[0] = make_float4(o[0], o[1], o[2], o[3]);
        reinterpret_cast<float4 *>(sig + i0)[1] = make_float4(o[4], o[5], o[6], o[7]);
        return;
    }
    for (int k = 0; k < nv; ++k) {
        const int64_t i = i0 + k;
        while (i >= r_end) { ++r; r_end = sig_off[r + 1]; sh = shift[r]; sc = scale[r]; }  // empty reads are skipped too
        sig[i] = (float)(((double)dacs[i] - sh) / sc);
    }
}

struct GeoArgs {
    rmr_reads d;
    const int32_t *chunk_read;
    int64_t *geo;
    int *max_seq_len;
    int64_t n_chunks;
};

__global__ void geometry_kernel(GeoArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int sl_i = 0;
    if (i < a.n_chunks) {
        const int r = a.chunk_read[i];
        const int64_t nb = a.d.seq_off[r + 1] - a.d.seq_off[r];
        const int64_t *map = a.d.seq_to_sig + a.d.seq_off[r] + r;  // nb + 1 entries
        const int64_t sig_len = a.d.sig_off[r + 1] - a.d.sig_off[r];
        // (rmr_geometry.h: the same function runs on the host for the single-read entry, rmr_call_read)
        sl_i = (int)chunk_geometry_row(map, nb, sig_len, a.d.focus_bases[i], a.d.base_start_justify, a.d.offset, a.d.cc_before,
                                       a.d.cc_after, a.geo + i * 6);
    }
    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(sl_i, o); sl_i = v > sl_i ? v : sl_i; }
    if ((threadIdx.x & 63) == 0) atomicMax(a.max_seq_len, sl_i);
}

int launch_geometry(rmr_engine *e, const rmr_reads &d, int64_t n_chunks, const int32_t *chunk_read,
                    float *sig_out, int64_t total_sig, const int32_t *sig_read, int64_t *geo,
                    int *d_max_seq_len) {
    if (total_sig > 0) {
        ProfScope ps(e, K_NORMALISE);
        hipLaunchKernelGGL(normalise_kernel, dim3((unsigned)((total_sig + 2047) / 2048)), dim3(256), 0,
                           e->stream, d.dacs, d.sig_off, d.n_reads, d.shift, d.scale, sig_out, total_sig);
        RMR_HIP(hipGetLastError());
    }
    if (n_chunks > 0) {
        GeoArgs a{d, chunk_read, geo, d_max_seq_len, n_chunks};
        ProfScope ps(e, K_GEOMETRY);
        hipLaunchKernelGGL(geometry_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0,
                           e->stream, a);
        RMR_HIP(hipGetLastError());
    }
    return 0;
}

// ======================================================================================
// X3/X6 fill: one wavefront per chunk writes the dataset-layout rows: signal window with
// zero padding (coalesced gather), mapping shifted to the chunk with forced ends, context
// sequence with -1 outside the read.
// ======================================================================================
struct FillArgs {
    rmr_reads d;
    const int32_t *chunk_read;
    const float *sig;
    const int64_t *geo;
    float *signal;
    int8_t *seqs;
    int16_t *maps;
    int16_t *lens;
    int64_t *rfb;
    int64_t n_chunks;
    int seq_w, map_w;
};

__global__ __launch_bounds__(256) void fill_kernel(FillArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= a.n_chunks) return;
    const int r = a.chunk_read[i];
    const int64_t nb = a.d.seq_off[r + 1] - a.d.seq_off[r];
    const int64_t *map = a.d.seq_to_sig + a.d.seq_off[r] + r;
    const int8_t *iseq = a.d.int_seq + a.d.seq_off[r];
    const float *rsig = a.sig + a.d.sig_off[r];
    const int64_t sig_len = a.d.sig_off[r + 1] - a.d.sig_off[r];
    const int64_t *g = a.geo + i * 6;
    const int64_t sl = g[0], seq_start = g[4], sig_start0 = g[5];
    const int L = a.d.cc_before + a.d.cc_after;
    // signal: chunk sample j <- read sample sig_start0 + j, zero outside [0, sig_len)
    float *so = a.signal + (size_t)i * L;
    for (int j = lane; j < L; j += 64) {
        const int64_t src = sig_start0 + j;
        so[j] = (src >= 0 && src < sig_len) ? rsig[src] : 0.0f;
    }
    // mapping: map[seq_start + k] - sig_start0 (== - (sig_start - seq_to_sig_offset)); ends forced
    int16_t *mo = a.maps + (size_t)i * a.map_w;
    for (int k = lane; k < a.map_w; k += 64) {
        int v = 0;
        if (k <= sl) {
            v = (int)(map[seq_start + k] - sig_start0);
            if (k == 0) v = 0;
            if (k == sl) v = L;
        }
        mo[k] = (int16_t)v;
    }
    int8_t *qo = a.seqs + (size_t)i * a.seq_w;
    const int64_t nctx = sl + a.d.kb + a.d.ka;
    for (int k = lane; k < a.seq_w; k += 64) {
        int8_t v = -1;
        if (k < nctx) {
            const int64_t src = seq_start - a.d.kb + k;
            if (src >= 0 && src < nb) v = iseq[src];
        }
        qo[k] = v;
    }
    if (lane == 0) {
        a.lens[i] = (int16_t)sl;
        a.rfb[i] = g[3];
    }
}

int launch_fill(rmr_engine *e, const rmr_reads &d, int64_t n_chunks, const int32_t *chunk_read,
                const float *sig, const int64_t *geo, float *signal, int8_t *seqs, int seq_w,
                int16_t *maps, int map_w, int16_t *lens, int64_t *rfb) {
    if (n_chunks <= 0) return 0;
    FillArgs a{d, chunk_read, sig, geo, signal, seqs, maps, lens, rfb, n_chunks, seq_w, map_w};
    ProfScope ps(e, K_FILL);
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

// read index of every chunk from the chunk offsets of the reads (device arrays): chunk i belongs to the read r with
// focus_off[r] <= i < focus_off[r + 1].  Was a host loop + upload + stream synchronisation in front of every geometry and
// fill launch of a device-resident batch.
__global__ void chunk_read_kernel(const int64_t *__restrict__ focus_off, int64_t n_reads, int64_t n_chunks, int32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chunks) return;
    int64_t lo = 0, hi = n_reads;  // last r with focus_off[r] <= i
    while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (focus_off[mid] <= i) lo = mid; else hi = mid; }
    out[i] = (int32_t)lo;
}

int launch_chunk_read(rmr_engine *e, const int64_t *focus_off, int64_t n_reads, int64_t n_chunks, int32_t *out) {
    if (n_chunks <= 0) return 0;
    hipLaunchKernelGGL(chunk_read_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, e->stream, focus_off, n_reads, n_chunks, out);
    RMR_HIP(hipGetLastError());
    return 0;
}

// ======================================================================================
// label tally: argmax (first maximum) histogram, block-level LDS bins then one atomic per bin
// ======================================================================================
__global__ __launch_bounds__(256) void count_kernel(const float *logits, int64_t n, int num_out,
                                                     unsigned long long *counts) {
    __shared__ unsigned int bins[16];
    if (threadIdx.x < 16) bins[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float *p = logits + i * num_out;
        int best = 0;
        float bv = p[0];
        for (int o = 1; o < num_out; ++o) {
            const float v = p[o];
            if (v > bv) { bv = v; best = o; }
        }
        atomicAdd(&bins[best], 1u);
    }
    __syncthreads();
    if (threadIdx.x < num_out && bins[threadIdx.x])
        atomicAdd(&counts[threadIdx.x], (unsigned long long)bins[threadIdx.x]);
}

// ---- M3: motif scan ------------------------------------------------------------------------
// 16 consecutive bases per thread: one bisection of seq_off per thread (the read only moves forward inside the 16),
// 16 flags leave as one 16-byte store when aligned.  HBM traffic: 1 B read (+ the motif window from cache) and 1 B
// written per base.
__global__ __launch_bounds__(256) void motif_kernel(const int8_t *__restrict__ seq, const int64_t *__restrict__ seq_off,
                                                    int n_reads, int64_t total, rmr_motif_set ms,
                                                    uint8_t *__restrict__ flags) {
    const int64_t b0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (b0 >= total) return;
    int lo = 0, hi = n_reads - 1;  // last read with seq_off[r] <= b0
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seq_off[mid] <= b0) lo = mid; else hi = mid - 1;
    }
    int r = lo;
    int64_t rs = seq_off[r], re = seq_off[r + 1];
    const int nv = (int)min((int64_t)16, total - b0);
    uint8_t out[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        out[q] = 0;
        if (q >= nv) continue;
        const int64_t b = b0 + q;
        while (b >= re) { ++r; rs = re; re = seq_off[r + 1]; }  // empty reads are skipped too
        uint8_t hit = 0;
        for (int m = 0; m < ms.n_motifs && !hit; ++m) {
            const int64_t j = b - ms.focus_pos[m];  // start of the window whose focus base is b
            if (j < rs || j + ms.len[m] > re) continue;
            bool ok = true;
            for (int k = 0; k < ms.len[m] && ok; ++k) {
                const int c = seq[j + k];
                ok = c >= 0 && ((ms.mask[m][k] >> c) & 1);
            }
            hit = ok;
        }
        out[q] = hit;
    }
    if (nv == 16 && (reinterpret_cast<uintptr_t>(flags + b0) & 15) == 0) {
        *reinterpret_cast<uint4 *>(flags + b0) = *reinterpret_cast<const uint4 *>(out);
    } else {
        for (int q = 0; q < nv; ++q) flags[b0 + q] = out[q];
    }
}

int launch_motif(rmr_engine *e, const int8_t *seq, const int64_t *seq_off, int n_reads, int64_t total,
                 const rmr_motif_set &ms, uint8_t *flags) {
    if (total <= 0) return 0;
    ProfScope ps(e, K_MOTIF);
    hipLaunchKernelGGL(motif_kernel, dim3((unsigned)((total + 4095) / 4096)), dim3(256), 0, e->stream, seq, seq_off,
                       n_reads, total, ms, flags);
    RMR_HIP(hipGetLastError());
    return 0;
}

// ---- M3 for batches, without a flag array: one wavefront per read walks its bases 64 at a time, tests the motifs,
// and either counts the hits (FILL = false) or writes their read-local positions in ascending order at
// focus[foc_off[read] ...] (FILL = true; wave ballot + prefix popcount compaction).  Replaces the
// nonzero / searchsorted / bincount passes over a 1-byte-per-base flag array of the first version.
template <bool FILL>
__global__ __launch_bounds__(256) void motif_focus_kernel(const int8_t *__restrict__ seq, const int64_t *__restrict__ seq_off,
                                                          int n_reads, rmr_motif_set ms, int64_t *__restrict__ counts,
                                                          const int64_t *__restrict__ foc_off, int64_t *__restrict__ focus) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_reads) return;
    const int64_t rs = seq_off[r], re = seq_off[r + 1];
    int64_t base = FILL ? foc_off[r] : 0, found = 0;
    auto test = [&](int64_t b) -> bool {
        if (b >= re) return false;
        for (int m = 0; m < ms.n_motifs; ++m) {
            const int64_t j = b - ms.focus_pos[m];  // start of the window whose focus base is b
            if (j < rs || j + ms.len[m] > re) continue;
            bool ok = true;
            for (int k = 0; k < ms.len[m] && ok; ++k) {
                const int c = seq[j + k];
                ok = c >= 0 && ((ms.mask[m][k] >> c) & 1);
            }
            if (ok) return true;
        }
        return false;
    };
    // four 64-base windows per round: their loads and tests are independent, so the wave has four of them in flight instead
    // of one (a read is one wave's serial walk: 79 rounds for 5 kb were 0.15 ms per launch of 512 reads, latency all of it)
    for (int64_t b0 = rs; b0 < re; b0 += 256) {
        bool hit[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) hit[u] = test(b0 + 64 * u + lane);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long mask = __ballot(hit[u]);
            if (FILL && hit[u]) focus[base + found + __popcll(mask & ((1ull << lane) - 1ull))] = b0 + 64 * u + lane - rs;
            found += __popcll(mask);
        }
    }
    if (!FILL && lane == 0) counts[r] = found;
}

int launch_motif_focus(rmr_engine *e, const int8_t *seq, const int64_t *seq_off, int n_reads, const rmr_motif_set &ms,
                       int64_t *counts, const int64_t *foc_off, int64_t *focus) {
    if (n_reads <= 0) return 0;
    ProfScope ps(e, K_MOTIF);
    if (focus) hipLaunchKernelGGL(motif_focus_kernel<true>, dim3((unsigned)((n_reads + 3) / 4)), dim3(256), 0, e->stream, seq, seq_off,
                                  n_reads, ms, counts, foc_off, focus);
    else hipLaunchKernelGGL(motif_focus_kernel<false>, dim3((unsigned)((n_reads + 3) / 4)), dim3(256), 0, e->stream, seq, seq_off,
                            n_reads, ms, counts, foc_off, focus);
    RMR_HIP(hipGetLastError());
    return 0;
}

// ---- validation tally (src/remora/validate.py:42-66, :69-99, :208-259 for one batch): per chunk the call, the winning
// probability and the cross entropy; per batch the confusion counts and the loss sum.  The model's km output columns are
// widened to the dataset's kf label columns first (add_unmodeled_labels: columns the model does not predict get -1000),
// softmax in float32 in the reference's order of operations (util.softmax_axis1 on the float32 logits: subtract the row
// maximum, exp, divide by the sequential sum), first maximum wins (np.argmax), cross entropy accumulated in double.
struct TallyArgs {
    const float *logits;      // [n][km]
    const int64_t *labels;    // [n]
    int64_t n;
    int km, kf;
    int colmap[16];           // full column -> model column, -1 = not modelled
    unsigned long long *conf; // [kf][kf] true x called, incremented
    float *win;               // [n] winning probability
    uint8_t *pred;            // [n] call
    double *loss_sum;         // [1] incremented by the batch's sum of cross entropies
};

__global__ __launch_bounds__(256) void validation_tally_kernel(TallyArgs a) {
    __shared__ unsigned int bins[256];
    __shared__ double loss_part[4];
    bins[threadIdx.x] = 0;
    __syncthreads();
    double loss = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float *p = a.logits + i * a.km;
        float v[16];
        float mx = -3.0e38f;
        int best = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < a.kf) {
                v[c] = a.colmap[c] >= 0 ? p[a.colmap[c]] : -1000.0f;
                if (v[c] > mx) { mx = v[c]; best = c; }
            }
        }
        float s = 0.0f, eb = 0.0f;
        double sd = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < a.kf) {
                const float e = expf(v[c] - mx);
                s += e;
                sd += exp((double)v[c] - (double)mx);
                if (c == best) eb = e;
            }
        }
        a.win[i] = eb / s;
        a.pred[i] = (uint8_t)best;
        int lab = (int)a.labels[i];
        lab = lab < 0 ? 0 : (lab >= a.kf ? a.kf - 1 : lab);
        float vl = v[0];
#pragma unroll
        for (int c = 1; c < 16; ++c)
            if (c == lab) vl = v[c];
        loss += (double)mx + log(sd) - (double)vl;
        atomicAdd(&bins[lab * a.kf + best], 1u);
    }
    // loss: lanes -> wave -> block -> one atomic per block
    for (int off = 32; off > 0; off >>= 1) loss += __shfl_down(loss, off);
    if ((threadIdx.x & 63) == 0) loss_part[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.loss_sum, loss_part[0] + loss_part[1] + loss_part[2] + loss_part[3]);
    if (threadIdx.x < a.kf * a.kf && bins[threadIdx.x]) atomicAdd(&a.conf[threadIdx.x], (unsigned long long)bins[threadIdx.x]);
}

int launch_validation_tally(rmr_engine *e, const float *logits, const int64_t *labels, int64_t n, int km, int kf, const int *colmap,
                            int64_t *conf, float *win, uint8_t *pred, double *loss_sum) {
    if (n <= 0) return 0;
    TallyArgs a;
    a.logits = logits; a.labels = labels; a.n = n; a.km = km; a.kf = kf;
    for (int c = 0; c < 16; ++c) a.colmap[c] = c < kf ? colmap[c] : -1;
    a.conf = reinterpret_cast<unsigned long long *>(conf); a.win = win; a.pred = pred; a.loss_sum = loss_sum;
    int64_t grid = (n + 255) / 256;
    if (grid > (int64_t)e->num_cus * 4) grid = (int64_t)e->num_cus * 4;
    ProfScope ps(e, K_COUNT);
    hipLaunchKernelGGL(validation_tally_kernel, dim3((unsigned)grid), dim3(256), 0, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

int launch_count(rmr_engine *e, const float *logits, int64_t n, int num_out, int64_t *counts) {
    if (n <= 0) return 0;
    if (num_out > 16) RMR_FAIL(RMR_ERR_INVALID, "num_out %d > 16", num_out);
    int64_t grid = (n + 255) / 256;
    if (grid > (int64_t)e->num_cus * 4) grid = (int64_t)e->num_cus * 4;
    ProfScope ps(e, K_COUNT);
    hipLaunchKernelGGL(count_kernel, dim3((unsigned)grid), dim3(256), 0, e->stream, logits, n, num_out,
                       reinterpret_cast<unsigned long long *>(counts));
    RMR_HIP(hipGetLastError());
    return 0;
}

}  // namespace rmr
