// N2: signal-mapping refinement on the GPU (banded dynamic programming).
//
// replaces, for a batch of reads, the body of refine_signal_mapping
// (src/remora/refine_signal_map.py:780-840): compute_sig_band (:631-683) +
// convert_to_seq_band (:740-772) + adjust_seq_band (refine_signal_map_core.pyx:31-74) +
// validate_band (:686-737) + extract_levels (core.pyx:87-101) + seq_banded_dp
// (core.pyx:403-473: forward steps :150-317 and traceback :119-148).
//
// Layout of the work: one 64-lane wave per read.
//   * refine_band_kernel: bands in closed form from the base breakpoints, the two
//     min-step recurrences as wave prefix/suffix scans, validation, row offsets, levels.
//   * refine_dp_kernel: the forward pass walks the band COLUMN by column (signal sample by
//     sample).  Cell (base i, sample s) only depends on cells of column < s, so all bases
//     whose row contains s are evaluated together: lane (i mod 64) hosts base i, the value
//     of row i-1 arrives with one DPP wave rotate, the history the dwell-penalty step needs
//     (D previous-row scores, D squared residuals, D un-penalised scores) lives in
//     registers.  Every cell executes exactly the reference's float32 operations in the
//     reference's order (contraction is off in this file), so scores, traceback and paths
//     are bit-identical.  The one non-local term of the dwell-penalty step
//     (LARGE_SCORE + last score of the previous row) is speculated as "never wins" and
//     verified when the previous row completes; reads that violate it, bands wider than 64
//     rows per column, or bands that are not strictly increasing are re-run by
//     refine_dp_rowwise_kernel, a row-by-row evaluation in global memory.
//   * the traceback is a pointer chase over the int32 traceback band (L2 resident).
#pragma clang fp contract(off)
#include <cmath>
#include <cstring>
#include <limits>

#include "rmr_internal.h"

struct rmr_refiner {
    rmr_engine *e = nullptr;
    float *d_levels = nullptr;  // [4^kmer_len]
    float *d_sdp = nullptr;     // [sd_len]
    int kmer_len = 0, center_idx = 0, sd_len = 0, algo = 1, hbw = 5, min_step = 2;
};

namespace rmr {
namespace {

constexpr float kLargeScore = 100.0f;  // refine_signal_map_core.pyx:22
constexpr int kRing = 256;             // staged row parameters per wave (power of two)
constexpr int kMaxD = 6;               // longest short-dwell penalty array on the register path

struct RefineReads {
    const int16_t *dacs;
    const int64_t *sig_off, *s2s, *seq_off;
    const int8_t *int_seq;
    const double *shift, *scale;
};

struct RefineScratch {
    int32_t *lo, *hi;      // [total_bases]  seq band, sample coordinates relative to s2s[0]
    float *lv;             // [total_bases]  expected level of each base
    uint32_t *tboff;       // [total_bases]  offset of each row in the read's traceback band
    int64_t *band_len;     // [n_reads]
    int64_t *tb_base;      // [n_reads]      offset of the read's band in `tb`
    int32_t *status;       // [n_reads]      0 ok, >0 rmr_refine_status, <0 needs the row-wise kernel
    int32_t *tb;           // traceback bands
};

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

// ---------------------------------------------------------------------------------------
// bands + levels
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void refine_band_kernel(RefineReads a, RefineScratch w, const float *__restrict__ levels,
                                                         int kmer_len, int center, int hbw, int ms) {
    const int r = blockIdx.x, lane = threadIdx.x;
    const int64_t q0 = a.seq_off[r];
    int n = (int)(a.seq_off[r + 1] - q0);
    const int64_t *m = a.s2s + q0 + r;
    int32_t *lo = w.lo + q0, *hi = w.hi + q0;
    if (lane == 0) { w.band_len[r] = 0; w.status[r] = 0; }
    if (n <= 0) { if (lane == 0) w.status[r] = RMR_REFINE_EMPTY; return; }
    const int64_t st = m[0];
    const int nsig = (int)(m[n] - st);

    // levels (core.pyx:87-101): bases without a full k-mer keep level 0
    const int8_t *seq = a.int_seq + q0;
    const int kmask = (1 << (2 * kmer_len)) - 1;
    for (int i = lane; i < n; i += 64) {
        const int pos = i - center;
        float v = 0.f;
        if (pos >= 0 && pos < n - kmer_len + 1) {
            int idx = 0;
            for (int k = 0; k < kmer_len; ++k) idx = idx * 4 + seq[pos + k];
            v = levels[idx & kmask];
        }
        w.lv[q0 + i] = v;
    }
    if (nsig <= 0) { if (lane == 0) w.status[r] = RMR_REFINE_ZERO_LEN; return; }

    // first / last base with a non-empty dwell (anchor the ends of the signal band)
    int j0 = n, jl = -1;
    for (int c = 0; c < n && j0 == n; c += 64) {
        const int i = c + lane;
        const unsigned long long b = __ballot(i < n && m[i + 1] > st);
        if (b) j0 = c + __ffsll((long long)b) - 1;
    }
    for (int c = ((n - 1) / 64) * 64; c >= 0 && jl < 0; c -= 64) {
        const int i = c + lane;
        const unsigned long long b = __ballot(i < n && m[i] - st < nsig);
        if (b) jl = c + 63 - __clzll((long long)b);
    }
    // convert_to_seq_band sizes the band by sig_band[1,-1] = min(jl+hbw+1, n_bases); a shorter band
    // is adjusted and validated like any other and then fails validate_band's length check
    // (refine_signal_map.py:733-735), so everything below works on the first n rows
    const int n_bases = n;
    n = min(jl + hbw + 1, n_bases);

    // raw seq band: base i is inside the half-width window of every sample of bases i-hbw..i+hbw
    const int jlead = max(j0 - hbw, 0);
    for (int i = lane; i < n; i += 64) {
        lo[i] = (int)(m[max(i - hbw, 0)] - st);
        hi[i] = (int)(m[min(max(i, jlead) + hbw + 1, n_bases)] - st);
    }
    __syncthreads();
    const int band_min = lo[0], band_max = hi[n - 1];

    // adjust_seq_band (core.pyx:31-74).  lower bounds: lo[p] = min(lo[p], lo[p+1]-ms), p = n-2..0
    {
        int carry = std::numeric_limits<int>::max();
        for (int c = ((n - 1) / 64) * 64; c >= 0; c -= 64) {
            const int p = c + lane;
            int v = (p < n) ? lo[p] - ms * p : std::numeric_limits<int>::max();
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_down(v, o);
                if (lane + o < 64) v = min(v, t);
            }
            v = min(v, carry);
            if (p < n) lo[p] = v + ms * p;
            carry = __shfl(v, 0);
        }
        __syncthreads();
        if (lane == 0) lo[0] = band_min;
        __syncthreads();
        // while lo[p] <= lo[p-1]: lo[p] = lo[p-1] + 1  ==  lo[p] = band_min + p up to the first p
        // whose adjusted bound already exceeds band_min + p - 1
        for (int c = 0; c < n; c += 64) {
            const int p = c + lane;
            const bool stop = (p >= 1 && p < n) ? (lo[p] > band_min + p - 1) : (p >= n);
            const unsigned long long b = __ballot(stop);
            const int first = b ? __ffsll((long long)b) - 1 : 64;
            if (p >= 1 && p < n && lane < first) lo[p] = band_min + p;
            if (b) break;
        }
    }
    // upper bounds: hi[p] = max(hi[p], hi[p-1]+ms), p = 1..n-1
    {
        int carry = std::numeric_limits<int>::min();
        for (int c = 0; c < n; c += 64) {
            const int p = c + lane;
            int v = (p < n) ? hi[p] - ms * p : std::numeric_limits<int>::min();
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(v, o);
                if (lane >= o) v = max(v, t);
            }
            v = max(v, carry);
            if (p < n) hi[p] = v + ms * p;
            carry = __shfl(v, 63);
        }
        __syncthreads();
        if (lane == 0) hi[n - 1] = band_max;
        __syncthreads();
        // while hi[p] >= hi[p+1]: hi[p] = hi[p+1] - 1, p = n-2 downwards
        for (int c = ((n - 1) / 64) * 64; c >= 0; c -= 64) {
            const int p = c + lane;
            const bool in = (p >= 0 && p <= n - 2);
            const bool stop = in ? (hi[p] < band_max - (n - 2 - p)) : (p < 0);
            const unsigned long long b = __ballot(stop);
            const int last = b ? 63 - __clzll((long long)b) : -1;
            if (in && lane > last) hi[p] = band_max - (n - 1 - p);
            if (b) break;
        }
    }
    __syncthreads();

    // validate_band (refine_signal_map.py:686-737) + row offsets (core.pyx:445-450)
    int bad = 0;
    uint32_t carry = 0;
    int64_t total = 0;
    for (int c = 0; c < n; c += 64) {
        const int p = c + lane;
        int wdt = 0;
        if (p < n) {
            const int l = lo[p], h = hi[p];
            wdt = h - l;
            if (wdt <= 0) bad |= 1;
            if (p + 1 < n) {
                if (lo[p + 1] < l) bad |= 2;
                if (hi[p + 1] < h) bad |= 4;
            }
        }
        uint32_t v = (uint32_t)max(wdt, 0);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(v, o);
            if (lane >= o) v += t;
        }
        if (p < n) w.tboff[q0 + p] = carry + v - (uint32_t)max(wdt, 0);
        const uint32_t tot = __shfl(v, 63);
        carry += tot;
        total += tot;
    }
    bad = wave_max_i((bad & 1) ? 1 : 0) | (wave_max_i((bad & 2) ? 1 : 0) << 1) | (wave_max_i((bad & 4) ? 1 : 0) << 2);
    if (lane == 0) {
        int s = 0;
        if (lo[0] != 0) s = RMR_REFINE_BAND_START;
        else if (bad & 1) s = RMR_REFINE_ZERO_LEN;
        else if (bad & 2) s = RMR_REFINE_START_ORDER;
        else if (bad & 4) s = RMR_REFINE_END_ORDER;
        else if (hi[n - 1] != nsig) s = RMR_REFINE_BAND_END;
        else if (n != n_bases) s = RMR_REFINE_BAND_LENGTH;
        w.status[r] = s;
        w.band_len[r] = total;
    }
}

// ---------------------------------------------------------------------------------------
// column-wise forward pass + traceback
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_ror1(float v) {
    // lane l receives lane (l-1) & 63
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, false));
}
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

__device__ void refine_traceback(const int32_t *__restrict__ lo, const uint32_t *__restrict__ tboff, const int32_t *tbr,
                                 int64_t band_len, int n, int nsig, int64_t st, int64_t *out) {
    // banded_traceback (core.pyx:119-148): path[n] = hi[n-1]; path[b] = look - tb[row b][look - lo[b]]
    const int lane = threadIdx.x;
    if (lane == 0) { out[0] = st; out[n] = st + nsig; }
    int pos = nsig;
    for (int c = ((n - 1) / 64) * 64; c >= 0; c -= 64) {
        const int p = c + lane;
        const int l = (p < n) ? lo[p] : 0;
        const uint32_t o = (p < n) ? tboff[p] : 0;
        int res = 0;
        for (int j = min(63, n - 1 - c); j >= 0; --j) {
            if (c + j == 0) break;
            const int look = pos - 1;
            int64_t idx = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)o, j) + look - __builtin_amdgcn_readlane(l, j);
            idx = idx < 0 ? 0 : (idx >= band_len ? band_len - 1 : idx);
            const int t = __builtin_amdgcn_readfirstlane(__builtin_nontemporal_load(tbr + idx));
            pos = look - t;
            if (lane == j) res = pos;
        }
        if (p >= 1 && p < n) out[p] = st + res;
    }
}

template <int D, int ALGO>
__global__ __launch_bounds__(64) void refine_dp_kernel(RefineReads a, RefineScratch w, const float *__restrict__ sdp_g,
                                                       int r0, int64_t *__restrict__ out_map) {
    __shared__ int4 ring[kRing];
    const int r = r0 + blockIdx.x, lane = threadIdx.x;
    if (w.status[r] != 0) return;
    const int64_t q0 = a.seq_off[r];
    const int n = (int)(a.seq_off[r + 1] - q0);
    const int64_t *m = a.s2s + q0 + r;
    const int64_t st = m[0];
    const int nsig = (int)(m[n] - st);
    const int16_t *dac = a.dacs + a.sig_off[r] + st;
    const double sh = a.shift[r], sc = a.scale[r];
    const int32_t *lo = w.lo + q0, *hi = w.hi + q0;
    const float *lv = w.lv + q0;
    const uint32_t *tboff = w.tboff + q0;
    int32_t *tbr = w.tb + w.tb_base[r];
    const float INF = std::numeric_limits<float>::infinity();
    constexpr int DD = (ALGO == 1) ? D : 1;
    float sdp[DD];
#pragma unroll
    for (int k = 0; k < DD; ++k) sdp[k] = (ALGO == 1) ? sdp_g[k] : 0.f;

    // per-lane row state
    bool act = false, is0 = false, lknown = false;
    int my_lo = 0, my_hi = 0, prev_hi = 0, fail = 0;
    uint32_t my_off = 0;
    float lvl = 0.f, cur = 0.f, L = INF, specmax = -INF;
    int ctb = 0;
    float P[DD + 1], Q[DD], U[DD + 1];
    int Ut[DD + 1];
#pragma unroll
    for (int k = 0; k <= DD; ++k) { P[k] = 0.f; U[k] = 0.f; Ut[k] = 0; }
#pragma unroll
    for (int k = 0; k < DD; ++k) Q[k] = 0.f;

    int ib_next = 0, next_lo = 0, staged_hi = 0;  // wave-uniform
    for (int s0 = 0; s0 < nsig; s0 += 64) {
        // rows that can start inside this block of 64 samples are staged in LDS
        while (staged_hi < n && staged_hi < ib_next + 65) {
            const int i = staged_hi + lane;
            if (i < n) ring[i & (kRing - 1)] = make_int4(lo[i], hi[i], __builtin_bit_cast(int, lv[i]), (int)tboff[i]);
            staged_hi += 64;
        }
        __syncthreads();
        if (s0 == 0) next_lo = ring[0].x;
        const int sidx = s0 + lane;
        const float sv = (sidx < nsig) ? (float)(((double)dac[sidx] - sh) / sc) : 0.f;
        const int send = min(64, nsig - s0);
        for (int j = 0; j < send; ++j) {
            const int s = s0 + j;
            const float x = readlane_f(sv, j);
            float pv = wave_ror1(cur);
            if (s == next_lo) {  // a new row starts (rows start at strictly increasing samples)
                const int i = ib_next;
                const int4 pr = ring[i & (kRing - 1)];
                const int ph = (i > 0) ? ring[(i - 1) & (kRing - 1)].y : (std::numeric_limits<int>::max() >> 1);
                if (lane == (i & 63)) {
                    if (act && s < my_hi) fail = 1;            // more than 64 rows in one column
                    if (i > 0 && (pr.x > ph || pr.y <= ph)) fail = 1;  // gap to / nested in the previous row
                    act = true; is0 = (i == 0);
                    my_lo = pr.x; my_hi = pr.y; lvl = __builtin_bit_cast(float, pr.z); my_off = (uint32_t)pr.w;
                    prev_hi = ph;
                    lknown = is0;
                    L = is0 ? ((pr.y == 1) ? kLargeScore : INF) : INF;
                    specmax = -INF;
                }
                ib_next = i + 1;
                next_lo = (ib_next < n) ? __builtin_amdgcn_readfirstlane(ring[ib_next & (kRing - 1)].x)
                                        : std::numeric_limits<int>::max();
                if (next_lo <= s) fail = 1;  // rows must start at strictly increasing samples
            }
            if (act && s < my_hi) {
                if (is0) pv = (s == 0) ? 0.f : INF;  // spoofed previous row [0, inf, ...] (core.pyx:360-362)
                const float dlt = lvl - x;
                const float qq = dlt * dlt;
                const int b = s - my_lo;
                float nc;
                int nt;
                if (ALGO == 0) {
                    // banded_forward_vit_step (core.pyx:256-317)
                    if (b == 0) { nc = pv + qq; nt = 0; }
                    else if (s <= prev_hi) {
                        const float mv = pv + qq, sy = cur + qq;
                        if (mv < sy) { nc = mv; nt = 0; } else { nc = sy; nt = ctb + 1; }
                    } else { nc = cur + qq; nt = ctb + 1; }
                } else {
                    // un-penalised Viterbi row (core.pyx:183-190)
                    float un;
                    int ut;
                    if (b == 0) { un = pv + qq; ut = 0; }
                    else if (s <= prev_hi) {
                        const float mv = pv + qq, sy = U[1] + qq;
                        if (mv < sy) { un = mv; ut = 0; } else { un = sy; ut = Ut[1] + 1; }
                    } else { un = U[1] + qq; ut = Ut[1] + 1; }
#pragma unroll
                    for (int k = DD; k >= 2; --k) P[k] = P[k - 1];
                    P[1] = pv;
#pragma unroll
                    for (int k = DD - 1; k >= 1; --k) Q[k] = Q[k - 1];
                    Q[0] = qq;
                    if (!lknown && s == prev_hi) {  // the previous row just completed: pv is its last score
                        L = kLargeScore + pv;
                        lknown = true;
                        if (!(specmax < L)) fail = 1;
                    }
                    // banded_forward_dwell_penalty_step (core.pyx:192-253)
                    if (s - prev_hi >= DD) { nc = cur + qq; nt = ctb + 1; }
                    else {
                        float best = lknown ? L : INF;
                        int bt = -1;
                        float run = 0.f;
#pragma unroll
                        for (int di = 0; di < DD; ++di) {
                            if (di <= b) {
                                run += Q[di];
                                if (s - di - 1 < prev_hi) {
                                    const float ps = (P[di + 1] + run) + sdp[di];
                                    if (ps < best) { best = ps; bt = di; }
                                }
                            }
                        }
                        if (b >= DD) {
                            const float ps = U[DD] + run;
                            if (ps < best) { best = ps; bt = Ut[DD] + DD; }
                        }
                        if (!lknown) specmax = fmaxf(specmax, best);
                        nc = best; nt = bt;
                    }
#pragma unroll
                    for (int k = DD; k >= 2; --k) { U[k] = U[k - 1]; Ut[k] = Ut[k - 1]; }
                    U[1] = un; Ut[1] = ut;
                }
                cur = nc; ctb = nt;
                tbr[(int64_t)my_off + b] = nt;
            }
        }
        if (__any(fail)) break;
    }
    if (__any(fail) || ib_next != n) {
        if (lane == 0) w.status[r] = -1;  // row-wise kernel takes this read
        return;
    }
    __threadfence();
    __syncthreads();
    refine_traceback(lo, tboff, tbr, w.band_len[r], n, nsig, st, out_map + q0 + r);
}

// ---------------------------------------------------------------------------------------
// row-wise evaluation for reads the column kernel hands back (status -1): one lane per read,
// rows in global memory, any band shape and any penalty length
// ---------------------------------------------------------------------------------------
__device__ float sqd(float l, float x) { const float t = l - x; return t * t; }

__device__ void row_vit(float *curr, int32_t *tb, const float *prev, int prev_n, float level, const float *sig, int bw,
                        int bsd) {
    if (bsd == 0) { curr[0] = kLargeScore + prev[prev_n - 1]; tb[0] = -1; }
    else { curr[0] = prev[bsd - 1] + sqd(level, sig[0]); tb[0] = 0; prev += bsd; prev_n -= bsd; }
    if (prev_n == bw) prev_n -= 1;
    for (int b = 1; b < prev_n + 1; ++b) {
        const float q = sqd(level, sig[b]);
        const float mv = prev[b - 1] + q, sy = curr[b - 1] + q;
        if (mv < sy) { curr[b] = mv; tb[b] = 0; } else { curr[b] = sy; tb[b] = tb[b - 1] + 1; }
    }
    for (int b = max(prev_n + 1, 1); b < bw; ++b) { curr[b] = curr[b - 1] + sqd(level, sig[b]); tb[b] = tb[b - 1] + 1; }
}

__device__ void row_dwell(float *curr, int32_t *tb, const float *prev, int prev_n, float level, const float *sig, int bw,
                          int bsd, const float *sdp, int d, float *unpen, int32_t *unpen_tb) {
    row_vit(unpen, unpen_tb, prev, prev_n, level, sig, bw, bsd);
    for (int b = 0; b < bw; ++b) {
        if (b + bsd - prev_n >= d) {
            if (b == 0) { curr[0] = kLargeScore + prev[prev_n - 1]; tb[0] = -1; continue; }  // not reachable for valid bands
            curr[b] = curr[b - 1] + sqd(level, sig[b]);
            tb[b] = tb[b - 1] + 1;
            continue;
        }
        float best = kLargeScore + prev[prev_n - 1];
        int bt = -1;
        if (!(b == 0 && bsd == 0)) {
            float run = 0.f;
            for (int di = 0; di < d; ++di) {
                if (di > b || (bsd == 0 && b == di)) break;
                run += sqd(level, sig[b - di]);
                if (b - di - 1 + bsd >= prev_n) continue;
                const float ps = (prev[b - di - 1 + bsd] + run) + sdp[di];
                if (ps < best) { best = ps; bt = di; }
            }
            if (b >= d) {
                const float ps = unpen[b - d] + run;
                if (ps < best) { best = ps; bt = unpen_tb[b - d] + d; }
            }
        }
        curr[b] = best; tb[b] = bt;
    }
}

__global__ __launch_bounds__(64) void refine_dp_rowwise_kernel(RefineReads a, RefineScratch w, const float *__restrict__ sdp,
                                                               int d, int algo, const int32_t *__restrict__ todo,
                                                               const int64_t *__restrict__ sc_base, float *scores,
                                                               float *sigbuf, int64_t *__restrict__ out_map) {
    const int r = todo[blockIdx.x], lane = threadIdx.x;
    const int64_t q0 = a.seq_off[r];
    const int n = (int)(a.seq_off[r + 1] - q0);
    const int64_t *m = a.s2s + q0 + r;
    const int64_t st = m[0];
    const int nsig = (int)(m[n] - st);
    const int16_t *dac = a.dacs + a.sig_off[r] + st;
    const double sh = a.shift[r], sc = a.scale[r];
    const int32_t *lo = w.lo + q0, *hi = w.hi + q0;
    const float *lv = w.lv + q0;
    const uint32_t *tboff = w.tboff + q0;
    int32_t *tbr = w.tb + w.tb_base[r];
    // scratch of this read: scores[band_len] | signal[nsig] | unpen[maxbw] | unpen_tb[maxbw] | spoof[hi0]
    float *sco = scores + sc_base[blockIdx.x];
    const int64_t bl = w.band_len[r];
    float *sig = sigbuf + sc_base[blockIdx.x];
    for (int i = lane; i < nsig; i += 64) sig[i] = (float)(((double)dac[i] - sh) / sc);
    int maxbw = 0;
    for (int i = lane; i < n; i += 64) maxbw = max(maxbw, hi[i] - lo[i]);
    maxbw = wave_max_i(maxbw);
    __syncthreads();
    if (lane == 0) {
        float *unpen = sig + nsig;
        int32_t *unpen_tb = reinterpret_cast<int32_t *>(unpen + maxbw + 1);
        float *spoof = reinterpret_cast<float *>(unpen_tb + maxbw + 1);
        int bw = hi[0];
        for (int i = 0; i < bw; ++i) spoof[i] = std::numeric_limits<float>::infinity();
        spoof[0] = 0.f;
        if (algo == 0) row_vit(sco, tbr, spoof, bw, lv[0], sig, bw, 1);
        else row_dwell(sco, tbr, spoof, bw, lv[0], sig, bw, 1, sdp, d, unpen, unpen_tb);
        int prev_bw = bw, prev_st = 0;
        uint32_t prev_off = 0;
        for (int b = 1; b < n; ++b) {
            const int s = lo[b];
            bw = hi[b] - s;
            const uint32_t off = tboff[b];
            if (algo == 0) row_vit(sco + off, tbr + off, sco + prev_off, prev_bw, lv[b], sig + s, bw, s - prev_st);
            else row_dwell(sco + off, tbr + off, sco + prev_off, prev_bw, lv[b], sig + s, bw, s - prev_st, sdp, d, unpen, unpen_tb);
            prev_st = s; prev_bw = bw; prev_off = off;
        }
        w.status[r] = 0;
    }
    __threadfence();
    __syncthreads();
    refine_traceback(lo, tboff, tbr, bl, n, nsig, st, out_map + q0 + r);
}

// ---- host helpers -------------------------------------------------------------------------
struct Bump {
    char *base = nullptr;
    size_t off = 0;
    template <typename T>
    T *take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T *p = reinterpret_cast<T *>(base + off);
        off += count * sizeof(T);
        return p;
    }
};
inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

template <int ALGO>
int launch_dp(rmr_engine *e, int d, const RefineReads &dr, const RefineScratch &w, const float *sdp, int r0, int nr,
              int64_t *out) {
    ProfScope ps(e, K_REFINE_DP);
#define RMR_DP_CASE(DV)                                                                                   \
    case DV:                                                                                              \
        hipLaunchKernelGGL((refine_dp_kernel<DV, ALGO>), dim3(nr), dim3(64), 0, e->stream, dr, w, sdp, r0, out); \
        break;
    if constexpr (ALGO == 0) {
        switch (1) { RMR_DP_CASE(1) }
    } else {
        switch (d) {
            RMR_DP_CASE(1)
            RMR_DP_CASE(2)
            RMR_DP_CASE(3)
            RMR_DP_CASE(4)
            RMR_DP_CASE(5)
            RMR_DP_CASE(6)
            default: RMR_FAIL(RMR_ERR_INVALID, "internal: penalty length");
        }
    }
#undef RMR_DP_CASE
    RMR_HIP(hipGetLastError());
    return 0;
}

}  // namespace
}  // namespace rmr

using namespace rmr;

extern "C" {

int rmr_refiner_create(rmr_engine *e, const rmr_refine_desc *desc, rmr_refiner **out) {
    if (!e || !desc || !out) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (!desc->kmer_levels || desc->kmer_len < 1 || desc->kmer_len > 12)
        RMR_FAIL(RMR_ERR_INVALID, "k-mer level table missing or k-mer length %d outside 1..12", desc->kmer_len);
    if (desc->center_idx < 0 || desc->center_idx >= desc->kmer_len)
        RMR_FAIL(RMR_ERR_INVALID, "center_idx %d outside the %d-mer", desc->center_idx, desc->kmer_len);
    if (desc->algo != RMR_REFINE_VITERBI && desc->algo != RMR_REFINE_DWELL_PENALTY)
        RMR_FAIL(RMR_ERR_INVALID, "unknown refinement algorithm %d", desc->algo);
    if (desc->algo == RMR_REFINE_DWELL_PENALTY && (!desc->sd_arr || desc->sd_len < 1))
        RMR_FAIL(RMR_ERR_INVALID, "dwell_penalty needs a short dwell penalty array");
    if (desc->half_bandwidth < 0 || desc->min_step < 1) RMR_FAIL(RMR_ERR_INVALID, "bad half_bandwidth / min_step");
    const size_t nk = (size_t)1 << (2 * desc->kmer_len);
    for (size_t i = 0; i < nk; ++i)
        if (std::isnan(desc->kmer_levels[i]))
            RMR_FAIL(RMR_ERR_INVALID, "k-mer level table contains NaN (NaN-anchored bands are not supported on the GPU path)");
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    auto *r = new rmr_refiner;
    r->e = e;
    r->kmer_len = desc->kmer_len; r->center_idx = desc->center_idx; r->algo = desc->algo;
    r->hbw = desc->half_bandwidth; r->min_step = desc->min_step;
    r->sd_len = desc->algo == RMR_REFINE_DWELL_PENALTY ? desc->sd_len : 0;
    hipError_t he = hipMalloc(&r->d_levels, nk * 4);
    if (he == hipSuccess && r->sd_len) he = hipMalloc(&r->d_sdp, (size_t)r->sd_len * 4);
    if (he == hipSuccess) he = hipMemcpy(r->d_levels, desc->kmer_levels, nk * 4, hipMemcpyHostToDevice);
    if (he == hipSuccess && r->sd_len) he = hipMemcpy(r->d_sdp, desc->sd_arr, (size_t)r->sd_len * 4, hipMemcpyHostToDevice);
    if (he != hipSuccess) {
        if (r->d_levels) (void)hipFree(r->d_levels);
        if (r->d_sdp) (void)hipFree(r->d_sdp);
        delete r;
        RMR_FAIL(RMR_ERR_HIP, "HIP error %s creating refiner", hipGetErrorString(he));
    }
    *out = r;
    return 0;
}

void rmr_refiner_destroy(rmr_refiner *r) {
    if (!r) return;
    (void)hipSetDevice(r->e->device);
    if (r->d_levels) (void)hipFree(r->d_levels);
    if (r->d_sdp) (void)hipFree(r->d_sdp);
    delete r;
}

const char *rmr_refine_status_message(int status) {
    switch (status) {  // texts of validate_band, src/remora/refine_signal_map.py:703-737
        case 0: return "ok";
        case RMR_REFINE_BAND_START: return "Band does not start with 0 coordinate.";
        case RMR_REFINE_ZERO_LEN: return "Band contains 0-length region";
        case RMR_REFINE_START_ORDER: return "Band start positions are not monotonically increasing";
        case RMR_REFINE_END_ORDER: return "Band end positions are not monotonically increasing";
        case RMR_REFINE_BAND_END: return "Invalid seq_band end coordinate";
        case RMR_REFINE_BAND_LENGTH: return "Invalid sig_band length";
        case RMR_REFINE_EMPTY: return "Read without bases";
        default: return "unknown refinement status";
    }
}

int rmr_refine_signal_maps(rmr_refiner *rf, int64_t n_reads, const int16_t *dacs, const int64_t *sig_off,
                           const int64_t *seq_to_sig, const int8_t *int_seq, const int64_t *seq_off,
                           const double *shift, const double *scale, int64_t *out_map, int32_t *status, int mem) {
    if (!rf || !dacs || !sig_off || !seq_to_sig || !int_seq || !seq_off || !shift || !scale || !out_map || !status)
        RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (n_reads < 0 || n_reads > (int64_t)1 << 30) RMR_FAIL(RMR_ERR_INVALID, "bad n_reads");
    if (n_reads == 0) return 0;
    rmr_engine *e = rf->e;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    const size_t n1 = (size_t)n_reads + 1;
    std::vector<int64_t> so(n1), qo(n1);
    if (mem == RMR_MEM_HOST) {
        memcpy(so.data(), sig_off, n1 * 8);
        memcpy(qo.data(), seq_off, n1 * 8);
    } else {
        RMR_HIP(hipMemcpy(so.data(), sig_off, n1 * 8, hipMemcpyDeviceToHost));
        RMR_HIP(hipMemcpy(qo.data(), seq_off, n1 * 8, hipMemcpyDeviceToHost));
    }
    const int64_t ts = so[n_reads], tb = qo[n_reads];
    for (int64_t r = 0; r < n_reads; ++r)
        if (so[r + 1] < so[r] || qo[r + 1] < qo[r] || qo[r + 1] - qo[r] > (int64_t)1 << 30)
            RMR_FAIL(RMR_ERR_INVALID, "offsets of read %lld are not increasing", (long long)r);

    size_t bytes = 4 * pad256((size_t)tb * 4 + 4) + 3 * pad256(n1 * 8) + pad256(n1 * 4) + pad256((size_t)(tb + n_reads) * 8) + 8192;
    if (mem == RMR_MEM_HOST)
        bytes += pad256((size_t)ts * 2) + 2 * pad256(n1 * 8) + pad256((size_t)(tb + n_reads) * 8) + pad256((size_t)tb) + 2 * pad256(n1 * 8);
    RMR_TRY(e->ensure(e->staging, bytes));
    Bump st;
    st.base = reinterpret_cast<char *>(e->staging.ptr);
    RefineReads dr{dacs, sig_off, seq_to_sig, seq_off, int_seq, shift, scale};
#define RMR_H2D(dst, src, b) RMR_HIP(hipMemcpyAsync((dst), (src), (b), hipMemcpyHostToDevice, e->stream))
#define RMR_D2H(dst, src, b) RMR_HIP(hipMemcpyAsync((dst), (src), (b), hipMemcpyDeviceToHost, e->stream))
    if (mem == RMR_MEM_HOST) {
        auto *d_dacs = st.take<int16_t>(ts + 1);
        auto *d_so = st.take<int64_t>(n1);
        auto *d_map = st.take<int64_t>(tb + n_reads);
        auto *d_seq = st.take<int8_t>(tb + 1);
        auto *d_qo = st.take<int64_t>(n1);
        auto *d_sh = st.take<double>(n1);
        auto *d_sc = st.take<double>(n1);
        if (ts) RMR_H2D(d_dacs, dacs, (size_t)ts * 2);
        RMR_H2D(d_so, sig_off, n1 * 8);
        RMR_H2D(d_map, seq_to_sig, (size_t)(tb + n_reads) * 8);
        if (tb) RMR_H2D(d_seq, int_seq, (size_t)tb);
        RMR_H2D(d_qo, seq_off, n1 * 8);
        RMR_H2D(d_sh, shift, (size_t)n_reads * 8);
        RMR_H2D(d_sc, scale, (size_t)n_reads * 8);
        dr = RefineReads{d_dacs, d_so, d_map, d_qo, d_seq, d_sh, d_sc};
    }
    RefineScratch w{};
    w.lo = st.take<int32_t>(tb + 1);
    w.hi = st.take<int32_t>(tb + 1);
    w.lv = st.take<float>(tb + 1);
    w.tboff = st.take<uint32_t>(tb + 1);
    w.band_len = st.take<int64_t>(n1);
    w.tb_base = st.take<int64_t>(n1);
    w.status = st.take<int32_t>(n1);
    int64_t *d_out = (mem == RMR_MEM_HOST) ? st.take<int64_t>(tb + n_reads) : out_map;
    int32_t *d_todo = reinterpret_cast<int32_t *>(st.take<int64_t>(n1));  // reused: row-wise work list
    int64_t *d_scb = st.take<int64_t>(n1);

    {
        ProfScope ps(e, K_REFINE_BAND);
        hipLaunchKernelGGL(refine_band_kernel, dim3((unsigned)n_reads), dim3(64), 0, e->stream, dr, w, rf->d_levels,
                           rf->kmer_len, rf->center_idx, rf->hbw, rf->min_step);
        RMR_HIP(hipGetLastError());
    }
    std::vector<int64_t> band_len(n_reads), tb_base(n_reads);
    std::vector<int32_t> hstat(n_reads);
    RMR_D2H(band_len.data(), w.band_len, (size_t)n_reads * 8);
    RMR_D2H(hstat.data(), w.status, (size_t)n_reads * 4);
    RMR_HIP(hipStreamSynchronize(e->stream));

    // groups of consecutive reads whose traceback bands fit the arena budget
    const int64_t cap_cells = (int64_t)tune_int("RMR_REFINE_TB_MIB", 4096) * (1 << 18);
    const bool force_rowwise = tune_int("RMR_REFINE_ROWWISE", 0) != 0 || rf->sd_len > kMaxD;
    int64_t r0 = 0;
    std::vector<int32_t> todo;
    while (r0 < n_reads) {
        int64_t cells = 0, r1 = r0;
        while (r1 < n_reads) {
            const int64_t need = hstat[r1] == 0 ? ((band_len[r1] + 63) & ~(int64_t)63) : 0;
            if (r1 > r0 && cells + need > cap_cells) break;
            tb_base[r1] = cells;
            cells += need;
            ++r1;
        }
        RMR_TRY(e->ensure(e->act, (size_t)cells * 4 + 256));
        w.tb = reinterpret_cast<int32_t *>(e->act.ptr);
        RMR_H2D(w.tb_base + r0, tb_base.data() + r0, (size_t)(r1 - r0) * 8);
        if (!force_rowwise) {
            if (rf->algo == RMR_REFINE_VITERBI) RMR_TRY(launch_dp<0>(e, 1, dr, w, rf->d_sdp, (int)r0, (int)(r1 - r0), d_out));
            else RMR_TRY(launch_dp<1>(e, rf->sd_len, dr, w, rf->d_sdp, (int)r0, (int)(r1 - r0), d_out));
            RMR_D2H(hstat.data() + r0, w.status + r0, (size_t)(r1 - r0) * 4);
            RMR_HIP(hipStreamSynchronize(e->stream));
        }
        // reads handed back by the column kernel (or all of them when forced)
        todo.clear();
        std::vector<int64_t> scb;
        int64_t sc_cells = 0;
        for (int64_t r = r0; r < r1; ++r)
            if (hstat[r] < 0 || (force_rowwise && hstat[r] == 0)) {
                todo.push_back((int32_t)r);
                scb.push_back(sc_cells);
                // scores[band_len] are followed by signal + unpen + unpen_tb + spoof of the same read
                sc_cells += ((band_len[r] + (so[r + 1] - so[r]) * 4 + 64) + 63) & ~(int64_t)63;
            }
        if (!todo.empty()) {
            // two arenas of sc_cells floats each would double count: scores and the per-read tail share one buffer,
            // laid out as [scores region of all reads][tail region of all reads] with identical offsets
            void *extra = nullptr;
            RMR_HIP(hipMalloc(&extra, (size_t)sc_cells * 8 + 256));
            float *scores = reinterpret_cast<float *>(extra);
            float *tails = scores + sc_cells;
            hipError_t he = hipMemcpyAsync(d_todo, todo.data(), todo.size() * 4, hipMemcpyHostToDevice, e->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(d_scb, scb.data(), scb.size() * 8, hipMemcpyHostToDevice, e->stream);
            if (he == hipSuccess) {
                ProfScope ps(e, K_REFINE_ROWWISE);
                hipLaunchKernelGGL(refine_dp_rowwise_kernel, dim3((unsigned)todo.size()), dim3(64), 0, e->stream, dr, w,
                                   rf->d_sdp, rf->sd_len, rf->algo, d_todo, d_scb, scores, tails, d_out);
                he = hipGetLastError();
            }
            if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
            (void)hipFree(extra);
            if (he != hipSuccess) RMR_FAIL(RMR_ERR_HIP, "HIP error %s in row-wise refinement", hipGetErrorString(he));
            for (int32_t r : todo) hstat[r] = 0;
        }
        r0 = r1;
    }
    if (mem == RMR_MEM_HOST) {
        RMR_D2H(out_map, d_out, (size_t)(tb + n_reads) * 8);
        memcpy(status, hstat.data(), (size_t)n_reads * 4);
    } else {
        RMR_H2D(status, hstat.data(), (size_t)n_reads * 4);
    }
    RMR_HIP(hipStreamSynchronize(e->stream));
#undef RMR_H2D
#undef RMR_D2H
    return 0;
}

}  // extern "C"
