// N2: signal-mapping refinement on the GPU (banded dynamic programming).
//
// replaces, for a batch of reads, the body of refine_signal_mapping
// (src/remora/refine_signal_map.py:780-840): compute_sig_band (:631-683) +
// convert_to_seq_band (:740-772) + adjust_seq_band (refine_signal_map_core.pyx:31-74) +
// validate_band (:686-737) + extract_levels (core.pyx:87-101) + seq_banded_dp
// (core.pyx:403-473: forward steps :150-317 and traceback :119-148).
//
// Layout of the work:
//   * refine_band_kernel (one wave per read): bands in closed form from the base breakpoints, the two
//     min-step recurrences as wave prefix/suffix scans, validation, row offsets, levels, widest column.
//   * refine_dp_kernel<W, D, ALGO> (persistent waves, 64 / W reads per wave): the forward pass walks the band
//     COLUMN by column (signal sample by sample).  Cell (base i, sample s) only depends on cells of column < s,
//     so all bases whose row contains s are evaluated together: lane (i mod W) of a W-lane group hosts base i,
//     the score of row i-1 arrives with one DPP rotate, the sample with a DPP row broadcast, the history the
//     dwell-penalty step needs (D previous-row scores, D squared residuals, D un-penalised scores) lives in
//     registers.  Every cell executes exactly the reference's float32 operations in the reference's order
//     (contraction is off in this file), so scores, traceback and paths are bit-identical.  The one term of
//     the dwell-penalty step that runs against the column order (LARGE_SCORE + last score of the previous
//     row) is speculated as "never wins" and verified when the previous row completes; on a violation the
//     now-known term is recorded in LDS and the wave replays from a checkpoint of its registers (one per 64
//     samples, 8 kept).  Traceback values leave as packed int16, 16 bytes per row per 8 samples.
//   * refine_dp_rowwise_kernel: row-by-row evaluation in global memory for what the column kernel does not
//     take (more than 64 rows per column, rows wider than 32767 samples, rows that do not start at strictly
//     increasing samples, penalty arrays longer than 6, replays out of checkpoint reach).
//   * the traceback is a pointer chase over the traceback band; a lookup outside its row (possible only
//     through "large score" cells) is mapped through the reference's flat band layout.
#pragma clang fp contract(off)
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <type_traits>

#include "rmr_internal.h"

struct rmr_refiner {
    rmr_engine *e = nullptr;
    float *d_levels = nullptr;  // [4^kmer_len]
    float *d_sdp = nullptr;     // [sd_len]
    int device = 0;  // copy: the engine may be gone when the refiner is destroyed at interpreter exit
    int kmer_len = 0, center_idx = 0, sd_len = 0, algo = 1, hbw = 5, min_step = 2;
    uint32_t *d_ckpt = nullptr;  // checkpoints of the persistent DP waves (dwell penalty only)
    int *d_counter = nullptr;    // work counter of the persistent DP waves
    int max_grid = 0;
};

namespace rmr {
namespace {

constexpr float kLargeScore = 100.0f;  // refine_signal_map_core.pyx:22
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
    uint32_t *tboff;       // [total_bases]  offset of each row in the read's traceback band (multiple of 8)
    uint32_t *flat;        // [total_bases]  offset of each row in the reference's flat band (core.pyx:445-450)
    int64_t *band_len;     // [n_reads]      traceback elements of the read (padded layout)
    int32_t *status;       // [n_reads]      0 ok, >0 rmr_refine_status, <0 needs the row-wise kernel
    int32_t *maxwin;       // [n_reads]      most rows that share one sample
    int16_t *tb;           // traceback bands of the reads in flight (one region per lane group of a wave);
                           // row i holds samples (lo[i] & ~7) .. ((hi[i] + 7) & ~7) - 1 at tboff[i], 16-byte groups
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
    if (lane == 0) { w.band_len[r] = 0; w.status[r] = 0; w.maxwin[r] = 0; }
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
    int bad = 0, maxw = 0;
    uint32_t carry = 0, fcarry = 0;
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
        // rows are stored from the 8-sample boundary below lo to the one above hi (16-byte groups of int16)
        const uint32_t wal = (p < n && wdt > 0) ? (uint32_t)(((hi[p] + 7) & ~7) - (lo[p] & ~7)) : 0u;
        maxw = max(maxw, wdt);
        uint32_t v = wal;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(v, o);
            if (lane >= o) v += t;
        }
        if (p < n) w.tboff[q0 + p] = carry + v - wal;
        const uint32_t tot = __shfl(v, 63);
        carry += tot;
        total += tot;
        uint32_t fv = (uint32_t)max(wdt, 0);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(fv, o);
            if (lane >= o) fv += t;
        }
        if (p < n) w.flat[q0 + p] = fcarry + fv - (uint32_t)max(wdt, 0);
        fcarry += __shfl(fv, 63);
    }
    bad = wave_max_i((bad & 1) ? 1 : 0) | (wave_max_i((bad & 2) ? 1 : 0) << 1) | (wave_max_i((bad & 4) ? 1 : 0) << 2);
    // widest column: rows p..j share sample hi[p]-1 when lo[j] <= hi[p]-1 (lo is non-decreasing for valid bands)
    int win = 0;
    if (!bad) {
        for (int p = lane; p < n; p += 64) {
            const int last = hi[p] - 1;
            int a0 = p, a1 = n - 1;  // largest j in [p, n) with lo[j] <= last
            while (a0 < a1) {
                const int mid = (a0 + a1 + 1) >> 1;
                if (lo[mid] <= last) a0 = mid; else a1 = mid - 1;
            }
            win = max(win, a0 - p + 1);
        }
    }
    win = wave_max_i(win);
    if (wave_max_i(maxw) > 32767) win = 1 << 20;  // stay counts would not fit the int16 traceback: row-wise kernel
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
        w.maxwin[r] = win;
    }
}

// ---------------------------------------------------------------------------------------
// column-wise forward pass + traceback
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

constexpr int kCk = 8;    // checkpoints kept per wave: a replay can reach back kCk blocks of 64 samples

// lane l receives the value of the previous lane of its W-lane group (wrapping inside the group)
template <int W>
__device__ __forceinline__ float rot1(float v) {
    constexpr int ctrl = (W == 64) ? 0x13C /* wave_ror:1 */ : 0x121 /* row_ror:1 */;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false));
}

// banded_traceback (core.pyx:119-148) for the read of each W-lane group:
// path[n] = hi[n-1]; path[b] = look - tb[flat[b] + look - lo[b]], look = path[b+1] - 1, where tb is the
// reference's flat band (rows back to back).  When a 'large score' cell (traceback -1) is on the path, look can
// fall outside row b and the reference then reads a cell of a neighbouring row; that is reproduced by mapping
// the flat index back to (row, sample).
// TB = int16_t: padded layout of the column kernel (row b at tboff[b], first element = sample lo[b] & ~7);
// TB = int32_t: the flat layout itself (row-wise kernel).
template <int W, typename TB>
__device__ void refine_traceback(bool valid, const int32_t *__restrict__ lo, const int32_t *__restrict__ hi,
                                 const uint32_t *__restrict__ tboff, const uint32_t *__restrict__ flat, const TB *tbr,
                                 int n, int nsig, int64_t st, int64_t *out) {
    constexpr bool PADDED = sizeof(TB) == 2;
    const int lane = threadIdx.x, gl = lane % W, gbase = lane - gl;
    if (valid && gl == 0) { out[0] = st; out[n] = st + nsig; }
    int pos = nsig;
    const int nmax = wave_max_i(valid ? n : 0);
    const int64_t flat_total = valid ? (int64_t)flat[n - 1] + (hi[n - 1] - lo[n - 1]) : 0;
    for (int c = ((nmax - 1) / W) * W; c >= 0; c -= W) {
        const int p = c + gl;
        const bool in = valid && p < n;
        const int l = in ? lo[p] : 0;
        const int wd = in ? hi[p] - l : 0;
        const int o = in ? (int)(PADDED ? tboff[p] : flat[p]) : 0;
        int res = 0;
        for (int j = W - 1; j >= 0; --j) {
            const int lj = __shfl(l, gbase + j), oj = __shfl(o, gbase + j), wj = __shfl(wd, gbase + j);
            const int b = c + j;
            if (valid && b >= 1 && b < n) {
                const int look = pos - 1;
                const int loc = look - lj;
                int64_t idx;
                if (loc >= 0 && loc < wj) {
                    idx = (int64_t)(uint32_t)oj + loc + (PADDED ? (lj & 7) : 0);
                } else {  // outside row b: the cell the reference's flat index lands on
                    int64_t f = (int64_t)flat[b] + loc;
                    f = f < 0 ? 0 : (f >= flat_total ? flat_total - 1 : f);
                    int r0 = 0, r1 = n - 1;  // last row with flat[row] <= f
                    while (r0 < r1) {
                        const int mid = (r0 + r1 + 1) >> 1;
                        if ((int64_t)flat[mid] <= f) r0 = mid; else r1 = mid - 1;
                    }
                    idx = PADDED ? (int64_t)tboff[r0] + (f - flat[r0]) + (lo[r0] & 7) : f;
                }
                pos = look - (int)__builtin_nontemporal_load(tbr + idx);
                if (gl == j) res = pos;
            }
        }
        if (in && p >= 1) out[p] = st + res;
    }
}

// W lanes per read (64 / W reads per wave).  Lane (i mod W) of a group hosts row (base) i.
template <int W, int D, int ALGO>
__global__ __launch_bounds__(64) void refine_dp_kernel(RefineReads a, RefineScratch w, const float *__restrict__ sdp_g,
                                                       const int32_t *__restrict__ order, int n_group, int *counter,
                                                       const int64_t *__restrict__ slot_base, uint32_t *ckpt_all,
                                                       int64_t *__restrict__ out_map) {
    constexpr int G = 64 / W;
    constexpr int RN = (W == 64) ? 256 : 64;   // staged row parameters per read (power of two)
    constexpr int SB = (W == 64) ? 64 : 16;    // samples between two staging points
    constexpr int kLT = (W == 64) ? 256 : 64;  // known "large score" terms per read, direct mapped by row: a
                                               // collision costs a replay (bounded by nroll), never correctness
    constexpr int DD = (ALGO == 1) ? D : 1;
    constexpr int NCK = 13 + 4 * DD;  // dwords per lane in a checkpoint
    constexpr int IMAX = std::numeric_limits<int>::max();
    __shared__ int4 ring[G][RN];   // {lo, hi, level bits, traceback offset} of staged rows
    __shared__ int2 ring2[G][RN];  // {hi of the previous row, lo of the next row}
    __shared__ float Ltab[G][(ALGO == 1) ? kLT : 1];
    __shared__ int Lrow[G][(ALGO == 1) ? kLT : 1];
    const int lane = threadIdx.x, grp = lane / W, gl = lane % W, gbase = grp * W;
    const float INF = std::numeric_limits<float>::infinity();
    uint32_t *ck = ckpt_all + (size_t)blockIdx.x * kCk * NCK * 64 + lane;
    float sdp[DD];
#pragma unroll
    for (int k = 0; k < DD; ++k) sdp[k] = (ALGO == 1) ? sdp_g[k] : 0.f;

    // Persistent waves: `order` lists the reads by decreasing band size; wave b starts with reads
    // b*G .. b*G+G-1 and then takes the next G from the counter, so the traceback region of a lane
    // group (sized for its first read) fits every later one.
    int16_t *const tb_slot = w.tb + slot_base[blockIdx.x * G + grp];
    for (int base = blockIdx.x * G;;) {
        if (base >= n_group) break;
        __syncthreads();  // the previous reads of this wave are done with the LDS tables

        // ---- the read of this lane group ----
        const int r = (base + grp < n_group) ? order[base + grp] : -1;
        bool gvalid = r >= 0;
        const int64_t q0 = gvalid ? a.seq_off[r] : 0;
        const int n = gvalid ? (int)(a.seq_off[r + 1] - q0) : 0;
        const int64_t *m = a.s2s + q0 + (gvalid ? r : 0);
        const int64_t st = gvalid ? m[0] : 0;
        const int nsig = gvalid ? (int)(m[n] - st) : 0;
        const int16_t *dac = a.dacs + (gvalid ? a.sig_off[r] + st : 0);
        const double sh = gvalid ? a.shift[r] : 0.0, sc = gvalid ? a.scale[r] : 1.0;
        const int32_t *lo = w.lo + q0, *hi = w.hi + q0;
        const float *lv = w.lv + q0;
        const uint32_t *tboff = w.tboff + q0;
        int16_t *tbr = tb_slot;
        if (ALGO == 1)
            for (int k = gl; k < kLT; k += W) Lrow[grp][k] = -1;

        // per-lane row state
        bool act = false, lknown = false;
        int my_i = 0, my_lo = 0, my_hi = 0, prev_hi = 0, fail = 0, ctb = 0;
        int my_toff = 0;  // traceback element offset of the row, minus (my_lo & ~7): sample s is stored at tbr[my_toff + s]
        uint32_t a[4] = {0, 0, 0, 0};  // the row's traceback values of the current 8-sample group (8 x int16)
        bool dirty = false;
        float lvl = 0.f, cur = 0.f, L = INF;
        // histories of the dwell-penalty step as rings of DD registers: slot (step mod DD) is rewritten every step, so
        // nothing is shifted; "canonical" (at sub-block boundaries) = most recent value in slot DD-1
        float Pa[DD], Qa[DD], Ua[DD];
        int Uta[DD];
#pragma unroll
        for (int k = 0; k < DD; ++k) { Pa[k] = 0.f; Qa[k] = 0.f; Ua[k] = 0.f; Uta[k] = 0; }
        bool use_ovr = false;  // row 0: the spoofed previous row [0, inf, ...] replaces the neighbour's score
        float ovr = INF;
        uint32_t specbits = 0;  // max over the speculated cells of the row, as the bit pattern of a non-negative float
        int ib_next = 0, next_lo = IMAX, staged_hi = 0;  // uniform inside a lane group
        int dnext = 0, dnext_s = -1;                     // raw samples fetched ahead, and the sample they start at
        bool gfail = false;
        bool have_known = false;  // some row of this read has a recorded 'large score' term (set by a replay)

        const int nblk = (wave_max_i(nsig) + 63) / 64;
        int blk = 0, nroll = 0;
        while (blk < nblk) {
            const int s0 = blk * 64;
            // rows that can start within the next SB samples are staged in LDS
            auto stage_rows = [&]() {
                bool any = false;
                for (;;) {
                    const bool need = gvalid && staged_hi < n && staged_hi < ib_next + SB + 1;
                    if (!__any(need)) break;
                    any = true;
                    if (need) {
                        const int i = staged_hi + gl;
                        if (i < n) {
                            ring[grp][i & (RN - 1)] = make_int4(lo[i], hi[i], __builtin_bit_cast(int, lv[i]), (int)tboff[i]);
                            ring2[grp][i & (RN - 1)] = make_int2(i > 0 ? hi[i - 1] : (IMAX >> 1), i + 1 < n ? lo[i + 1] : IMAX);
                        }
                        staged_hi += W;
                    }
                }
                if (any) __syncthreads();
            };
            stage_rows();
            if (blk == 0 && gvalid) next_lo = ring[grp][0].x;

            if (ALGO == 1) {  // checkpoint of the state at the start of block `blk`
                uint32_t *c = ck + (size_t)(blk % kCk) * NCK * 64;
                int k = 0;
#define CK_I(v) c[(k++) * 64] = (uint32_t)(v);
#define CK_F(v) c[(k++) * 64] = __builtin_bit_cast(uint32_t, v);
                CK_I((act ? 1 : 0) | (lknown ? 2 : 0) | (use_ovr ? 4 : 0))
                CK_I(my_i) CK_I(my_lo) CK_I(my_hi) CK_I(prev_hi) CK_I(ctb) CK_I(ib_next) CK_I(next_lo)
                CK_I(my_toff)
                CK_F(lvl) CK_F(cur) CK_F(L) CK_I(specbits)
#pragma unroll
                for (int d = 0; d < DD; ++d) { CK_F(Pa[d]) CK_F(Ua[d]) CK_I(Uta[d]) CK_F(Qa[d]) }
#undef CK_I
#undef CK_F
            }

            int viol = IMAX;
#pragma nounroll
            for (int kq = 0; kq < 64 / SB; ++kq) {
                if (kq > 0) stage_rows();
                const int sq = s0 + SB * kq;
                // SB samples of every read, normalised as the reference does ((dac - shift) / scale in f64); the raw
                // value of the following SB samples is fetched now and converted when its turn comes
                const int tq = sq + gl;
                if (dnext_s != sq) dnext = (tq < nsig) ? (int)dac[tq] : 0;
                const float sv = (tq < nsig) ? (float)(((double)dnext - sh) / sc) : 0.f;
                dnext = (tq + SB < nsig) ? (int)dac[tq + SB] : 0;
                dnext_s = sq + SB;
                // one cell of every active row: sample sq + jj, whose value x is broadcast inside the lane group
                // J >= 0: step number inside the sub-block as a compile-time constant (the ring slots and the traceback
                // slot are then immediates); J < 0: run-time step number, the rings are brought back to canonical after
                // every step
                auto step = [&](auto jc, const int jj, const float x) {
                    constexpr int J = decltype(jc)::value;
                    constexpr bool ST = J >= 0;
                    constexpr int r = ST ? (J % DD) : 0;  // slot written in this step
                    const int s = sq + jj;
                    const float pv = rot1<W>(cur);
                    if (s == next_lo) {  // a new row starts (rows start at strictly increasing samples)
                        const int i = ib_next;
                        const int4 pr = ring[grp][i & (RN - 1)];
                        const int2 pq = ring2[grp][i & (RN - 1)];
                        const int ph = pq.x;
                        if (gl == (i & (W - 1))) {
                            if (act && s < my_hi) fail = 1;                    // more than W rows in one column
                            if (i > 0 && (pr.x > ph || pr.y <= ph)) fail = 1;  // gap to / nested in the previous row
                            if (dirty) {  // the row this lane hosted before ended inside the current group: flush it
                                if (!ST) {
                                    for (int t = s & 7; t < 8; ++t) {
                                        a[0] = __builtin_amdgcn_alignbit(a[1], a[0], 16); a[1] = __builtin_amdgcn_alignbit(a[2], a[1], 16);
                                        a[2] = __builtin_amdgcn_alignbit(a[3], a[2], 16); a[3] >>= 16;
                                    }
                                }
                                *reinterpret_cast<uint4 *>(tbr + ((int64_t)my_toff + (s & ~7))) = make_uint4(a[0], a[1], a[2], a[3]);
                                dirty = false;
                            }
                            act = true; my_i = i;
                            my_lo = pr.x; my_hi = pr.y; lvl = __builtin_bit_cast(float, pr.z);
                            my_toff = pr.w - (pr.x & ~7);
                            prev_hi = ph;
                            lknown = (i == 0);
                            use_ovr = (i == 0);
                            ovr = 0.f;
                            L = (i == 0 && pr.y == 1) ? kLargeScore : INF;
                            specbits = 0;
                            if (ALGO == 0) { cur = INF; ctb = -1; }  // the first cell can only be a move
#pragma unroll
                            for (int k = 0; k < DD; ++k) { Pa[k] = INF; Ua[k] = INF; Uta[k] = -1; Qa[k] = 0.f; }
                            if (ALGO == 1 && have_known && i > 0 && Lrow[grp][i & (kLT - 1)] == i) {  // learnt in an earlier pass
                                lknown = true;
                                L = Ltab[grp][i & (kLT - 1)];
                            }
                        }
                        ib_next = i + 1;
                        next_lo = pq.y;
                        if (next_lo <= s) fail = 1;  // rows must start at strictly increasing samples
                    }
                    if (act && s < my_hi) {
                        // score of row i-1 at sample s-1, +inf once that row has ended: a move from it, and every
                        // penalised candidate built on it, then loses all '<' tests exactly as the reference's
                        // index checks skip them.  Row 0 sees the spoofed row [0, inf, ...] (core.pyx:360-362).
                        float pvm = (s <= prev_hi) ? pv : INF;
                        pvm = use_ovr ? ovr : pvm;
                        ovr = INF;
                        const float dlt = lvl - x;
                        const float qq = dlt * dlt;
                        float nc;
                        int nt;
                        if (ALGO == 0) {
                            // banded_forward_vit_step (core.pyx:256-317); cur = +inf, ctb = -1 at the row start
                            const float mv = pvm + qq, sy = cur + qq;
                            const bool c = mv < sy;
                            nc = c ? mv : sy;
                            nt = c ? 0 : ctb + 1;
                        } else {
                            constexpr int sU1 = ((r - 1) % DD + DD) % DD;  // un-penalised score one sample back
                            // un-penalised Viterbi row (core.pyx:183-190); +inf / -1 at the row start
                            const float mv = pvm + qq, sy = Ua[sU1] + qq;
                            const bool c = mv < sy;
                            const float un = c ? mv : sy;
                            const int ut = c ? 0 : Uta[sU1] + 1;
                            if (!lknown && s == prev_hi) {  // the previous row just completed: pv is its last score
                                L = kLargeScore + pv;
                                lknown = true;
                                if (!(specbits < __builtin_bit_cast(uint32_t, L))) {  // a cell of this row should have taken L
                                    viol = min(viol, my_lo);
                                    Ltab[grp][my_i & (kLT - 1)] = L;
                                    Lrow[grp][my_i & (kLT - 1)] = my_i;
                                }
                            }
                            // banded_forward_dwell_penalty_step (core.pyx:192-253).  P(k) = row i-1 at sample s-k
                            // (+inf before this row started or after row i-1 ended), Q(k) = residual at s-k
                            // (0 before the row started), U(k) = un-penalised score at s-k (+inf before the row
                            // started): candidates that the reference does not evaluate are +inf here
                            Pa[r] = pvm;
                            Qa[r] = qq;
                            float best = lknown ? L : INF;
                            int bt = -1;
                            float run = 0.f;
#pragma unroll
                            for (int di = 0; di < DD; ++di) {
                                const int sl = ((r - di) % DD + DD) % DD;  // P(di + 1) and Q(di) share the slot
                                run += Qa[sl];
                                const float ps = (Pa[sl] + run) + sdp[di];
                                const bool cc = ps < best;
                                best = cc ? ps : best;
                                bt = cc ? di : bt;
                            }
                            {
                                const float ps = Ua[r] + run;  // U(DD): the slot about to be rewritten
                                const bool cc = ps < best;
                                best = cc ? ps : best;
                                bt = cc ? Uta[r] + DD : bt;
                            }
                            const bool tail = (s - prev_hi >= DD);  // beyond the reach of row i-1: stay (core.pyx:201-209)
                            const uint32_t sp = (!lknown && !tail) ? __builtin_bit_cast(uint32_t, best) : 0u;
                            specbits = max(specbits, sp);
                            nc = tail ? cur + qq : best;
                            nt = tail ? ctb + 1 : bt;
                            Ua[r] = un; Uta[r] = ut;
                            if (!ST && DD > 1) {  // back to canonical: most recent value in slot DD-1
                                const float p0 = Pa[0], q0 = Qa[0], u0 = Ua[0];
                                const int t0 = Uta[0];
#pragma unroll
                                for (int k = 0; k + 1 < DD; ++k) { Pa[k] = Pa[k + 1]; Qa[k] = Qa[k + 1]; Ua[k] = Ua[k + 1]; Uta[k] = Uta[k + 1]; }
                                Pa[DD - 1] = p0; Qa[DD - 1] = q0; Ua[DD - 1] = u0; Uta[DD - 1] = t0;
                            }
                        }
                        cur = nc; ctb = nt;
                        dirty = true;
                    }
                    // traceback values leave as one 16-byte store per row every 8 samples (a 4-byte store per cell makes
                    // the kernel store-issue bound): written into their 16-bit slot when the step number is an immediate,
                    // through a 128-bit shift register otherwise
                    if (ST) {
                        constexpr int sl = (J >= 0 ? J : 0) & 7;
                        if (sl & 1) a[sl >> 1] |= (uint32_t)ctb << 16;
                        else a[sl >> 1] = (uint32_t)ctb & 0xffffu;
                    } else {
                        a[0] = __builtin_amdgcn_alignbit(a[1], a[0], 16); a[1] = __builtin_amdgcn_alignbit(a[2], a[1], 16);
                        a[2] = __builtin_amdgcn_alignbit(a[3], a[2], 16); a[3] = __builtin_amdgcn_alignbit((uint32_t)ctb, a[3], 16);
                    }
                    if ((jj & 7) == 7) {
                        if (dirty) *reinterpret_cast<uint4 *>(tbr + ((int64_t)my_toff + (s & ~7))) = make_uint4(a[0], a[1], a[2], a[3]);
                        dirty = false;
                    }
                };
                if constexpr (W == 64) {
                    for (int jj = 0; jj < 64; ++jj) step(std::integral_constant<int, -1>{}, jj, readlane_f(sv, jj));
                } else {
                    // 16 steps with the broadcast lane and the ring slots as immediates (DPP row_newbcast)
#define RMR_STEP16(J) step(std::integral_constant<int, J>{}, J, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sv), 0x150 + J, 0xf, 0xf, false)));
                    RMR_STEP16(0) RMR_STEP16(1) RMR_STEP16(2) RMR_STEP16(3) RMR_STEP16(4) RMR_STEP16(5) RMR_STEP16(6) RMR_STEP16(7)
                    RMR_STEP16(8) RMR_STEP16(9) RMR_STEP16(10) RMR_STEP16(11) RMR_STEP16(12) RMR_STEP16(13) RMR_STEP16(14) RMR_STEP16(15)
#undef RMR_STEP16
                    if (ALGO == 1 && 16 % DD != 0) {  // 16 steps later the rings are rotated by 16 mod DD: back to canonical
                        float pn[DD], qn[DD], un_[DD];
                        int tn[DD];
#pragma unroll
                        for (int m = 0; m < DD; ++m) {
                            const int src = ((15 - m) % DD + DD) % DD;
                            pn[DD - 1 - m] = Pa[src]; qn[DD - 1 - m] = Qa[src]; un_[DD - 1 - m] = Ua[src]; tn[DD - 1 - m] = Uta[src];
                        }
#pragma unroll
                        for (int k = 0; k < DD; ++k) { Pa[k] = pn[k]; Qa[k] = qn[k]; Ua[k] = un_[k]; Uta[k] = tn[k]; }
                    }
                }
            }
            // a lane group that hit an unsupported band shape stops here (its read goes to the row-wise kernel)
            {
                const unsigned long long fb = __ballot(fail != 0);
                const unsigned long long gm = ((W == 64) ? ~0ull : ((1ull << W) - 1)) << gbase;
                if (fb & gm) { gfail = true; gvalid = false; act = false; next_lo = IMAX; fail = 0; viol = IMAX; }
            }
            if (!__any(gvalid)) break;
            if (ALGO == 1) {
                const int v = wave_min_i(viol);
                if (v != IMAX) {
                    const int tblk = v / 64;  // block holding the first sample of the earliest row to redo
                    ++nroll;
                    if (blk - tblk >= kCk || nroll > 4 * nblk + 64) {
                        // out of reach of the checkpoints: the reads of this wave that asked for it go row-wise
                        if (viol != IMAX) fail = 1;
                        const unsigned long long fb = __ballot(fail != 0);
                        const unsigned long long gm = ((W == 64) ? ~0ull : ((1ull << W) - 1)) << gbase;
                        if ((fb & gm) || nroll > 4 * nblk + 64) { gfail = true; gvalid = false; act = false; next_lo = IMAX; }
                        fail = 0;
                        if (!__any(gvalid)) break;
                        ++blk;
                        continue;
                    }
                    {   // lane groups that recorded a term look it up at row starts from now on
                        const unsigned long long vb = __ballot(viol != IMAX);
                        const unsigned long long gm = ((W == 64) ? ~0ull : ((1ull << W) - 1)) << gbase;
                        if (vb & gm) have_known = true;
                    }
                    __syncthreads();  // Ltab / Lrow written above are read after the restore
                    const uint32_t *c = ck + (size_t)(tblk % kCk) * NCK * 64;
                    int k = 0;
#define CK_I(v) v = (int)c[(k++) * 64];
#define CK_F(v) v = __builtin_bit_cast(float, c[(k++) * 64]);
                    int fl;
                    CK_I(fl)
                    CK_I(my_i) CK_I(my_lo) CK_I(my_hi) CK_I(prev_hi) CK_I(ctb) CK_I(ib_next) CK_I(next_lo)
                    CK_I(my_toff)
                    CK_F(lvl) CK_F(cur) CK_F(L)
                    { int sb_; CK_I(sb_) specbits = (uint32_t)sb_; }
#pragma unroll
                    for (int d = 0; d < DD; ++d) { CK_F(Pa[d]) CK_F(Ua[d]) CK_I(Uta[d]) CK_F(Qa[d]) }
#undef CK_I
#undef CK_F
                    act = (fl & 1) != 0; lknown = (fl & 2) != 0; use_ovr = (fl & 4) != 0; ovr = INF;
                    if (!gvalid) { act = false; next_lo = IMAX; }
                    staged_hi = max(ib_next - 1, 0) / W * W;  // re-stage the row parameters from there
                    dnext_s = -1;
                    blk = tblk;
                    continue;
                }
            }
            ++blk;
        }
        if (!gfail && r >= 0 && ib_next != n) gfail = true;
        if (gfail && r >= 0 && gl == 0) w.status[r] = -1;  // the row-wise kernel takes this read
        __threadfence();
        __syncthreads();
        refine_traceback<W, int16_t>(r >= 0 && !gfail, lo, hi, tboff, w.flat + q0, tbr, n, nsig, st,
                                     out_map + q0 + (r >= 0 ? r : 0));
        if (lane == 0) base = atomicAdd(counter, G);
        base = __builtin_amdgcn_readfirstlane(base);
    }
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
                                                               float *sigbuf, int32_t *tbbuf,
                                                               int64_t *__restrict__ out_map) {
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
    const uint32_t *tboff = w.flat + q0;  // rows back to back, as the reference stores them
    int32_t *tbr = tbbuf + sc_base[blockIdx.x];
    // scratch of this read: scores[band_len] | signal[nsig] | unpen[maxbw] | unpen_tb[maxbw] | spoof[hi0]
    float *sco = scores + sc_base[blockIdx.x];
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
    refine_traceback<64, int32_t>(true, lo, hi, tboff, tboff, tbr, n, nsig, st, out_map + q0 + r);
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

template <int W, int ALGO>
int launch_dp(rmr_refiner *rf, const RefineReads &dr, const RefineScratch &w, const int32_t *d_order, int n_group,
              int grid, const int64_t *d_slot_base, int64_t *out) {
    rmr_engine *e = rf->e;
    constexpr int G = 64 / W;
    const int first = grid * G;  // reads handed out statically
    RMR_HIP(hipMemcpyAsync(rf->d_counter, &first, sizeof(int), hipMemcpyHostToDevice, e->stream));
    ProfScope ps(e, K_REFINE_DP);
#define RMR_DP_CASE(DV)                                                                                           \
    case DV:                                                                                                      \
        hipLaunchKernelGGL((refine_dp_kernel<W, DV, ALGO>), dim3(grid), dim3(64), 0, e->stream, dr, w, rf->d_sdp,  \
                           d_order, n_group, rf->d_counter, d_slot_base, rf->d_ckpt, out);                                      \
        break;
    if constexpr (ALGO == 0) {
        switch (1) { RMR_DP_CASE(1) }
    } else {
        switch (rf->sd_len) {
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


// ---------------------------------------------------------------------------------------
// rough re-scale: the 19 quantiles of the normalised centre samples and of the expected levels of one read
// (RemoraRead-level caller: src/remora/refine_signal_map.py:330-420, rough_rescale; np.quantile "linear")
// ---------------------------------------------------------------------------------------
// One block per read.  The kept bases (all of a read <= 2*clip bases long, else the middle n - 2*clip) are evaluated
// into LDS, sorted there (bitonic network, +inf padding to a power of two) and the quantiles interpolated with
// numpy's own two-sided formula (numpy/lib/_function_base_impl.py _lerp: a + (b-a)*t, and b - (b-a)*(1-t) where
// t >= 0.5; the difference b - a in the array's dtype).  Every product and sum is rounded on its own (fp contract
// off: no FMAs) so the values equal numpy's bit for bit.
template <typename T>
__device__ __forceinline__ void lds_bitonic_sort(T *buf, int n_pad, int tid, int nt) {
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n_pad >> 1); t += nt) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // insert a 0 bit at position log2(j)
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const T a = buf[lo], b = buf[hi];
                if ((a > b) == up) { buf[lo] = b; buf[hi] = a; }
            }
            __syncthreads();
        }
    }
}

template <typename T>
__device__ __forceinline__ void lerp_quantiles(const T *buf, int cnt, const double *quants, int nq, double *out, int tid) {
    if (tid < nq) {
#pragma clang fp contract(off)  // hipcc contracts a * b + c into an FMA by default (and __dmul_rn / __dadd_rn are plain operators)
        const double top = (double)(cnt - 1);
        const double vi = top * quants[tid];
        const double prev = floor(vi);
        const double gamma = vi - prev;
        int pi = (int)prev, ni = pi + 1;
        if (vi >= top) pi = ni = cnt - 1;
        const T lo = buf[pi], hi = buf[ni];
        const T diff = hi - lo;  // in the array's own dtype, as numpy's subtract(b, a)
        const double scaled_lo = (double)diff * gamma, scaled_hi = (double)diff * (1.0 - gamma);
        const double res = gamma >= 0.5 ? (double)hi - scaled_hi : (double)lo + scaled_lo;
        out[tid] = res;
    }
}

__global__ __launch_bounds__(256) void rescale_quantiles_kernel(RefineReads a, const float *__restrict__ levels, int kmer_len,
                                                                int center, int clip, int nq,
                                                                const double *__restrict__ quants, int max_pad,
                                                                double *__restrict__ sig_q, double *__restrict__ lvl_q,
                                                                int32_t *__restrict__ status) {
    extern __shared__ double rq_lds[];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int64_t q0 = a.seq_off[r];
    const int n = (int)(a.seq_off[r + 1] - q0);
    const bool clipped = clip > 0 && n > 2 * clip;
    const int off = clipped ? clip : 0, cnt = clipped ? n - 2 * clip : n;
    int n_pad = 2;
    while (n_pad < cnt) n_pad <<= 1;
    if (cnt < 1 || n_pad > max_pad) {  // empty read / longer than the LDS sort holds: the caller's general path
        if (tid == 0) status[r] = cnt < 1 ? 2 : 1;
        return;
    }
    if (tid == 0) status[r] = 0;
    const int64_t *m = a.s2s + q0 + r;
    const int16_t *dacs = a.dacs + a.sig_off[r];
    const double shift = a.shift[r], scale = a.scale[r];
    // normalised centre sample of every kept base, float64 ((dacs - shift) / scale as RemoraRead.sig evaluates it
    // in float64 here: rough_rescale works on the float64 normalisation, refine_signal_map.py:330-352)
    for (int j = tid; j < n_pad; j += 256) {
        double v = __builtin_inf();
        if (j < cnt) {
            const int64_t mid = (m[off + j] + m[off + j + 1]) / 2;
            v = ((double)dacs[mid] - shift) / scale;
        }
        rq_lds[j] = v;
    }
    __syncthreads();
    lds_bitonic_sort(rq_lds, n_pad, tid, 256);
    lerp_quantiles(rq_lds, cnt, quants, nq, sig_q + (size_t)r * nq, tid);
    __syncthreads();
    // expected level of every kept base (core.pyx:87-101), float32
    float *lv = reinterpret_cast<float *>(rq_lds);
    const int8_t *seq = a.int_seq + q0;
    const int64_t kmax = ((int64_t)1 << (2 * kmer_len)) - 1;
    for (int j = tid; j < n_pad; j += 256) {
        float v = __builtin_inff();
        if (j < cnt) {
            const int pos = off + j - center;
            v = 0.f;
            if (pos >= 0 && pos + kmer_len <= n) {
                int64_t idx = 0;
                for (int k = 0; k < kmer_len; ++k) idx = idx * 4 + seq[pos + k];
                v = levels[idx < 0 ? 0 : (idx > kmax ? kmax : idx)];
            }
        }
        lv[j] = v;
    }
    __syncthreads();
    lds_bitonic_sort(lv, n_pad, tid, 256);
    lerp_quantiles(lv, cnt, quants, nq, lvl_q + (size_t)r * nq, tid);
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
    r->device = e->device;
    r->kmer_len = desc->kmer_len; r->center_idx = desc->center_idx; r->algo = desc->algo;
    r->hbw = desc->half_bandwidth; r->min_step = desc->min_step;
    r->sd_len = desc->algo == RMR_REFINE_DWELL_PENALTY ? desc->sd_len : 0;
    hipError_t he = hipMalloc(&r->d_levels, nk * 4);
    if (he == hipSuccess && r->sd_len) he = hipMalloc(&r->d_sdp, (size_t)r->sd_len * 4);
    if (he == hipSuccess) he = hipMemcpy(r->d_levels, desc->kmer_levels, nk * 4, hipMemcpyHostToDevice);
    if (he == hipSuccess && r->sd_len) he = hipMemcpy(r->d_sdp, desc->sd_arr, (size_t)r->sd_len * 4, hipMemcpyHostToDevice);
    r->max_grid = e->num_cus * 16;
    if (he == hipSuccess) he = hipMalloc(&r->d_counter, 256);
    if (he == hipSuccess && r->sd_len && r->sd_len <= kMaxD)
        he = hipMalloc(&r->d_ckpt, (size_t)r->max_grid * kCk * (13 + 4 * r->sd_len) * 64 * sizeof(uint32_t));
    if (he != hipSuccess) {
        if (r->d_levels) (void)hipFree(r->d_levels);
        if (r->d_sdp) (void)hipFree(r->d_sdp);
        if (r->d_counter) (void)hipFree(r->d_counter);
        if (r->d_ckpt) (void)hipFree(r->d_ckpt);
        delete r;
        RMR_FAIL(RMR_ERR_HIP, "HIP error %s creating refiner", hipGetErrorString(he));
    }
    *out = r;
    return 0;
}

void rmr_refiner_destroy(rmr_refiner *r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (r->d_levels) (void)hipFree(r->d_levels);
    if (r->d_sdp) (void)hipFree(r->d_sdp);
    if (r->d_counter) (void)hipFree(r->d_counter);
    if (r->d_ckpt) (void)hipFree(r->d_ckpt);
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

    size_t bytes = 5 * pad256((size_t)tb * 4 + 4) + 3 * pad256(n1 * 8) + 2 * pad256(n1 * 4) + pad256((size_t)(tb + n_reads) * 8) +
                   pad256(((size_t)rf->max_grid * 4 + 4) * 8) + 8192;
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
    w.flat = st.take<uint32_t>(tb + 1);
    w.band_len = st.take<int64_t>(n1);
    w.status = st.take<int32_t>(n1);
    w.maxwin = st.take<int32_t>(n1);
    int64_t *d_out = (mem == RMR_MEM_HOST) ? st.take<int64_t>(tb + n_reads) : out_map;
    int32_t *d_todo = reinterpret_cast<int32_t *>(st.take<int64_t>(n1));  // reused: row-wise work list
    int64_t *d_scb = st.take<int64_t>(std::max<size_t>(n1, (size_t)rf->max_grid * 4 + 4));

    {
        ProfScope ps(e, K_REFINE_BAND);
        hipLaunchKernelGGL(refine_band_kernel, dim3((unsigned)n_reads), dim3(64), 0, e->stream, dr, w, rf->d_levels,
                           rf->kmer_len, rf->center_idx, rf->hbw, rf->min_step);
        RMR_HIP(hipGetLastError());
    }
    std::vector<int64_t> band_len(n_reads);
    std::vector<int32_t> hstat(n_reads), maxwin(n_reads);
    RMR_D2H(band_len.data(), w.band_len, (size_t)n_reads * 8);
    RMR_D2H(hstat.data(), w.status, (size_t)n_reads * 4);
    RMR_D2H(maxwin.data(), w.maxwin, (size_t)n_reads * 4);
    RMR_HIP(hipStreamSynchronize(e->stream));

    const bool force_rowwise = tune_int("RMR_REFINE_ROWWISE", 0) != 0 || rf->sd_len > kMaxD;
    const int64_t cap_cells = (int64_t)32768 * (1 << 19);
    std::vector<int32_t> l16, l64, todo;
    if (!force_rowwise) {
        // 16 lanes per read (4 reads per wave) when no sample is shared by more than 16 rows, else 64;
        // largest bands first: the persistent waves finish together and a wave's traceback region,
        // sized for its first read, fits all its later ones
        const int force_w = tune_int("RMR_REFINE_W", 0);
        for (int64_t r = 0; r < n_reads; ++r) {
            if (hstat[r] != 0) continue;
            if (band_len[r] >= ((int64_t)1 << 31)) continue;  // 32-bit row offsets of the column kernel: row-wise instead
            if (maxwin[r] <= 16 && force_w != 64) l16.push_back((int32_t)r);
            else if (maxwin[r] <= 64) l64.push_back((int32_t)r);
        }
        auto by_len = [&](int32_t x, int32_t y) { return band_len[x] > band_len[y]; };
        std::stable_sort(l16.begin(), l16.end(), by_len);
        std::stable_sort(l64.begin(), l64.end(), by_len);
        std::vector<int64_t> slot_base;
        for (int pass = 0; pass < 2; ++pass) {
            const std::vector<int32_t> &lst = pass == 0 ? l16 : l64;
            if (lst.empty()) continue;
            const int G = pass == 0 ? 4 : 1;
            int grid = std::min((int)((lst.size() + G - 1) / G), rf->max_grid);
            int64_t cells = 0;
            for (;;) {  // one traceback region per lane group in flight, within the arena budget
                slot_base.assign((size_t)grid * G, 0);
                cells = 0;
                for (size_t k = 0; k < slot_base.size(); ++k) {
                    slot_base[k] = cells;
                    if (k < lst.size()) cells += (band_len[lst[k]] + 63) & ~(int64_t)63;
                }
                if (cells <= cap_cells || grid == 1) break;
                grid = std::max(1, grid / 2);
            }
            RMR_TRY(e->ensure(e->act, (size_t)cells * 2 + 256));
            w.tb = reinterpret_cast<int16_t *>(e->act.ptr);
            RMR_H2D(d_todo, lst.data(), lst.size() * 4);
            RMR_H2D(d_scb, slot_base.data(), slot_base.size() * 8);
            if (pass == 0) {
                if (rf->algo == RMR_REFINE_VITERBI) RMR_TRY((launch_dp<16, 0>(rf, dr, w, d_todo, (int)lst.size(), grid, d_scb, d_out)));
                else RMR_TRY((launch_dp<16, 1>(rf, dr, w, d_todo, (int)lst.size(), grid, d_scb, d_out)));
            } else {
                if (rf->algo == RMR_REFINE_VITERBI) RMR_TRY((launch_dp<64, 0>(rf, dr, w, d_todo, (int)lst.size(), grid, d_scb, d_out)));
                else RMR_TRY((launch_dp<64, 1>(rf, dr, w, d_todo, (int)lst.size(), grid, d_scb, d_out)));
            }
            RMR_HIP(hipStreamSynchronize(e->stream));  // host lists and the arena are reused by the next pass
        }
        RMR_D2H(hstat.data(), w.status, (size_t)n_reads * 4);
        RMR_HIP(hipStreamSynchronize(e->stream));
        for (int64_t r = 0; r < n_reads; ++r)
            if ((maxwin[r] > 64 || band_len[r] >= ((int64_t)1 << 31)) && hstat[r] == 0) hstat[r] = -1;  // not for the column kernel
    }
    // reads handed back by the column kernel (or all of them when forced) are evaluated row by row
    {
        std::vector<int64_t> scb;
        int64_t sc_cells = 0;
        for (int64_t r = 0; r < n_reads; ++r)
            if (hstat[r] < 0 || (force_rowwise && hstat[r] == 0)) {
                todo.push_back((int32_t)r);
                scb.push_back(sc_cells);
                // per read: scores[band_len] | signal + unpen + unpen_tb + spoof (<= 4 * samples + 8) | tb[band_len],
                // three buffers with identical offsets
                sc_cells += ((band_len[r] + (so[r + 1] - so[r]) * 4 + 64) + 63) & ~(int64_t)63;
            }
        if (!todo.empty()) {
            void *extra = nullptr;
            RMR_HIP(hipMalloc(&extra, (size_t)sc_cells * 12 + 256));
            float *scores = reinterpret_cast<float *>(extra);
            float *tails = scores + sc_cells;
            int32_t *tbbuf = reinterpret_cast<int32_t *>(tails + sc_cells);
            hipError_t he = hipMemcpyAsync(d_todo, todo.data(), todo.size() * 4, hipMemcpyHostToDevice, e->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(d_scb, scb.data(), scb.size() * 8, hipMemcpyHostToDevice, e->stream);
            if (he == hipSuccess) {
                ProfScope ps(e, K_REFINE_ROWWISE);
                hipLaunchKernelGGL(refine_dp_rowwise_kernel, dim3((unsigned)todo.size()), dim3(64), 0, e->stream, dr, w,
                                   rf->d_sdp, rf->sd_len, rf->algo, d_todo, d_scb, scores, tails, tbbuf, d_out);
                he = hipGetLastError();
            }
            if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
            (void)hipFree(extra);
            if (he != hipSuccess) RMR_FAIL(RMR_ERR_HIP, "HIP error %s in row-wise refinement", hipGetErrorString(he));
            for (int32_t r : todo) hstat[r] = 0;
        }
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


int rmr_rescale_quantiles(rmr_refiner *rf, int64_t n_reads, const int16_t *dacs, const int64_t *sig_off,
                          const int64_t *seq_to_sig, const int8_t *int_seq, const int64_t *seq_off, const double *shift,
                          const double *scale, int64_t max_read_bases, int clip_bases, int n_quants, const double *quants,
                          double *sig_q, double *lvl_q, int32_t *status) {
    if (!rf || !dacs || !sig_off || !seq_to_sig || !int_seq || !seq_off || !shift || !scale || !quants || !sig_q || !lvl_q ||
        !status)
        RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (n_reads < 0 || n_reads > (int64_t)1 << 30) RMR_FAIL(RMR_ERR_INVALID, "bad n_reads");
    if (n_quants < 1 || n_quants > 256 || clip_bases < 0) RMR_FAIL(RMR_ERR_INVALID, "bad n_quants / clip_bases");
    for (int q = 0; q < n_quants; ++q)
        if (!(quants[q] >= 0.0 && quants[q] <= 1.0)) RMR_FAIL(RMR_ERR_INVALID, "quantile %d outside [0, 1]", q);
    if (n_reads == 0) return 0;
    rmr_engine *e = rf->e;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    RMR_TRY(e->ensure(e->staging, 4096));
    double *d_q = reinterpret_cast<double *>(e->staging.ptr);
    RMR_HIP(hipMemcpyAsync(d_q, quants, (size_t)n_quants * 8, hipMemcpyHostToDevice, e->stream));
    // LDS holds the kept bases of one read as float64, padded to a power of two: sized for the longest read the caller
    // names (0: unknown), at most 16384 (128 KB); longer reads come back with status 1
    const int cap = std::min(16384, std::max(64, 16384));
    int max_pad = 64;
    while (max_pad < cap && (max_read_bases <= 0 || max_pad < max_read_bases)) max_pad <<= 1;
    RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(rmr::rescale_quantiles_kernel)));
    rmr::RefineReads dr{dacs, sig_off, seq_to_sig, seq_off, int_seq, shift, scale};
    {
        ProfScope ps(e, K_RESCALE_Q);
        hipLaunchKernelGGL(rmr::rescale_quantiles_kernel, dim3((unsigned)n_reads), dim3(256), (size_t)max_pad * 8, e->stream,
                           dr, rf->d_levels, rf->kmer_len, rf->center_idx, clip_bases, n_quants, d_q, max_pad, sig_q, lvl_q,
                           status);
        RMR_HIP(hipGetLastError());
    }
    RMR_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

}  // extern "C"
