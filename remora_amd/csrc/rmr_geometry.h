// Geometry of one chunk - RemoraRead.iter_chunks + the index arithmetic of extract_chunk
// (src/remora/data_chunks.py:443-453, :341-373) - as ONE function for the device (geometry_kernel, k_data.hip: a thread per
// chunk of a batch) and the host (rmr_call_read, engine.hip: the chunks of a single read, computed while its arrays cross
// PCIe, so that the call needs no stream synchronisation between extraction and network).  Integer arithmetic only: the two
// sides cannot differ.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rmr {

// first index with a[idx] > v  (np.searchsorted side="right")
__host__ __device__ inline int64_t ub_right(const int64_t *a, int64_t n, int64_t v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}
// first index with a[idx] >= v  (side="left")
__host__ __device__ inline int64_t lb_left(const int64_t *a, int64_t n, int64_t v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// The same two searches started from a base that is known to lie near the answer (the chunk's focus base: the window ends
// are cc_before / cc_after samples away, a few bases): gallop outwards from `hint` until the answer is bracketed, bisect
// inside.  On a non-decreasing array every correct search returns the same index, so callers use these only where the
// mapping has been checked to be monotone (rmr_call_read on the host; a handful of cache-local probes instead of
// log2(n_bases) scattered ones - the geometry of a 5 kb read's 312 chunks in 10 us instead of 60).
__host__ __device__ inline int64_t ub_right_near(const int64_t *a, int64_t n, int64_t v, int64_t hint) {
    if (n <= 0) return 0;
    if (hint > n - 1) hint = n - 1;  // (a probe always exists: nothing is assumed about a hint outside the array)
    if (hint < 0) hint = 0;
    int64_t lo, hi;  // invariant: a[lo - 1] <= v (or lo == 0), a[hi] > v (or hi == n)
    if (a[hint] > v) {
        hi = hint;
        int64_t step = 1;
        lo = hi - step;
        while (lo > 0 && a[lo] > v) { hi = lo; step <<= 1; lo = hi - step; }
        if (lo < 0) lo = 0;
    } else {
        lo = hint + 1;
        int64_t step = 1;
        hi = lo + step;
        while (hi < n && a[hi] <= v) { lo = hi + 1; step <<= 1; hi = lo + step; }
        if (hi > n) hi = n;
    }
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}
__host__ __device__ inline int64_t lb_left_near(const int64_t *a, int64_t n, int64_t v, int64_t hint) {
    if (n <= 0) return 0;
    if (hint > n - 1) hint = n - 1;
    if (hint < 0) hint = 0;
    int64_t lo, hi;  // invariant: a[lo - 1] < v (or lo == 0), a[hi] >= v (or hi == n)
    if (a[hint] >= v) {
        hi = hint;
        int64_t step = 1;
        lo = hi - step;
        while (lo > 0 && a[lo] >= v) { hi = lo; step <<= 1; lo = hi - step; }
        if (lo < 0) lo = 0;
    } else {
        lo = hint + 1;
        int64_t step = 1;
        hi = lo + step;
        while (hi < n && a[hi] < v) { lo = hi + 1; step <<= 1; hi = lo + step; }
        if (hi > n) hi = n;
    }
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// g[6] = {seq_len, chunk_sig_focus_idx, chunk_focus_base, read_focus_base, seq_start, sig_start (signed, unclipped)};
// returns seq_len.  `map`: the read's nb + 1 mapping entries.  bsj == 2: `focus` is a SIGNAL index and `offset` the read
// focus base to report (RemoraRead.extract_chunk's own arguments, include/remora_hip.h).
// `near`: the mapping is known to be non-decreasing - search from the focus base outwards (same result, fewer probes).
__host__ __device__ inline int64_t chunk_geometry_row(const int64_t *map, int64_t nb, int64_t sig_len, int64_t focus, int bsj, int offset,
                                                      int cc_before, int cc_after, int64_t *g, bool near = false) {
    int64_t fb, fsig;
    if (bsj == 2) {
        fsig = focus;
        fb = offset;
    } else {
        fb = focus + offset;
        if (fb > nb - 1) fb = nb - 1;   // map.size - 2
        if (fb < 0) fb = 0;
        fsig = bsj ? map[fb] : (map[fb] + map[fb + 1]) / 2;
    }
    const int64_t sig_start0 = fsig - cc_before;
    int64_t sig_start = sig_start0, sig_end = fsig + cc_after;
    if (sig_start < 0) sig_start = 0;
    if (sig_end > sig_len) sig_end = sig_len;
    const int64_t hint = bsj == 2 ? 0 : fb;
    const int64_t seq_start = (near && bsj != 2 ? ub_right_near(map, nb + 1, sig_start, hint) : ub_right(map, nb + 1, sig_start)) - 1;
    const int64_t seq_end = near && bsj != 2 ? lb_left_near(map, nb + 1, sig_end, hint) : lb_left(map, nb + 1, sig_end);
    const int64_t sl = seq_end - seq_start;
    g[0] = sl;
    g[1] = fsig - sig_start;
    g[2] = fb - seq_start;
    g[3] = fb;
    g[4] = seq_start;
    g[5] = sig_start0;
    return sl;
}

}  // namespace rmr
