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

// g[6] = {seq_len, chunk_sig_focus_idx, chunk_focus_base, read_focus_base, seq_start, sig_start (signed, unclipped)};
// returns seq_len.  `map`: the read's nb + 1 mapping entries.  bsj == 2: `focus` is a SIGNAL index and `offset` the read
// focus base to report (RemoraRead.extract_chunk's own arguments, include/remora_hip.h).
__host__ __device__ inline int64_t chunk_geometry_row(const int64_t *map, int64_t nb, int64_t sig_len, int64_t focus, int bsj, int offset,
                                                      int cc_before, int cc_after, int64_t *g) {
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
    const int64_t seq_start = ub_right(map, nb + 1, sig_start) - 1;
    const int64_t seq_end = lb_left(map, nb + 1, sig_end);
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
