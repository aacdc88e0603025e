// ref_to_signal.cpp — the signal coordinate of every reference position of an alignment, host code (N3 / N4: reference-anchored
// reads of `infer --reference-anchored` and `dataset prepare`).
// replaces: compute_ref_to_signal = map_ref_to_signal(make_sequence_coordinate_mapping(cigar)), src/remora/data_chunks.py:60-122,
// which are two np.interp passes over float64 arrays of the read's length plus a dozen small numpy calls per read - 0.15 ms
// of the 0.33 ms of host time a 9 kb reference-anchored read costs.  Here: one walk over the CIGAR.
//
// The result is the reference's, bit for bit, because the arithmetic is np.interp's:
//   knots (x = reference, y = query): (0, 0), per match run (M = X) its first and its last base, (ref_len, query_len)
//   - operations behind the last match run are dropped first; knots may repeat an x (a leading soft clip, a match run of
//   one base): np.interp takes the LAST knot with x_j <= x, so only segments of positive width are ever evaluated;
//   y(x) = y_j when x == x_j, else slope * (x - x_j) + y_j with slope = (y_j+1 - y_j) / (x_j+1 - x_j) in float64
//   (exact wherever the slope is 1, i.e. inside match runs; a fraction only across deletions / skips);
//   signal(x) = floor(interp(y(x); 0..n-1 -> query_to_signal)): q2s[k] when y == k, q2s[n-1] when y >= n-1, else
//   (q2s[k+1] - q2s[k]) * (y - k) + q2s[k].
#include <cmath>
#include <cstdint>
#include <vector>

#include "rmr_internal.h"

extern "C" int rmr_ref_to_signal(const uint32_t *cigar, int64_t n_ops, int reverse, const int64_t *query_to_signal, int64_t n_knots,
                                 int64_t *ref_to_signal, int64_t cap, int64_t *n_out) {
#pragma STDC FP_CONTRACT OFF
    if (!cigar || n_ops < 0 || !query_to_signal || n_knots < 1 || !ref_to_signal || !n_out) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    // M I D N S H P = X: aligns a base / consumes the query / consumes the reference
    static const bool MATCH[9] = {true, false, false, false, false, false, false, true, true};
    static const bool QUERY[9] = {true, true, false, false, true, false, false, true, true};
    static const bool REF[9] = {true, false, true, true, false, false, false, true, true};
    auto op_at = [&](int64_t i) { return cigar[reverse ? n_ops - 1 - i : i]; };
    int64_t last_match = -1;
    for (int64_t i = 0; i < n_ops; ++i) {
        const uint32_t op = op_at(i) & 0xF;
        if (op > 8) RMR_FAIL(RMR_ERR_INVALID, "Invalid cigar op(s)");
        if (MATCH[op]) last_match = i;
    }
    if (last_match < 0) RMR_FAIL(RMR_ERR_INVALID, "No match operations found in alignment cigar");
    // the knots
    std::vector<int64_t> xs, ys;
    xs.reserve(2 * (size_t)n_ops + 2);
    ys.reserve(2 * (size_t)n_ops + 2);
    xs.push_back(0);
    ys.push_back(0);
    int64_t r = 0, q = 0;
    for (int64_t i = 0; i <= last_match; ++i) {
        const uint32_t c = op_at(i), op = c & 0xF;
        const int64_t len = (int64_t)(c >> 4);
        if (REF[op]) r += len;
        if (QUERY[op]) q += len;
        if (MATCH[op]) {  // (ref_end - len, ref_end - 1): also for a run of length 0, as the reference's arithmetic does
            xs.push_back(r - len);
            ys.push_back(q - len);
            xs.push_back(r - 1);
            ys.push_back(q - 1);
        }
    }
    xs.push_back(r);
    ys.push_back(q);
    const int64_t total = r + 1;
    *n_out = total;
    if (cap < total) RMR_FAIL(RMR_ERR_INVALID, "ref_to_signal needs %lld entries, %lld given", (long long)total, (long long)cap);
    for (size_t j = 1; j < xs.size(); ++j)  // np.interp wants ascending x; a run of length 0 behind nothing would break that
        if (xs[j] < xs[j - 1]) RMR_FAIL(RMR_ERR_INVALID, "cigar with an empty match run");
    const int64_t last_knot = n_knots - 1;
    auto signal_at = [&](double y) -> int64_t {
        if (y >= (double)last_knot) return query_to_signal[last_knot];
        const double fk = std::floor(y);
        const int64_t k = (int64_t)fk;
        if (y == fk) return query_to_signal[k];
        const double y0 = (double)query_to_signal[k], slope = ((double)query_to_signal[k + 1] - y0) / (((double)k + 1.0) - (double)k);
        const double prod = slope * (y - fk);
        return (int64_t)std::floor(prod + y0);
    };
    const size_t nk = xs.size();
    for (size_t j = 0; j + 1 < nk; ++j) {
        const int64_t x0 = xs[j], x1 = xs[j + 1];
        if (x1 <= x0) continue;
        const int64_t y0 = ys[j], y1 = ys[j + 1];
        if (y1 - y0 == x1 - x0 && y0 >= 0 && y1 <= last_knot) {  // slope exactly 1 inside the move table: a copy
            for (int64_t x = x0; x < x1; ++x) ref_to_signal[x] = query_to_signal[y0 + (x - x0)];
            continue;
        }
        const double slope = ((double)y1 - (double)y0) / ((double)x1 - (double)x0);
        ref_to_signal[x0] = signal_at((double)y0);
        for (int64_t x = x0 + 1; x < x1; ++x) {
            const double prod = slope * ((double)x - (double)x0);
            ref_to_signal[x] = signal_at(prod + (double)y0);
        }
    }
    ref_to_signal[r] = signal_at((double)ys[nk - 1]);  // x == the last knot: its y
    return RMR_OK;
}
