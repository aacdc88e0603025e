// AddressSanitizer harness for rmr_ref_to_signal (remora_amd/csrc/ref_to_signal.cpp): random and hostile CIGARs against move
// tables and output buffers of exact size - every read and write the walk makes stays inside them, a buffer that is too
// small is refused before anything is written, and the result is monotone wherever the move table is.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../remora_amd/csrc/ref_to_signal.cpp"

namespace rmr {
void set_error(const char *, ...) {}
}  // namespace rmr

int main() {
    std::mt19937_64 rng(7);
    long ok = 0, refused = 0, small = 0;
    for (int trial = 0; trial < 200000; ++trial) {
        const int n_ops = 1 + (int)(rng() % 12);
        std::vector<uint32_t> cig((size_t)n_ops);
        int64_t q_len = 0, r_len = 0;
        const bool hostile = trial % 9 == 0;
        for (auto &c : cig) {
            const uint32_t op = hostile ? (uint32_t)(rng() % 16) : (uint32_t)(rng() % 9);
            const uint32_t len = hostile && rng() % 4 == 0 ? (uint32_t)(rng() % (1u << 28)) : (uint32_t)(rng() % 40);
            c = (len << 4) | op;
            if (op <= 8) {
                if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) q_len += len;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) r_len += len;
            }
        }
        // the move table: the query length the CIGAR implies, or something else
        int64_t n_knots = q_len + 1;
        if (trial % 5 == 0) n_knots = 1 + (int64_t)(rng() % 60);
        if (n_knots > 5000) n_knots = 5000;
        std::vector<int64_t> q2s((size_t)n_knots);
        int64_t s = 0;
        for (auto &v : q2s) v = (s += (int64_t)(rng() % 9));
        int64_t cap = r_len + 1;
        if (trial % 7 == 0) cap = (int64_t)(rng() % 20);
        if (cap > 100000) cap = 100000;
        int64_t *out = (int64_t *)malloc(sizeof(int64_t) * (size_t)(cap > 0 ? cap : 1));  // exact size: ASan sees one entry too many
        int64_t n_out = -1;
        const int rc = rmr_ref_to_signal(cig.data(), n_ops, (int)(rng() & 1), q2s.data(), n_knots, out, cap, &n_out);
        if (rc == 0) {
            if (n_out > cap) { printf("wrote beyond the capacity\n"); return 1; }
            for (int64_t i = 1; i < n_out; ++i)
                if (out[i] < out[i - 1]) { printf("not monotone\n"); return 1; }
            ++ok;
        } else if (n_out > cap) {
            ++small;
        } else {
            ++refused;
        }
        free(out);
    }
    printf("%ld mapped, %ld refused (operations / no match run), %ld turned away for room\n", ok, refused, small);
    return 0;
}
