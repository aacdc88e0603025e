// AddressSanitizer harness for the batch forms of round 5 (host code, no GPU): rmr_ref_anchor_batch (move table x CIGAR of a
// whole BAM batch on native threads: hostile move tables, strides, signal lengths and CIGARs against output slots of exact
// size) and rmr_orient_bases (strand-aware bases + codes of selected records: ragged, empty and adjacent records, every byte
// value, exact-size outputs).
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../remora_amd/csrc/ref_to_signal.cpp"
#include "../../remora_amd/csrc/pack_reads.cpp"

namespace rmr {
void set_error(const char *, ...) {}
}  // namespace rmr

int main() {
    std::mt19937_64 rng(11);
    long anchored = 0, turned_away = 0;
    for (int trial = 0; trial < 4000; ++trial) {
        const int64_t n = 1 + (int64_t)(rng() % 40);
        std::vector<int8_t> mv;
        std::vector<int64_t> mv_off{0}, sig_len, seq_len, cigar_off{0}, ref_len, r2s_off{0};
        std::vector<uint32_t> cigar;
        std::vector<uint8_t> rev;
        for (int64_t i = 0; i < n; ++i) {
            const bool hostile = rng() % 6 == 0;
            const int stride = hostile ? (int)(rng() % 9) - 2 : 5;
            const int64_t nb = (int64_t)(rng() % 120);  // bases the move table encodes
            int64_t q = 0, r = 0;
            const int n_ops = 1 + (int)(rng() % 8);
            for (int k = 0; k < n_ops; ++k) {
                const uint32_t op = hostile ? (uint32_t)(rng() % 16) : (uint32_t)(rng() % 9);
                uint32_t len = (uint32_t)(rng() % 30);
                if (hostile && rng() % 5 == 0) len = (uint32_t)(rng() % (1u << 27));
                cigar.push_back((len << 4) | op);
                if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) q += len;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) r += len;
            }
            cigar_off.push_back((int64_t)cigar.size());
            if (rng() % 11 != 0) {  // a move table (else: none)
                mv.push_back((int8_t)stride);
                int64_t placed = 0;
                const int64_t moves = nb * 2 + (int64_t)(rng() % 5);
                for (int64_t k = 0; k < moves; ++k) {
                    const bool one = placed < nb && (k % 2 == 0);
                    mv.push_back(hostile ? (int8_t)(rng() % 3 - 1) : (int8_t)one);
                    placed += one;
                }
                sig_len.push_back(hostile && rng() % 2 ? (int64_t)(rng() % 2000) : moves * (stride > 0 ? stride : 1));
            } else {
                sig_len.push_back((int64_t)(rng() % 100));
            }
            mv_off.push_back((int64_t)mv.size());
            seq_len.push_back(rng() % 7 == 0 ? (int64_t)(rng() % 200) : (rng() % 3 == 0 ? q : nb));
            int64_t rl = rng() % 9 == 0 ? -1 : (rng() % 4 == 0 ? (int64_t)(rng() % 300) : (r > 100000 ? 100000 : r));
            ref_len.push_back(rl);
            r2s_off.push_back(r2s_off.back() + (rl < 0 ? 0 : rl + 1));
            rev.push_back((uint8_t)(rng() & 1));
        }
        // exact-size output: one malloc per call, ASan sees a write one entry past a record's slot only at the very end, so
        // poison the slots' neighbours by checking that untouched sentinels stay untouched
        std::vector<int64_t> r2s((size_t)r2s_off.back() + 1, INT64_MIN);
        int64_t *out = (int64_t *)malloc(sizeof(int64_t) * (size_t)(r2s_off.back() > 0 ? r2s_off.back() : 1));
        std::vector<int32_t> status((size_t)n, 77);
        if (mv.empty()) mv.push_back(0);
        if (cigar.empty()) cigar.push_back(0);
        const int rc = rmr_ref_anchor_batch(n, mv.data(), mv_off.data(), sig_len.data(), seq_len.data(), cigar.data(), cigar_off.data(),
                                            rev.data(), ref_len.data(), out, r2s_off.data(), status.data(), 1 + (int)(rng() % 6));
        if (rc != 0) { printf("rmr_ref_anchor_batch rc %d\n", rc); return 1; }
        for (int64_t i = 0; i < n; ++i) {
            if (status[i] == 77) { printf("status not written\n"); return 1; }
            if (status[i] == 0) {
                ++anchored;
                const int64_t *m = out + r2s_off[i];
                for (int64_t k = 1; k <= ref_len[i]; ++k)
                    if (m[k] < m[k - 1]) { printf("reference-to-signal knots not ascending\n"); return 1; }
            } else {
                ++turned_away;
            }
        }
        free(out);
    }
    // ---- rmr_orient_bases ----
    uint8_t comp[256];
    int8_t code[256];
    for (int c = 0; c < 256; ++c) { comp[c] = (uint8_t)c; code[c] = -1; }
    const char *from = "ACGTacgt", *to = "TGCAtgca";
    for (int k = 0; k < 8; ++k) comp[(uint8_t)from[k]] = (uint8_t)to[k];
    code['A'] = 0; code['C'] = 1; code['G'] = 2; code['T'] = 3;
    long bases = 0;
    for (int trial = 0; trial < 3000; ++trial) {
        const int64_t blob_n = (int64_t)(rng() % 3000);
        uint8_t *blob = (uint8_t *)malloc((size_t)(blob_n ? blob_n : 1));
        for (int64_t i = 0; i < blob_n; ++i) blob[i] = trial % 4 ? (uint8_t)"ACGTNacgtn"[rng() % 10] : (uint8_t)(rng() & 255);
        const int64_t n = (int64_t)(rng() % 30);
        std::vector<int64_t> start, len;
        std::vector<uint8_t> rev;
        int64_t total = 0;
        for (int64_t i = 0; i < n; ++i) {
            const int64_t s = blob_n ? (int64_t)(rng() % (uint64_t)blob_n) : 0;
            const int64_t l = blob_n ? (int64_t)(rng() % (uint64_t)(blob_n - s + 1)) : 0;
            start.push_back(s); len.push_back(rng() % 5 == 0 ? 0 : l); rev.push_back((uint8_t)(rng() & 1));
            total += len.back();
        }
        uint8_t *fwd = (uint8_t *)malloc((size_t)(total ? total : 1)), *ori = (uint8_t *)malloc((size_t)(total ? total : 1));
        int8_t *codes = (int8_t *)malloc((size_t)(total ? total : 1));
        const int upper = (int)(rng() & 1);
        const int rc = rmr_orient_bases(blob, start.data(), len.data(), rev.data(), n, upper, comp, code, trial % 3 ? fwd : nullptr, ori, codes,
                                        1 + (int)(rng() % 5));
        if (rc != 0) { printf("rmr_orient_bases rc %d\n", rc); return 1; }
        int64_t o = 0;
        for (int64_t i = 0; i < n; ++i) {
            for (int64_t j = 0; j < len[i]; ++j) {
                uint8_t b = blob[start[i] + (rev[i] ? len[i] - 1 - j : j)];
                if (upper && b >= 'a' && b <= 'z') b = (uint8_t)(b - 32);
                if (rev[i]) b = comp[b];
                if (ori[o + j] != b || codes[o + j] != code[b]) { printf("orient mismatch\n"); return 1; }
            }
            o += len[i];
            bases += len[i];
        }
        free(blob); free(fwd); free(ori); free(codes);
    }
    printf("%ld records anchored, %ld turned away; %ld bases oriented\n", anchored, turned_away, bases);
    return 0;
}
