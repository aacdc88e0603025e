/* AddressSanitizer harness for csrc/pyset_order.c (no CPython involved): motif scans of ragged batches into an output of the
 * exact size the contract asks for (one entry per base), several threads, reads of every length around the table's growth
 * points; every read's focus bases must be distinct positions inside the read, as many as a plain scan counts. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int64_t rmr_py_focus_bases_set_order(const int8_t *iseq, const int64_t *seq_off, int64_t n_reads, int32_t n_motifs, const int32_t *mot_len,
                                     const int32_t *mot_focus, const uint8_t *mot_mask, int64_t *focus, int64_t *foc_off, int32_t threads);
int64_t rmr_py_set_order(const int64_t *keys, int64_t n, int64_t *out);

static uint64_t s = 88172645463325252ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

int main(void) {
    long reads = 0, hits = 0;
    for (int trial = 0; trial < 3000; ++trial) {
        const int64_t n = 1 + (int64_t)(rnd() % 24);
        int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
        off[0] = 0;
        for (int64_t g = 0; g < n; ++g) off[g + 1] = off[g] + (int64_t)(trial % 3 ? rnd() % 700 : rnd() % 9);
        const int64_t total = off[n];
        int8_t *seq = (int8_t *)malloc((size_t)(total ? total : 1));
        for (int64_t i = 0; i < total; ++i) seq[i] = (int8_t)(trial % 5 ? rnd() % 4 : (int64_t)(rnd() % 6) - 1);  /* -1 and 4 occur */
        int32_t n_mot = 1 + (int32_t)(rnd() % 3), len[3], foc[3];
        uint8_t mask[3 * 16];
        for (int m = 0; m < n_mot; ++m) {
            len[m] = 1 + (int32_t)(rnd() % (trial % 7 ? 4 : 16));
            foc[m] = (int32_t)(rnd() % (uint64_t)len[m]);
            for (int k = 0; k < 16; ++k) mask[16 * m + k] = (uint8_t)(1 + rnd() % 15);
        }
        int64_t *focus = (int64_t *)malloc(sizeof(int64_t) * (size_t)(total ? total : 1));
        int64_t *foc_off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
        const int64_t got = rmr_py_focus_bases_set_order(seq, off, n, n_mot, len, foc, mask, focus, foc_off, 1 + (int32_t)(rnd() % 5));
        if (got < 0 || got != foc_off[n]) { printf("rc %lld\n", (long long)got); return 1; }
        for (int64_t g = 0; g < n; ++g) {
            const int64_t rl = off[g + 1] - off[g];
            uint8_t *seen = (uint8_t *)calloc((size_t)(rl ? rl : 1), 1);
            int64_t want = 0;
            for (int64_t b = 0; b < rl; ++b) {
                int any = 0;
                for (int m = 0; m < n_mot && !any; ++m) {
                    const int64_t j = b - foc[m];
                    if (j < 0 || j + len[m] > rl) continue;
                    int ok = 1;
                    for (int k = 0; k < len[m] && ok; ++k) {
                        const int c = seq[off[g] + j + k];
                        ok = c >= 0 && c < 4 && ((mask[16 * m + k] >> c) & 1);
                    }
                    any = ok;
                }
                want += any;
            }
            if (foc_off[g + 1] - foc_off[g] != want) { printf("count mismatch\n"); return 1; }
            for (int64_t k = foc_off[g]; k < foc_off[g + 1]; ++k) {
                if (focus[k] < 0 || focus[k] >= rl || seen[focus[k]]) { printf("position outside the read or twice\n"); return 1; }
                seen[focus[k]] = 1;
            }
            free(seen);
            hits += want;
            ++reads;
        }
        free(off); free(seq); free(focus); free(foc_off);
    }
    /* the table alone across its growth points, with duplicates */
    for (int64_t n = 0; n < 3000; n += 1 + n / 7) {
        int64_t *keys = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1)), *out = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
        for (int64_t i = 0; i < n; ++i) keys[i] = (int64_t)(rnd() % (uint64_t)(n + 3));
        const int64_t cnt = rmr_py_set_order(keys, n, out);
        if (cnt < 0 || cnt > n) { printf("set order rc\n"); return 1; }
        free(keys); free(out);
    }
    printf("%ld reads, %ld focus bases\n", reads, hits);
    return 0;
}
