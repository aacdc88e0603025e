/* A plain C99 caller that INFERS through the C ABI: reads a fixture (tools/export_c_fixture.py: reference weights in
 * rmr_model_create order, chunk arrays, the reference's logits), then
 *     rmr_engine_create -> rmr_model_create -> rmr_infer_chunks(RMR_MEM_HOST, with label counts) -> rmr_count_labels
 * and compares: logits within 1e-4 of the reference's, the fused call's label tally equal to rmr_count_labels on the
 * returned logits and to the argmax histogram of the reference's logits.  Exit code 0 = all of it held; every non-zero
 * return code of the library is printed with rmr_last_error().  This is the interface that stands in for
 * encoded_kmers.pyx:13-45 + models/ConvLSTM_w_ref.py:39-58 (include/remora_hip.h), with no Python in the process.
 *
 *     infer_from_c FIXTURE.bin [dtype]        dtype: 0 fp32 (default), 1 bf16, 2 bf16x3, 3 bf16x6, 4 f16, 5 f16x3; tolerance
 *                                             1e-4 for fp32 / bf16x6 / f16x3, 5e-4 for bf16x3, 3e-2 for bf16, 4e-3 for f16 */
#include "remora_hip.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != 0) {                                                               \
            fprintf(stderr, "%s -> rc %d: %s\n", #call, rc_, rmr_last_error());       \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

static void *read_block(FILE *fh, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (p == NULL || fread(p, 1, bytes, fh) != bytes) {
        fprintf(stderr, "fixture: short read of %lu bytes\n", (unsigned long)bytes);
        exit(3);
    }
    return p;
}

int main(int argc, char **argv) {
    FILE *fh;
    char magic[4];
    int32_t h[9];
    int64_t nn[2], n, i, j, counts[16], recount[16], want[16];
    rmr_model_desc d;
    rmr_engine *e = NULL;
    rmr_model *m = NULL;
    float *weights, *signal, *ref, *logits;
    int8_t *seqs;
    int16_t *maps, *lens;
    int L, num_out, kb, ka, seq_w, map_w, dtype, bad = 0;
    double tol, worst = 0.0;

    if (argc < 2) {
        fprintf(stderr, "usage: %s FIXTURE.bin [dtype]\n", argv[0]);
        return 3;
    }
    dtype = argc > 2 ? atoi(argv[2]) : 0;
    tol = dtype == 1 ? 3e-2 : dtype == 2 ? 5e-4 : dtype == 4 ? 4e-3 : 1e-4;
    fh = fopen(argv[1], "rb");
    if (fh == NULL || fread(magic, 1, 4, fh) != 4 || memcmp(magic, "RMRC", 4) != 0 || fread(h, 4, 9, fh) != 9 || fread(nn, 8, 2, fh) != 2) {
        fprintf(stderr, "fixture: cannot read the header of %s\n", argv[1]);
        return 3;
    }
    memset(&d, 0, sizeof d);
    d.arch = h[0];
    d.size = h[1];
    d.kmer_len = h[2];
    d.num_out = num_out = h[3];
    d.chunk_len = L = h[4];
    d.dtype = dtype;
    kb = h[5];
    ka = h[6];
    seq_w = h[7];
    map_w = h[8];
    n = nn[0];
    if (num_out > 16 || rmr_model_weight_count(&d) != (size_t)nn[1]) {
        fprintf(stderr, "fixture holds %ld weights, rmr_model_weight_count says %lu\n", (long)nn[1], (unsigned long)rmr_model_weight_count(&d));
        return 3;
    }
    weights = (float *)read_block(fh, (size_t)nn[1] * 4);
    signal = (float *)read_block(fh, (size_t)n * L * 4);
    seqs = (int8_t *)read_block(fh, (size_t)n * seq_w);
    maps = (int16_t *)read_block(fh, (size_t)n * map_w * 2);
    lens = (int16_t *)read_block(fh, (size_t)n * 2);
    ref = (float *)read_block(fh, (size_t)n * num_out * 4);
    fclose(fh);
    logits = (float *)calloc((size_t)n * num_out, 4);
    memset(counts, 0, sizeof counts);
    memset(recount, 0, sizeof recount);
    memset(want, 0, sizeof want);

    printf("%s\n", rmr_version());
    CHECK(rmr_engine_create(0, NULL, RMR_ENGINE_OWN_STREAM, &e));
    CHECK(rmr_model_create(e, &d, weights, (size_t)nn[1], &m));
    CHECK(rmr_infer_chunks(m, signal, seqs, seq_w, maps, map_w, lens, kb, ka, n, logits, counts, RMR_MEM_HOST));
    CHECK(rmr_count_labels(e, logits, n, num_out, recount, RMR_MEM_HOST));
    CHECK(rmr_engine_synchronize(e));

    for (i = 0; i < n; i++) {
        int am = 0;
        for (j = 0; j < num_out; j++) {
            double dlt = fabs((double)logits[i * num_out + j] - (double)ref[i * num_out + j]);
            if (!(dlt <= worst)) worst = dlt; /* also catches NaN */
            if (ref[i * num_out + j] > ref[i * num_out + am]) am = (int)j;
        }
        want[am]++;
    }
    printf("chunks=%ld max_abs_diff=%.3e tol=%.0e\n", (long)n, worst, tol);
    if (!(worst <= tol)) bad |= 1;
    for (j = 0; j < num_out; j++) {
        printf("label %ld: fused=%ld recount=%ld reference_argmax=%ld\n", (long)j, (long)counts[j], (long)recount[j], (long)want[j]);
        if (counts[j] != recount[j]) bad |= 2;
        if (dtype == 0 && counts[j] != want[j]) bad |= 4;
    }
    /* an invalid call must come back as an error code + message, not a crash */
    if (rmr_infer_chunks(m, signal, seqs, seq_w, maps, map_w, lens, kb + 1, ka, n, logits, NULL, RMR_MEM_HOST) == 0) bad |= 8;
    else printf("bad k-mer context refused: %s\n", rmr_last_error());
    rmr_model_destroy(m);
    rmr_engine_destroy(e);
    printf(bad ? "FAILED (mask %d)\n" : "OK%.0d\n", bad);
    return bad ? 1 : 0;
}
