/* The sharding side of the C ABI from plain C99, no GPU involved: N workers split a BAM by byte ranges with
 * rmr_bam_guess_start (no pass over the file), each reads its share with rmr_bam_read_batch up to the record the next
 * share begins with, and the shares are checked to be exactly the file's records, in order (tests/test_host_cpu.py builds
 * and runs this file: `bam_shares_from_c <bam> <workers>`; exit code 0 = the shares partition the file).
 * The Python host does the same in io.bam_byte_shard / _iter_bam_records_native. */
#include "remora_hip.h"
#include <stdio.h>
#include <stdlib.h>

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s: rc=%d %s\n", #call, rc_, rmr_last_error()); return 2; } } while (0)

static int64_t file_size(const char *path) {
    FILE *f = fopen(path, "rb");
    long n;
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    n = ftell(f);
    fclose(f);
    return (int64_t)n;
}

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : NULL;
    const int workers = argc > 2 ? atoi(argv[2]) : 4;
    int64_t size, total = 0, whole = 0, next_expected = -1;
    rmr_bam *b = NULL;
    rmr_bam_batch bb;
    int w;
    if (!path || workers < 1) { fprintf(stderr, "usage: bam_shares_from_c <bam> <workers>\n"); return 2; }
    size = file_size(path);
    if (size < 0) { fprintf(stderr, "cannot open %s\n", path); return 2; }
    /* the whole file, for the count */
    CHECK(rmr_bam_open(path, &b));
    for (;;) {
        CHECK(rmr_bam_read_batch(b, 256, 0, &bb));
        whole += bb.n_records;
        if (bb.n_records < 256) break;
    }
    rmr_bam_close(b);
    for (w = 0; w < workers; ++w) {
        int64_t start, end = -1, mine = 0, i;
        int done = 0;
        CHECK(rmr_bam_open(path, &b));
        if (w + 1 < workers) CHECK(rmr_bam_guess_start(b, size * (w + 1) / workers, &end));   /* where the next share begins */
        CHECK(rmr_bam_guess_start(b, size * w / workers, &start));                            /* leaves the handle there */
        if (next_expected >= 0 && start != next_expected) { fprintf(stderr, "worker %d starts at %lld, the share in front ended at %lld\n", w, (long long)start, (long long)next_expected); return 1; }
        if (start >= 0 && start != end) {
            while (!done) {
                CHECK(rmr_bam_read_batch(b, 64, 0, &bb));
                for (i = 0; i < bb.n_records; ++i) {
                    if (end >= 0 && bb.voffset[i] >= end) {
                        if (bb.voffset[i] != end) { fprintf(stderr, "worker %d ran past its end mark\n", w); return 1; }
                        done = 1;
                        break;
                    }
                    ++mine;
                }
                if (bb.n_records < 64) done = 1;
            }
        }
        printf("worker %d: %lld records from voffset %lld to %lld\n", w, (long long)mine, (long long)start, (long long)end);
        total += mine;
        next_expected = end;
        rmr_bam_close(b);
    }
    printf("%lld records in the file, %lld in the shares\n", (long long)whole, (long long)total);
    return total == whole ? 0 : 1;
}
