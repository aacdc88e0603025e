/* pyset_order.c (part of _pyglue.so) - the focus bases of a batch of reads in the order a CPython `set` iterates them.
 *
 * The reference collects the motif hits of a read in a python set and iterates it (src/remora/util.py:413-426); the order of
 * the chunks in a prepared dataset - and which focus bases np.random.choice keeps - follows that iteration order, so a
 * byte-identical dataset needs it.  Through the interpreter that is a list, a set and an iterator per read: 30 us for the 300
 * hits of a 5 kb read, the largest item of `dataset prepare`'s host time per read.  Here the same order on raw arrays: the
 * open-addressing table of Objects/setobject.c (CPython 3.7 - 3.12: LINEAR_PROBES 9, PERTURB_SHIFT 5, growth when
 * fill * 5 >= mask * 3 to the first power of two above 4 x used, 2 x used beyond 50 000 entries), restated for keys that are
 * non-negative integers below 2^61 - 1 (their hash is the integer itself) and for insertion only (no dummies).  The caller
 * checks the restatement against the interpreter's own set when the module is loaded (data_chunks._set_order_glue) and keeps
 * the interpreter's path when they differ; tests/test_host_cpu.py compares them on random batches.
 *
 * No CPython API in this file; it lives in _pyglue.so because only a Python host needs it. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EMPTY ((int64_t)-1)
#define LINEAR_PROBES 9
#define PERTURB_SHIFT 5

typedef struct {
    int64_t *table;
    size_t mask, fill, cap; /* cap: allocated entries */
} iset;

static void insert_clean(int64_t *table, size_t mask, int64_t key) {
    size_t perturb = (size_t)key, i = (size_t)key & mask;
    for (;;) {
        size_t e = i;
        if (table[e] == EMPTY) { table[e] = key; return; }
        if (i + LINEAR_PROBES <= mask)
            for (int j = 0; j < LINEAR_PROBES; ++j)
                if (table[++e] == EMPTY) { table[e] = key; return; }
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & mask;
    }
}

static int resize(iset *s, int64_t **spare, size_t *spare_cap, size_t minused) {
    size_t newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    if (*spare_cap < newsize) {
        free(*spare);
        *spare = (int64_t *)malloc(newsize * sizeof(int64_t));
        if (!*spare) { *spare_cap = 0; return -1; }
        *spare_cap = newsize;
    }
    int64_t *nt = *spare;
    memset(nt, 0xff, newsize * sizeof(int64_t));
    for (size_t e = 0; e <= s->mask; ++e)
        if (s->table[e] != EMPTY) insert_clean(nt, newsize - 1, s->table[e]);
    /* the old table becomes the spare */
    *spare = s->table; { size_t c = s->cap; s->cap = *spare_cap; *spare_cap = c; }
    s->table = nt;
    s->mask = newsize - 1;
    return 0;
}

static int add(iset *s, int64_t **spare, size_t *spare_cap, int64_t key) {
    const size_t mask = s->mask;
    size_t perturb = (size_t)key, i = (size_t)key & mask, e;
    for (;;) {
        int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
        e = i;
        do {
            if (s->table[e] == EMPTY) goto found_unused;
            if (s->table[e] == key) return 0;
            ++e;
        } while (probes--);
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & mask;
    }
found_unused:
    s->table[e] = key;
    s->fill++;
    if (s->fill * 5 < mask * 3) return 0;
    return resize(s, spare, spare_cap, s->fill > 50000 ? s->fill * 2 : s->fill * 4);
}

/* One contiguous range of reads; read g's focus bases are written at focus + seq_off[g] (room for its bases), their number to
 * foc_off[g + 1]. */
typedef struct {
    const int8_t *iseq; const int64_t *seq_off; int64_t g0, g1; int32_t n_motifs; const int32_t *mot_len, *mot_focus;
    const uint8_t *mot_mask; int64_t *focus, *foc_off; int rc;
} job;

static void *run_job(void *arg) {
    job *w = (job *)arg;
    iset s = {0};
    int64_t *spare = NULL;
    size_t spare_cap = 0;
    s.table = (int64_t *)malloc(8 * sizeof(int64_t));
    if (!s.table) { w->rc = -2; return NULL; }
    s.cap = 8;
    for (int64_t g = w->g0; g < w->g1 && w->rc == 0; ++g) {
        const int8_t *q = w->iseq + w->seq_off[g];
        const int64_t n = w->seq_off[g + 1] - w->seq_off[g];
        /* a fresh set(): the eight-entry small table */
        memset(s.table, 0xff, 8 * sizeof(int64_t));
        s.mask = 7;
        s.fill = 0;
        for (int m = 0; m < w->n_motifs && w->rc == 0; ++m) {
            /* Shift-And scan: bit k of `state` = the last k + 1 bases match the motif's first k + 1 positions */
            const int len = w->mot_len[m];
            const uint8_t *mk = w->mot_mask + 16 * m;
            uint32_t allow[4] = {0, 0, 0, 0};
            for (int k = 0; k < len; ++k)
                for (int c = 0; c < 4; ++c)
                    if ((mk[k] >> c) & 1) allow[c] |= 1u << k;
            const uint32_t last = 1u << (len - 1);
            const int64_t back = len - 1 - w->mot_focus[m];
            uint32_t state = 0;
            for (int64_t i = 0; i < n; ++i) {
                const unsigned c = (unsigned)(int)q[i];
                state = ((state << 1) | 1u) & (c < 4 ? allow[c] : 0u);
                if ((state & last) && add(&s, &spare, &spare_cap, i - back) != 0) { w->rc = -2; break; }
            }
        }
        int64_t *out = w->focus + w->seq_off[g], cnt = 0;
        for (size_t e = 0; e <= s.mask; ++e)
            if (s.table[e] != EMPTY) out[cnt++] = s.table[e];
        w->foc_off[g + 1] = cnt;
    }
    free(s.table);
    free(spare);
    return NULL;
}

/* iseq: base codes -1..3 of all reads, read g at [seq_off[g], seq_off[g+1]), seq_off[0] = 0.  Motif m: mot_len[m] positions,
 * position k allows code c when bit c of mot_mask[m * 16 + k] is set; a hit starting at j contributes j + mot_focus[m].  For
 * every read the hits of motif 0 are inserted in ascending order, then motif 1's, ... (set.update of one list per motif), and
 * the table is read out slot by slot.  focus: room for seq_off[n_reads] entries (a read cannot have more distinct focus bases
 * than bases); foc_off: n_reads + 1.  The reads are dealt to `threads` native threads in contiguous ranges.  Returns the number
 * of focus bases, -2 out of memory, -3 a motif the restatement does not cover (focus position outside the motif: keys could be
 * negative). */
int64_t rmr_py_focus_bases_set_order(const int8_t *iseq, const int64_t *seq_off, int64_t n_reads, int32_t n_motifs,
                                     const int32_t *mot_len, const int32_t *mot_focus, const uint8_t *mot_mask,
                                     int64_t *focus, int64_t *foc_off, int32_t threads) {
    for (int m = 0; m < n_motifs; ++m)
        if (mot_len[m] < 1 || mot_len[m] > 16 || mot_focus[m] < 0 || mot_focus[m] >= mot_len[m]) return -3;
    if (threads < 1) threads = 1;
    if (threads > 16) threads = 16;
    if (n_reads < 4 * threads) threads = 1;
    job jobs[16];
    pthread_t tid[16];
    const int64_t total_bases = n_reads > 0 ? seq_off[n_reads] : 0;
    int64_t g0 = 0;
    int used = 0;
    for (int t = 0; t < threads && g0 < n_reads; ++t) {
        int64_t g1 = g0;
        const int64_t want = total_bases * (t + 1) / threads;
        while (g1 < n_reads && (seq_off[g1 + 1] <= want || g1 == g0)) ++g1;
        if (t == threads - 1) g1 = n_reads;
        job j = {iseq, seq_off, g0, g1, n_motifs, mot_len, mot_focus, mot_mask, focus, foc_off, 0};
        jobs[used] = j;
        g0 = g1;
        ++used;
    }
    for (int t = 1; t < used; ++t)
        if (pthread_create(&tid[t], NULL, run_job, &jobs[t]) != 0) { run_job(&jobs[t]); tid[t] = 0; }
    if (used) run_job(&jobs[0]);
    int64_t rc = 0;
    for (int t = 0; t < used; ++t) {
        if (t && tid[t]) pthread_join(tid[t], NULL);
        if (jobs[t].rc) rc = jobs[t].rc;
    }
    if (rc) return rc;
    /* close the gaps: read g's entries move from seq_off[g] down to the running total (never upwards) */
    int64_t total = 0;
    foc_off[0] = 0;
    for (int64_t g = 0; g < n_reads; ++g) {
        const int64_t cnt = foc_off[g + 1];
        if (total != seq_off[g]) memmove(focus + total, focus + seq_off[g], (size_t)cnt * sizeof(int64_t));
        total += cnt;
        foc_off[g + 1] = total;
    }
    return total;
}

/* The table alone, for the load-time check and the tests: keys (non-negative, < 2^61 - 1) inserted in the given order, the
 * iteration order written to out (room for n).  Returns the number of distinct keys or -2. */
int64_t rmr_py_set_order(const int64_t *keys, int64_t n, int64_t *out) {
    iset s = {0};
    int64_t *spare = NULL;
    size_t spare_cap = 0;
    s.table = (int64_t *)malloc(8 * sizeof(int64_t));
    if (!s.table) return -2;
    memset(s.table, 0xff, 8 * sizeof(int64_t));
    s.cap = 8; s.mask = 7;
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; ++i)
        if (keys[i] < 0 || add(&s, &spare, &spare_cap, keys[i]) != 0) { cnt = -2; break; }
    if (cnt == 0)
        for (size_t e = 0; e <= s.mask; ++e)
            if (s.table[e] != EMPTY) out[cnt++] = s.table[e];
    free(s.table);
    free(spare);
    return cnt;
}
