/*
 * remora_oracle.c — CPU restatement (plain C99) of the reference algorithms on the
 * per-read modified-base-call hot path of nanoporetech/remora v3.2.0.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library, and only as the checker.  The
 * product path (remora_amd/, libremora_hip.so) never links, imports or calls it.
 *
 * Parity status: PINNED.  Every function here is checked against golden vectors produced
 * by running the reference itself (tools/gen_golden.py -> the .npz files under tests/golden); see
 * tests/test_oracle_golden.py.
 *
 * Each function cites the reference file:line it follows (paths relative to the
 * reference checkout).  Nothing is copied: the reference is Cython/Python/torch.nn, this
 * is a from-scratch C statement of the same arithmetic.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------
 * E1  k-mer one-hot encode with move-table (seq->signal) expansion.
 *     reference: src/remora/encoded_kmers.pyx:13-45 (compute_encoded_kmer_batch)
 *     out[c, 4*kp + seqs[c, p+kp], s] = 1 for p in [0,len_c), kp in [0,K), s in
 *     [map[c,p], map[c,p+1]); bases == -1 skipped; sig_len = map[0, len_0] (:23).
 *     `out` must hold n * 4K * sig_len floats; it is zero-filled here (:26).
 * ---------------------------------------------------------------------------------- */
ORC_API int orc_encode_kmers(int kb, int ka, const int8_t *seqs, int seq_w,
                             const int16_t *maps, int map_w, const int16_t *lens,
                             int64_t n, float *out, int *sig_len_out) {
    if (n <= 0) return -1;
    const int sig_len = maps[lens[0]];
    const int K = kb + ka + 1;
    const int rows = 4 * K;
    if (sig_len_out) *sig_len_out = sig_len;
    if (!out) return 0;
    memset(out, 0, (size_t)n * rows * sig_len * sizeof(float));
    for (int64_t c = 0; c < n; ++c) {
        const int sl = lens[c];
        const int8_t *cs = seqs + c * seq_w;
        const int16_t *cm = maps + c * map_w;
        float *co = out + (size_t)c * rows * sig_len;
        for (int kp = 0; kp < K; ++kp) {
            for (int p = 0; p < sl; ++p) {
                const int base = cs[p + kp];
                if (base == -1) continue;
                const int st = cm[p], en = cm[p + 1];
                float *row = co + (size_t)(4 * kp + base) * sig_len;
                for (int s = st; s < en; ++s) row[s] = 1.0f;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * T1  trim stored chunk context to a smaller model context, in place.
 *     reference: src/remora/data_chunks_core.pyx:10-45 (trim_sb_chunk_context_core);
 *     the caller has already subtracted st_diff from `maps` (data_chunks.py:1555-1563).
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_trim_chunk_context(int stored_before, int stored_after, int cc_before,
                                    int cc_after, int total_seq_context, int8_t *seqs,
                                    int seq_w, int16_t *maps, int map_w, int16_t *lens,
                                    int64_t n) {
    const int16_t cc_width = (int16_t)(cc_before + cc_after);
    if (stored_before > cc_before) {
        for (int64_t c = 0; c < n; ++c) {
            int16_t *cm = maps + c * map_w;
            int8_t *cs = seqs + c * seq_w;
            int16_t st_clip = 0;
            while (cm[st_clip + 1] <= 0) st_clip++;
            const int16_t sl = lens[c];
            for (int i = 0; i < sl + 1 - st_clip; ++i) cm[i] = cm[st_clip + i];
            for (int i = 0; i < sl + total_seq_context - st_clip; ++i) cs[i] = cs[i + st_clip];
            lens[c] = (int16_t)(sl - st_clip);
            cm[0] = 0;
        }
    }
    if (stored_after > cc_after) {
        for (int64_t c = 0; c < n; ++c) {
            int16_t *cm = maps + c * map_w;
            while (cm[lens[c] - 1] >= cc_width) lens[c]--;
            cm[lens[c]] = cc_width;
        }
    }
}

/* ------------------------------------------------------------------------------------
 * M1  move-table expansion.
 *     reference: src/remora/io.py:394-407 (parse_move_tag)
 *     q2s = concat(nonzero(mv[1:]) * stride, [sig_len]); reverse: sig_len - q2s[::-1].
 *     returns number of entries written (= #moves + 1), or
 *       -1 "Move table discordant with basecalls", -2 "... with signal".
 * ---------------------------------------------------------------------------------- */
ORC_API int64_t orc_parse_move_tag(const int8_t *mv_tag, int64_t mv_tag_len, int64_t sig_len,
                                   int64_t seq_len /* <0: None */, int check, int reverse,
                                   int64_t *q2s /* cap >= mv_tag_len */) {
    const int64_t stride = mv_tag[0];
    const int64_t nmv = mv_tag_len - 1;
    int64_t k = 0;
    for (int64_t i = 0; i < nmv; ++i)
        if (mv_tag[1 + i] != 0) q2s[k++] = i * stride;
    q2s[k++] = sig_len;
    if (reverse) {
        for (int64_t i = 0; i < k / 2; ++i) {
            int64_t a = q2s[i], b = q2s[k - 1 - i];
            q2s[i] = sig_len - b;
            q2s[k - 1 - i] = sig_len - a;
        }
        if (k & 1) q2s[k / 2] = sig_len - q2s[k / 2];
    }
    if (check && seq_len >= 0 && k - 1 != seq_len) return -1;
    if (check && nmv != sig_len / stride) return -2;
    return k;
}

/* ------------------------------------------------------------------------------------
 * X1  signal normalisation: ((dacs - shift) / scale).astype(float32), arithmetic in
 *     float64.  reference: src/remora/data_chunks.py:191-197 (RemoraRead.sig)
 * ---------------------------------------------------------------------------------- */
ORC_API void orc_normalise_signal(const int16_t *dacs, int64_t n, double shift, double scale,
                                  float *sig) {
    for (int64_t i = 0; i < n; ++i) sig[i] = (float)(((double)dacs[i] - shift) / scale);
}

static int64_t searchsorted_right(const int64_t *a, int64_t n, int64_t v) {
    int64_t lo = 0, hi = n; /* first idx with a[idx] > v */
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
static int64_t searchsorted_left(const int64_t *a, int64_t n, int64_t v) {
    int64_t lo = 0, hi = n; /* first idx with a[idx] >= v */
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* ------------------------------------------------------------------------------------
 * X2+X3  chunk geometry for one focus base.
 *     reference: src/remora/data_chunks.py:425-466 (iter_chunks: offset/clip, focus
 *     signal index) and :331-423 (extract_chunk: signal window with zero padding, the
 *     two searchsorted calls, mapping shift with forced ends, -1 filled context seq).
 *     Outputs (caller-sized): signal[chunk_len]; seq_out[cap_seq] (first seq_len+kb+ka
 *     valid); map_out[cap_map] (first seq_len+1 valid); geo[5] = {seq_len,
 *     chunk_sig_focus_idx, chunk_focus_base, read_focus_base, seq_start}.
 *     returns 0, or -1 if capacity is too small.
 * ---------------------------------------------------------------------------------- */
ORC_API int orc_extract_chunk(const float *sig, int64_t sig_len, const int64_t *map,
                              int64_t nbases, const int8_t *int_seq, int64_t focus_base_in,
                              int cc_before, int cc_after, int kb, int ka,
                              int base_start_justify, int offset, float *signal,
                              int8_t *seq_out, int64_t cap_seq, int32_t *map_out,
                              int64_t cap_map, int64_t *geo) {
    const int64_t map_size = nbases + 1;
    int64_t fb = focus_base_in + offset;                       /* :446-448 */
    if (fb > map_size - 2) fb = map_size - 2;
    if (fb < 0) fb = 0;
    const int64_t focus_sig = base_start_justify ? map[fb] : (map[fb] + map[fb + 1]) / 2;
    const int chunk_len = cc_before + cc_after;
    int64_t sig_start = focus_sig - cc_before, sig_end = focus_sig + cc_after;
    int64_t s2s_off = 0;
    if (sig_start >= 0 && sig_end <= sig_len) {                /* :345-347 */
        memcpy(signal, sig + sig_start, chunk_len * sizeof(float));
    } else {                                                   /* :348-368 */
        for (int i = 0; i < chunk_len; ++i) signal[i] = 0.0f;
        int64_t fill_st = 0, fill_en = chunk_len;
        if (sig_start < 0) { fill_st = -sig_start; s2s_off = -sig_start; sig_start = 0; }
        if (sig_end > sig_len) { fill_en = sig_len - sig_start + s2s_off; sig_end = sig_len; }
        for (int64_t i = fill_st; i < fill_en; ++i) signal[i] = sig[sig_start + (i - fill_st)];
    }
    const int64_t seq_start = searchsorted_right(map, map_size, sig_start) - 1; /* :370-372 */
    const int64_t seq_end = searchsorted_left(map, map_size, sig_end);          /* :373 */
    const int64_t sl = seq_end - seq_start;
    if (sl + 1 > cap_map || sl + kb + ka > cap_seq) return -1;
    for (int64_t i = 0; i <= sl; ++i)                          /* :376-382 */
        map_out[i] = (int32_t)(map[seq_start + i] - (sig_start - s2s_off));
    map_out[0] = 0;
    map_out[sl] = chunk_len;
    for (int64_t i = 0; i < sl + kb + ka; ++i) {               /* :385-409 */
        const int64_t src = seq_start - kb + i;
        seq_out[i] = (src >= 0 && src < nbases) ? int_seq[src] : (int8_t)-1;
    }
    geo[0] = sl;
    geo[1] = focus_sig - sig_start;                            /* :415 (after clipping) */
    geo[2] = fb - seq_start;
    geo[3] = fb;
    geo[4] = seq_start;
    return 0;
}

/* ------------------------------------------------------------------------------------
 * F1/F2  network forward, eval mode, fp32.  Building blocks follow torch.nn semantics as
 *     used by models/ConvLSTM_w_ref.py:39-58 and models/Conv_w_ref.py:44-62:
 *     Conv1d (valid, stride), BatchNorm1d eval (eps 1e-5), swish = x*sigmoid(x)
 *     (src/remora/activations.py:4-18), LSTM gate order i,f,g,o with zero initial state,
 *     Linear.
 * ---------------------------------------------------------------------------------- */
static float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
static float swishf_(float x) { return x * sigmoidf_(x); }

typedef struct {
    const float *w, *b;          /* conv weight [oc][ic][k], bias [oc] */
    const float *g, *beta, *mean, *var; /* batchnorm */
    int ic, oc, k, stride;
} conv_bn_t;

static void conv_bn_swish(const conv_bn_t *L, const float *in, int lin, float *out, int *lout_p) {
    const int lout = (lin - L->k) / L->stride + 1;
    for (int o = 0; o < L->oc; ++o) {
        const float inv = 1.0f / sqrtf(L->var[o] + 1e-5f);
        for (int t = 0; t < lout; ++t) {
            float acc = L->b[o];
            for (int i = 0; i < L->ic; ++i) {
                const float *wi = L->w + ((size_t)o * L->ic + i) * L->k;
                const float *xi = in + (size_t)i * lin + t * L->stride;
                for (int j = 0; j < L->k; ++j) acc += wi[j] * xi[j];
            }
            float y = (acc - L->mean[o]) * inv * L->g[o] + L->beta[o];
            out[(size_t)o * lout + t] = swishf_(y);
        }
    }
    *lout_p = lout;
}

/* one LSTM layer over T steps; x [T][H] -> hs [T][H] (torch.nn.LSTM, 1 layer) */
static void lstm_layer(const float *w_ih, const float *w_hh, const float *b_ih,
                       const float *b_hh, int H, const float *x, int T, float *hs) {
    float *h = (float *)calloc(H, sizeof(float));
    float *c = (float *)calloc(H, sizeof(float));
    float *gates = (float *)malloc((size_t)4 * H * sizeof(float));
    for (int t = 0; t < T; ++t) {
        const float *xt = x + (size_t)t * H;
        for (int r = 0; r < 4 * H; ++r) {
            float acc = b_ih[r] + b_hh[r];
            const float *wi = w_ih + (size_t)r * H, *wh = w_hh + (size_t)r * H;
            for (int k = 0; k < H; ++k) acc += wi[k] * xt[k];
            for (int k = 0; k < H; ++k) acc += wh[k] * h[k];
            gates[r] = acc;
        }
        for (int u = 0; u < H; ++u) {
            const float ig = sigmoidf_(gates[u]), fg = sigmoidf_(gates[H + u]);
            const float gg = tanhf(gates[2 * H + u]), og = sigmoidf_(gates[3 * H + u]);
            c[u] = fg * c[u] + ig * gg;
            h[u] = og * tanhf(c[u]);
            hs[(size_t)t * H + u] = h[u];
        }
    }
    free(h); free(c); free(gates);
}

/* Flat weight blob layout shared with tests (see oracle/oracle.py: pack_state_for_oracle):
 * for every conv layer in forward order: w, b, bn_g, bn_b, bn_mean, bn_var; then for
 * conv_lstm: lstm1 {w_ih, w_hh, b_ih, b_hh}, lstm2 {same}; then fc {w, b}. */
static const float *take(const float **p, size_t n) { const float *r = *p; *p += n; return r; }
static void take_conv(const float **p, conv_bn_t *L, int ic, int oc, int k, int stride) {
    L->ic = ic; L->oc = oc; L->k = k; L->stride = stride;
    L->w = take(p, (size_t)oc * ic * k); L->b = take(p, oc);
    L->g = take(p, oc); L->beta = take(p, oc); L->mean = take(p, oc); L->var = take(p, oc);
}

/* arch 0 = ConvLSTM_w_ref, 1 = Conv_w_ref.  sigs [n][1][L], seqs [n][4K][L] (dense
 * floats, exactly the tensors the reference's forward receives) -> logits [n][num_out] */
ORC_API int orc_forward(int arch, int size, int kmer_len, int num_out, int L,
                        const float *blob, const float *sigs, const float *seqs, int64_t n,
                        float *logits) {
    const int EC = 4 * kmer_len;
    const float *p = blob;
    size_t maxbuf = (size_t)2 * size * L + (size_t)EC * L + 64;
    float *a = (float *)malloc(maxbuf * sizeof(float));
    float *b = (float *)malloc(maxbuf * sizeof(float));
    float *cat = (float *)malloc(maxbuf * sizeof(float));
    if (!a || !b || !cat) return -2;
    if (arch == 0) {
        conv_bn_t s1, s2, s3, q1, q2, m1;
        take_conv(&p, &s1, 1, 4, 5, 1); take_conv(&p, &s2, 4, 16, 5, 1);
        take_conv(&p, &s3, 16, size, 9, 3);
        take_conv(&p, &q1, EC, 16, 5, 1); take_conv(&p, &q2, 16, size, 13, 3);
        take_conv(&p, &m1, 2 * size, size, 5, 1);
        const int H = size;
        const float *l1[4], *l2[4];
        l1[0] = take(&p, (size_t)4 * H * H); l1[1] = take(&p, (size_t)4 * H * H);
        l1[2] = take(&p, 4 * H); l1[3] = take(&p, 4 * H);
        l2[0] = take(&p, (size_t)4 * H * H); l2[1] = take(&p, (size_t)4 * H * H);
        l2[2] = take(&p, 4 * H); l2[3] = take(&p, 4 * H);
        const float *fw = take(&p, (size_t)num_out * H), *fb = take(&p, num_out);
        for (int64_t c = 0; c < n; ++c) {
            int l1o, l2o, ls, lq, T;
            conv_bn_swish(&s1, sigs + (size_t)c * L, L, a, &l1o);        /* :41 */
            conv_bn_swish(&s2, a, l1o, b, &l2o);                          /* :42 */
            conv_bn_swish(&s3, b, l2o, cat, &ls);                         /* :43 */
            conv_bn_swish(&q1, seqs + (size_t)c * EC * L, L, a, &l1o);    /* :45 */
            conv_bn_swish(&q2, a, l1o, cat + (size_t)size * ls, &lq);     /* :46, cat :48 */
            if (ls != lq) return -3;
            conv_bn_swish(&m1, cat, ls, a, &T);                           /* :50 */
            /* permute(2,0,1): time-major x[T][H]                          :51 */
            float *x = b;
            for (int t = 0; t < T; ++t)
                for (int u = 0; u < H; ++u) x[(size_t)t * H + u] = a[(size_t)u * T + t];
            float *hs = a;
            lstm_layer(l1[0], l1[1], l1[2], l1[3], H, x, T, hs);          /* :52 */
            for (int i = 0; i < T * H; ++i) hs[i] = swishf_(hs[i]);
            /* flip, lstm2, swish, flip, take [-1]                         :53-54 */
            for (int t = 0; t < T; ++t)
                memcpy(x + (size_t)t * H, hs + (size_t)(T - 1 - t) * H, H * sizeof(float));
            float *h2 = cat;
            lstm_layer(l2[0], l2[1], l2[2], l2[3], H, x, T, h2);
            /* after flipping back, index T-1 is reversed-time index 0 */
            float z[1024];
            for (int u = 0; u < H; ++u) z[u] = swishf_(h2[u]);
            for (int o = 0; o < num_out; ++o) {                           /* :56 */
                float acc = fb[o];
                for (int u = 0; u < H; ++u) acc += fw[(size_t)o * H + u] * z[u];
                logits[c * num_out + o] = acc;
            }
        }
    } else {
        conv_bn_t s1, s2, s3, q1, q2, q3, m1, m2, m3, m4;
        take_conv(&p, &s1, 1, 4, 11, 1); take_conv(&p, &s2, 4, 16, 11, 1);
        take_conv(&p, &s3, 16, size, 9, 3);
        take_conv(&p, &q1, EC, 16, 11, 1); take_conv(&p, &q2, 16, 32, 11, 1);
        take_conv(&p, &q3, 32, size, 9, 3);
        take_conv(&p, &m1, 2 * size, size, 5, 1); take_conv(&p, &m2, size, size, 5, 1);
        take_conv(&p, &m3, size, size, 3, 2); take_conv(&p, &m4, size, size, 3, 2);
        const float *fw = take(&p, (size_t)num_out * size * 3), *fb = take(&p, num_out);
        for (int64_t c = 0; c < n; ++c) {
            int l1o, l2o, ls, lq, t1, t2, t3, t4;
            conv_bn_swish(&s1, sigs + (size_t)c * L, L, a, &l1o);         /* :45 */
            conv_bn_swish(&s2, a, l1o, b, &l2o);
            conv_bn_swish(&s3, b, l2o, cat, &ls);
            conv_bn_swish(&q1, seqs + (size_t)c * EC * L, L, a, &l1o);    /* :49 */
            conv_bn_swish(&q2, a, l1o, b, &l2o);
            conv_bn_swish(&q3, b, l2o, cat + (size_t)size * ls, &lq);
            if (ls != lq) return -3;
            conv_bn_swish(&m1, cat, ls, a, &t1);                          /* :54-57 */
            conv_bn_swish(&m2, a, t1, b, &t2);
            conv_bn_swish(&m3, b, t2, a, &t3);
            conv_bn_swish(&m4, a, t3, b, &t4);
            if (t4 != 3) return -4;                                       /* fc in = size*3 (:42) */
            for (int o = 0; o < num_out; ++o) {                           /* flatten + fc :59-60 */
                float acc = fb[o];
                for (int i = 0; i < size * 3; ++i) acc += fw[(size_t)o * size * 3 + i] * b[i];
                logits[c * num_out + o] = acc;
            }
        }
    }
    free(a); free(b); free(cat);
    return 0;
}

/* ====================================================================================
 * N2  signal-mapping refinement core (banded dynamic programming).
 *     reference: src/remora/refine_signal_map_core.pyx
 *       adjust_seq_band :31-74, extract_levels :87-101, banded_traceback :119-148,
 *       banded_forward_dwell_penalty_step :150-253, banded_forward_vit_step :256-317,
 *       banded_forward_dp :320-400, seq_banded_dp :403-473
 *     All score arithmetic is float32, one operation per rounding (no fused multiply-add),
 *     comparisons strict '<' exactly as in the reference, so paths are reproduced bit for bit.
 * ==================================================================================== */
#define ORC_LARGE_SCORE 100.0f

ORC_API void orc_adjust_seq_band(int32_t *band, int n, int min_step) {
    int32_t *lo = band, *hi = band + n;
    const int32_t band_min = lo[0];
    for (int p = n - 2; p >= 0; --p)
        if (lo[p] > lo[p + 1] - min_step) lo[p] = lo[p + 1] - min_step;
    lo[0] = band_min;
    int p = 1;
    /* the reference has no bound here (boundscheck off: undefined past the row); stop at n */
    while (p < n && lo[p] <= lo[p - 1]) { lo[p] = lo[p - 1] + 1; p++; }
    const int32_t band_max = hi[n - 1];
    for (p = 1; p < n; ++p)
        if (hi[p] < hi[p - 1] + min_step) hi[p] = hi[p - 1] + min_step;
    hi[n - 1] = band_max;
    p = n - 2;
    while (p >= 0 && hi[p] >= hi[p + 1]) { hi[p] = hi[p + 1] - 1; p--; }
}

ORC_API void orc_extract_levels(const int32_t *int_seq, int n, const float *kmer_levels, int kmer_len,
                                int center_idx, float *levels) {
    for (int i = 0; i < n; ++i) levels[i] = 0.0f;
    for (int pos = 0; pos < n - kmer_len + 1; ++pos) {
        int idx = 0, mul = 1;
        for (int kp = 0; kp < kmer_len; ++kp) { idx += int_seq[pos + kmer_len - kp - 1] * mul; mul *= 4; }
        levels[pos + center_idx] = kmer_levels[idx];
    }
}

static float orc_sq(float s, float l) { const float t = s - l; return t * t; }

/* prev has prev_n entries, curr/cur_sig have bw entries; follows :256-317 */
static void orc_vit_step(float *curr, int32_t *tb, const float *prev, int prev_n, float level,
                         const float *sig, int bw, int bsd) {
    if (bsd == 0) {
        curr[0] = ORC_LARGE_SCORE + prev[prev_n - 1];
        tb[0] = -1;
    } else {
        curr[0] = prev[bsd - 1] + orc_sq(level, sig[0]);
        tb[0] = 0;
        prev += bsd; prev_n -= bsd;
    }
    if (prev_n == bw) prev_n -= 1;
    for (int b = 1; b < prev_n + 1; ++b) {
        const float base = orc_sq(level, sig[b]);
        const float move = prev[b - 1] + base, stay = curr[b - 1] + base;
        if (move < stay) { curr[b] = move; tb[b] = 0; }
        else { curr[b] = stay; tb[b] = tb[b - 1] + 1; }
    }
    for (int b = prev_n + 1; b < bw; ++b) {
        curr[b] = curr[b - 1] + orc_sq(level, sig[b]);
        tb[b] = tb[b - 1] + 1;
    }
}

/* follows :150-253 */
static void orc_dwell_step(float *curr, int32_t *tb, const float *prev, int prev_n, float level,
                           const float *sig, int bw, int bsd, const float *sdp, int d,
                           float *unpen, int32_t *unpen_tb) {
    orc_vit_step(unpen, unpen_tb, prev, prev_n, level, sig, bw, bsd);
    for (int b = 0; b < bw; ++b) {
        if (b + bsd - prev_n >= d) {
            curr[b] = curr[b - 1] + orc_sq(level, sig[b]);
            tb[b] = tb[b - 1] + 1;
            continue;
        }
        curr[b] = ORC_LARGE_SCORE + prev[prev_n - 1];
        tb[b] = -1;
        if (b == 0 && bsd == 0) continue;
        float run = 0.0f;
        for (int di = 0; di < d; ++di) {
            if (di > b || (bsd == 0 && b == di)) break;
            run += orc_sq(level, sig[b - di]);
            if (b - di - 1 + bsd >= prev_n) continue;
            const float ps = prev[b - di - 1 + bsd] + run + sdp[di];
            if (ps < curr[b]) { curr[b] = ps; tb[b] = di; }
        }
        if (b >= d) {
            const float ps = unpen[b - d] + run;
            if (ps < curr[b]) { curr[b] = ps; tb[b] = unpen_tb[b - d] + d; }
        }
    }
}

/* seq_band: [2][n] (lower row then upper row); base_offsets: [n+1] (uint32 in the reference);
 * all_scores / traceback: [band_len]; path: [n+1].  method 0 = Viterbi, 1 = dwell_penalty.
 * returns 0, or -1 on allocation failure. */
ORC_API int orc_seq_banded_dp(const float *signal, const float *levels, int n, const int32_t *seq_band,
                              const float *sdp, int d, int method, float *all_scores, int32_t *path,
                              int32_t *traceback, uint32_t *base_offsets) {
    const int32_t *lo = seq_band, *hi = seq_band + n;
    base_offsets[0] = 0;
    int maxbw = 0;
    for (int b = 0; b < n; ++b) {
        base_offsets[b + 1] = base_offsets[b] + (uint32_t)(hi[b] - lo[b]);
        if (hi[b] - lo[b] > maxbw) maxbw = hi[b] - lo[b];
    }
    float *unpen = (float *)malloc((size_t)(maxbw + 1) * sizeof(float));
    int32_t *unpen_tb = (int32_t *)malloc((size_t)(maxbw + 1) * sizeof(int32_t));
    int bw = hi[0];
    float *spoof = (float *)malloc((size_t)(bw + 1) * sizeof(float));
    if (!unpen || !unpen_tb || !spoof) return -1;
    for (int i = 0; i < bw; ++i) spoof[i] = HUGE_VALF;
    spoof[0] = 0.0f;
    if (method == 0) orc_vit_step(all_scores, traceback, spoof, bw, levels[0], signal, bw, 1);
    else orc_dwell_step(all_scores, traceback, spoof, bw, levels[0], signal, bw, 1, sdp, d, unpen, unpen_tb);
    int prev_bw = bw, prev_st = 0;
    uint32_t prev_off = 0;
    for (int b = 1; b < n; ++b) {
        const int st = lo[b], en = hi[b];
        bw = en - st;
        const uint32_t off = base_offsets[b];
        if (method == 0)
            orc_vit_step(all_scores + off, traceback + off, all_scores + prev_off, prev_bw, levels[b], signal + st, bw, st - prev_st);
        else
            orc_dwell_step(all_scores + off, traceback + off, all_scores + prev_off, prev_bw, levels[b], signal + st, bw,
                           st - prev_st, sdp, d, unpen, unpen_tb);
        prev_st = st; prev_bw = bw; prev_off = off;
    }
    /* traceback :119-148 */
    path[0] = 0;
    path[n] = hi[n - 1];
    for (int b = n - 1; b > 0; --b) {
        const int look = path[b + 1] - 1;
        path[b] = look - traceback[base_offsets[b] + look - lo[b]];
    }
    free(unpen); free(unpen_tb); free(spoof);
    return 0;
}

/* ====================================================================================
 * N1  POD5 signal rows: the VBZ layer below zstd (streamvbyte16 -> zigzag -> running sum).
 *     The algorithm lives in pod5's C++ library (pod5 >= 0.2, `svb16` + `decompress_signal`), a
 *     dependency of the reference that is absent here; this restates its published format:
 *     ceil(n/8) key bytes (bit j of byte i = sample 8i+j: 0 one data byte, 1 two, little endian),
 *     then the data bytes; value -> (v >> 1) ^ -(v & 1); samples = prefix sums in int16.
 *     Pinned on the reference's own tests/data/can_reads.pod5: the decoded signals feed the
 *     reference's Read/call_read_mods golden (signal checksums in tests/golden/real_reads_can.npz).
 *     returns 0, or -1 when the byte count disagrees with the keys.
 * ==================================================================================== */
ORC_API int orc_vbz_decode(const uint8_t *svb, int64_t nbytes, int32_t n, int16_t *out) {
    const int64_t nkeys = ((int64_t)n + 7) / 8;
    if (nbytes < nkeys) return -1;
    const uint8_t *data = svb + nkeys;
    int64_t p = 0;
    const int64_t avail = nbytes - nkeys;
    uint16_t acc = 0;
    for (int32_t i = 0; i < n; ++i) {
        const int two = (svb[i >> 3] >> (i & 7)) & 1;
        if (p + 1 + two > avail) return -1;
        uint32_t v = data[p];
        if (two) v |= (uint32_t)data[p + 1] << 8;
        p += 1 + two;
        acc = (uint16_t)(acc + (uint16_t)((v >> 1) ^ (0u - (v & 1u))));
        out[i] = (int16_t)acc;
    }
    return p == avail ? 0 : -1;
}
