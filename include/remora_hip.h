/*
 * remora_hip.h — C ABI of libremora_hip.so, the MI355X (gfx950) engine for the per-read
 * modified-base-call hot path of nanoporetech/remora (v3.2.0).
 *
 * Every entry point replaces one reference interface on that path; the reference
 * file:line it stands in for is cited on each declaration (paths relative to the
 * reference checkout).  Plain pointers and sizes only — no torch / numpy types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative rmr_status otherwise; the message
 *     is available (thread-local) from rmr_last_error().  Nothing aborts the process.
 *     The Python host turns non-zero into remora_amd.RemoraError — the one exception type
 *     the reference's callers catch (src/remora/__init__.py:4-7, inference.py:88).
 *   - `mem` says where the caller's data buffers live: RMR_MEM_HOST (numpy / malloc; the
 *     engine stages them through device scratch and copies results back before returning;
 *     rmr_infer_chunks uploads large batches sub-batch by sub-batch through pinned slots on
 *     a second stream, under the kernels of the previous sub-batch)
 *     or RMR_MEM_DEVICE (hipMalloc / torch `data_ptr()`; work is enqueued on the engine
 *     stream and the call returns without synchronising — call rmr_engine_synchronize).
 *   - the caller owns every input and output buffer; the engine owns only weights,
 *     scratch and its stream; no caller pointer is retained past return
 *     (mirrors the Cython memoryview borrow semantics, src/remora/encoded_kmers.pyx:16-18).
 *   - chunk arrays use the CoreRemoraDataset layout (src/remora/data_chunks.py:786-816,
 *     :942-948): signal f32[n,1,L]; sequence i8[n,seq_w]; sequence_to_signal_mapping
 *     i16[n,map_w]; sequence_lengths i16[n]; all C-contiguous.
 *   - an engine (and the models created on it) may be used from several host threads; calls
 *     on one engine are serialised by an internal mutex (reference runs the model from a
 *     Thread: src/remora/inference.py:544-550, :973-982).
 */
#ifndef REMORA_HIP_H
#define REMORA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#pragma GCC visibility push(default)

typedef struct rmr_engine rmr_engine;
typedef struct rmr_model rmr_model;

enum rmr_status {
    RMR_OK = 0,
    RMR_ERR_INVALID = -1,     /* bad argument / unsupported configuration */
    RMR_ERR_HIP = -2,         /* HIP runtime error (message carries hipGetErrorString) */
    RMR_ERR_NOMEM = -3,
    RMR_ERR_DISCORDANT_SEQ = -4, /* "Move table discordant with basecalls" io.py:403-404 */
    RMR_ERR_DISCORDANT_SIG = -5  /* "Move table discordant with signal"    io.py:405-406 */
};

enum rmr_mem { RMR_MEM_HOST = 0, RMR_MEM_DEVICE = 1 };
enum rmr_arch {
    RMR_ARCH_CONV_LSTM = 0, /* models/ConvLSTM_w_ref.py */
    RMR_ARCH_CONV_ONLY = 1  /* models/Conv_w_ref.py     */
};

const char *rmr_last_error(void);
const char *rmr_version(void); /* "remora_hip <major>.<minor> (gfx950)"; the minor number changes with every struct or entry-point change */

/* ---- engine ------------------------------------------------------------------------ */

/* One engine per process per GPU.  With RMR_ENGINE_USE_STREAM in `flags` the engine enqueues
 * on the caller's hipStream_t `stream` (NULL = the legacy default stream; e.g. torch's current
 * stream, so that ordering with the caller's tensors is implicit); otherwise `stream` is
 * ignored and the engine creates its own non-blocking stream.
 * replaces: `tensor.to(device)` single-device plumbing, src/remora/inference.py:311-314,
 *           src/remora/util.py:81-92 (parse_device). */
enum rmr_engine_flags { RMR_ENGINE_OWN_STREAM = 0, RMR_ENGINE_USE_STREAM = 1 };
int rmr_engine_create(int device, void *stream, int flags, rmr_engine **out);
void rmr_engine_destroy(rmr_engine *e);
int rmr_engine_synchronize(rmr_engine *e);
/* Stream-ordered hand-over between two engines of one device: everything queued on `producer` so far is finished before
 * anything queued on `waiter` from now on starts - an event, no host wait.  (A caller that extracts chunks on one engine
 * and runs the network on another - the reads pipeline - needs no hipStreamSynchronize in between.)
 * replaces: nothing in the reference (one stream, one thread: src/remora/inference.py:277-316). */
int rmr_engine_wait_for(rmr_engine *waiter, rmr_engine *producer);
/* chunks per internal sub-batch of the fused pipeline (scratch = ~33 KB/chunk); 0 = default */
int rmr_engine_set_subbatch(rmr_engine *e, int64_t chunks);

/* ---- model (L4: model_util.load_model hands the weights over) ------------------------ */

typedef struct {
    int32_t arch;       /* rmr_arch */
    int32_t size;       /* model_params["size"]: any int >= 1 (src/remora/parsers.py:858-862; constants.py:1 default 64).  The
                           kernels run at 16 / 32 / 64 channels (weight slices resident in registers) or, above 64, at the next
                           multiple of 16 up to 256 (weights streamed from L2, fp32 only); a size in between is run at the next
                           of those with zero-weight channels added (rmr_model_padded_size) - same logits */
    int32_t kmer_len;   /* kmer_context_bases[0] + [1] + 1 */
    int32_t num_out;    /* len(mod_bases) + 1, <= 16 */
    int32_t chunk_len;  /* sum(chunk_context) */
    int32_t dtype;      /* 0 = fp32 MFMA (exact fp32); bf16 MFMA with fp32 accumulate and split operands:
                           1 = bf16, 2 = bf16x3 (2-part split, ~2^-16), 3 = bf16x6 (3-part split, fp32 class);
                           4 = f16: IEEE-half operands, fp32 accumulate, on the fused kernels (conv_lstm, size 64,
                           k-mer length 9 or 6; rmr_infer_chunks only) - the 16-bit pipeline with 10 mantissa bits;
                           5 = f16x3: operands split into two IEEE-half parts, three products (hi hi, hi lo, lo hi) on the
                           half MFMA, fp32 accumulate - 22 significand bits (fp32 class) at bf16x3's cost, in IEEE half's
                           RANGE: an input or folded weight beyond +-65504 becomes infinity (NaN logits), values below
                           2^-3 keep fewer than 22 bits (the low part is a half subnormal); bf16x6 has fp32's exponent range */
} rmr_model_desc;

/* `weights`: host fp32 blob, the torch state_dict tensors flattened in forward order —
 *   for each conv layer: conv.weight[oc][ic][k], conv.bias, bn.weight, bn.bias,
 *   bn.running_mean, bn.running_var; conv layers in the order
 *     conv_lstm: sig_conv1..3, seq_conv1..2, merge_conv1
 *     conv_only: sig_conv1..3, seq_conv1..3, merge_conv1..4
 *   then (conv_lstm) lstm1 {weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0}, lstm2 {same};
 *   then fc.weight, fc.bias.
 * The engine folds BatchNorm (eval, eps 1e-5) into the convolutions and repacks everything
 * into MFMA fragment order on the device.
 * replaces: torch.jit.load + ScriptModule.state_dict(), src/remora/model_util.py:468-481,
 *           :532-563; layer sets as in :231-263. */
int rmr_model_create(rmr_engine *e, const rmr_model_desc *desc, const float *weights,
                     size_t n_floats, rmr_model **out);
void rmr_model_destroy(rmr_model *m);
/* number of floats rmr_model_create expects for `desc` (0 if desc is unsupported) */
size_t rmr_model_weight_count(const rmr_model_desc *desc);
/* The channel count the kernels run a network of desc->size channels at (0 if desc is unsupported), and the network's blob
 * at that size: `weights` (rmr_model_weight_count(desc) floats) with zero-weight channels added -> `out` (out == NULL: only
 * *out_n and *padded_desc are set).  rmr_model_create applies this itself; the entry exists so that a host can see - and a
 * test can check against the reference's forward - exactly which network the kernels evaluate.
 * replaces: nothing (models/ConvLSTM_w_ref.py:11-37 builds layers of any `size`; torch needs no padding). */
int rmr_model_padded_size(const rmr_model_desc *desc);
int rmr_model_pad_weights(const rmr_model_desc *desc, const float *weights, size_t n_floats, rmr_model_desc *padded_desc,
                          float *out, size_t out_cap, size_t *out_n);

/* ---- E1: k-mer one-hot encode with move-table expansion ------------------------------ */
/* replaces: encoded_kmers.compute_encoded_kmer_batch, src/remora/encoded_kmers.pyx:13-45.
 * out: f32[n, 4*(kb+ka+1), sig_len].  The reference takes sig_len from chunk 0
 * (:23, maps[0, lens[0]]); the host wrapper does the same and passes it in. */
int rmr_encode_kmers(rmr_engine *e, int kb, int ka, const int8_t *seqs, int seq_w,
                     const int16_t *maps, int map_w, const int16_t *lens, int64_t n,
                     int sig_len, float *out, int mem);

/* ---- T1: trim stored chunk context to the model's (in place) -------------------------- */
/* replaces: data_chunks_core.trim_sb_chunk_context_core, src/remora/data_chunks_core.pyx:10-45.
 * As in the reference the caller has already subtracted (stored_before - cc_before) from
 * `maps` (src/remora/data_chunks.py:1555-1563). */
int rmr_trim_chunk_context(rmr_engine *e, int stored_before, int stored_after, int cc_before,
                           int cc_after, int total_seq_context, int8_t *seqs, int seq_w,
                           int16_t *maps, int map_w, int16_t *lens, int64_t n, int mem);

/* ---- M1: move-table expansion ---------------------------------------------------------- */
/* replaces: io.parse_move_tag, src/remora/io.py:394-407.  mv_tag = [stride, m0, m1, ...]
 * (int8, host or device).  q2s capacity must be >= mv_tag_len.  *n_out = #moves + 1.
 * seq_len < 0 means None.  Errors: RMR_ERR_DISCORDANT_SEQ / _SIG when `check`. */
int rmr_parse_moves(rmr_engine *e, const int8_t *mv_tag, int64_t mv_tag_len, int64_t sig_len,
                    int64_t seq_len, int check, int reverse_signal, int64_t *q2s,
                    int64_t *n_out, int mem);
/* The same for a batch of reads in one launch (one block per move table): `mv_tags` is the concatenation of
 * the tables, table i at [mv_off[i], mv_off[i+1]); its coordinates are written to q2s[mv_off[i] ...] (a table of
 * m entries yields at most m), counts[i] of them; status[i] = 0 or the code rmr_parse_moves would return for
 * that read (RMR_ERR_DISCORDANT_SEQ / _SIG when `check`, RMR_ERR_INVALID for an empty table or stride <= 0).
 * The call itself fails only on argument errors.  Used by the POD5+BAM ingest, which parses the move tables of
 * a whole batch of alignments at once instead of one kernel launch per read. */
int rmr_parse_moves_batch(rmr_engine *e, const int8_t *mv_tags, const int64_t *mv_off,
                          const int64_t *sig_len, const int64_t *seq_len, int64_t n_reads, int check,
                          int reverse_signal, int64_t *q2s, int64_t *counts, int32_t *status, int mem);
/* Counting half of the median / MAD scaling of reads without sm / sd tags (io.Read.compute_pa_to_norm_scaling,
 * src/remora/io.py:1851-1856: np.median of the pA signal and of its absolute deviations - order statistics of a function of the
 * int16 samples, so counts are all the GPU has to deliver).  `signal`: device-resident decoded samples; span i = signal[start[i]
 * .. start[i] + len[i]) (start, len: host arrays).  Called twice, like the motif scan: with hist == NULL it writes the smallest
 * and largest sample of every span to lo / hi (host int32[n]; an empty span gets lo > hi); with hist != NULL the caller passes
 * those lo / hi back, hist_off (host int64[n + 1], the running sum of hi - lo + 1, 0 for empty spans) and receives
 * hist[hist_off[i] + (v - lo[i])] = the number of samples of span i equal to v (host uint32[hist_off[n]]).  Synchronous. */
int rmr_signal_histograms(rmr_engine *e, const int16_t *signal, const int64_t *start, const int64_t *len, int64_t n, int32_t *lo,
                          int32_t *hi, const int64_t *hist_off, uint32_t *hist);

/* replaces: for a whole batch, the tail of io.Read.add_alignment and Read.into_remora_read (src/remora/io.py:2003-2012:
 * the signal of an alignment is dacs[sp:][ts:ns]; :2123-2177: the read keeps dacs[q2s[0]:q2s[-1]] and the mapping q2s -
 * q2s[0]) on arrays that are already resident: `signal` (device) holds the decoded samples of the batch back to back,
 * read i's trimmed signal starts at signal[src_start[i]]; its move-table coordinates are the seq_len[i] + 1 values at
 * q2s[q2s_off[i] ...] (device, as rmr_parse_moves_batch wrote them).  Written (device): dacs, s2s [sum(seq_len) + n_reads],
 * d_sig_off / d_seq_off [n_reads + 1] - the rmr_reads layout; sig_off (host) receives the sample offsets.  src_start,
 * q2s_off, seq_len are host arrays.  Lets the POD5 + BAM ingest hand batches to the extraction without per-read host work. */
int rmr_assemble_reads(rmr_engine *e, int64_t n_reads, const int16_t *signal, const int64_t *src_start, const int64_t *q2s,
                       const int64_t *q2s_off, const int64_t *seq_len, int16_t *dacs, int64_t dacs_cap, int64_t *s2s,
                       int64_t *d_sig_off, int64_t *d_seq_off, int64_t *sig_off);

/* ---- N1: BAM records for the POD5+BAM ingest (host code: BGZF inflate + record / tag decode) ---------- */
/* replaces: what ReadIndexedBam / pysam hand to io.Read.add_alignment (src/remora/io.py:184-358, :1972-2084):
 * per alignment the flag, reference id / start, mapping quality, name, CIGAR, sequence, the tags mv, ts, ns, sp,
 * sm, sd, pi, MD, the record bytes as stored (for writing the record back with MM/ML tags), and - when
 * `want_ref` - the reference bases spanned by the alignment rebuilt from query + CIGAR + MD
 * (pysam.AlignedSegment.get_reference_sequence: mismatches lower case; ref_ok = 0 without MD / when MD and
 * CIGAR disagree).  Records arrive `max_records` at a time as flat arrays with [n+1] offset tables; every
 * pointer of the batch stays valid until the next rmr_bam_read_batch / rmr_bam_close on the same handle.
 * `has` bit i set = tag present: 0 mv (B:c), 1 ts, 2 ns, 3 sp, 4 sm, 5 sd, 6 pi, 7 MD.  `mv` excludes nothing:
 * [stride, m0, m1, ...] as stored.  n_records < max_records means end of file.
 * want_ref: bit 0 = rebuild the reference bases; bit 1 = identifiers only (flags, positions, names, the scalar tags, pi and
 * `has`; record bytes, bases, CIGAR, move table, MD and reference stay empty - their offset tables are all zero): what a
 * pass that only counts records against the signal file needs (get_read_ids, src/remora/io.py:362-391). */
typedef struct rmr_bam rmr_bam;
typedef struct rmr_bam_batch {
    int64_t n_records;
    const int32_t *flag, *ref_id, *pos, *mapq, *l_seq, *n_cigar;
    const int64_t *raw_off;   const uint8_t *raw;      /* record bytes (without the leading block_size) */
    const int64_t *name_off;  const char *names;
    const int64_t *seq_off;   const char *seq;         /* ASCII bases */
    const int64_t *cigar_off; const uint32_t *cigar;   /* BAM CIGAR words: len << 4 | op */
    const int64_t *tags_off;                           /* [n] offset of the tag region inside the record */
    const uint8_t *has;
    const int64_t *mv_off;    const int8_t *mv;
    const int32_t *ts, *ns, *sp;
    const float *sm, *sd;
    const int64_t *pi_off;    const char *pi;
    const int64_t *md_off;    const char *md;
    const uint8_t *ref_ok;
    const int64_t *refseq_off; const char *refseq;
    const int64_t *voffset;                            /* [n] BGZF virtual offset of the record (for rmr_bam_seek) */
} rmr_bam_batch;
int rmr_bam_open(const char *path, rmr_bam **out);
/* the same with the number of BGZF inflate workers of this handle given by the caller (>= 1; 0 = rmr_bam_open's default:
 * RMR_BAM_INFLATE_THREADS, else 8) - a caller that sizes its thread pools per rank passes the count instead of setting
 * an environment variable in a process whose native threads read the environment (pysam's `threads=` of
 * AlignmentFile, src/remora/io.py:236) */
int rmr_bam_open_threads(const char *path, int inflate_threads, rmr_bam **out);
void rmr_bam_close(rmr_bam *b);
/* everything before the first record (magic, header text, reference dictionary), uncompressed */
int rmr_bam_header(rmr_bam *b, const uint8_t **bytes, int64_t *n_bytes, int64_t *n_refs);
const char *rmr_bam_ref_name(rmr_bam *b, int64_t ref_id); /* NULL when out of range */
int rmr_bam_read_batch(rmr_bam *b, int64_t max_records, int want_ref, rmr_bam_batch *out);
/* continue reading at a record's virtual offset (compressed block offset << 16 | offset inside the inflated block),
 * as returned in rmr_bam_batch.voffset - the random access ReadIndexedBam.get_alignments needs
 * (src/remora/io.py:303-325) */
int rmr_bam_seek(rmr_bam *b, int64_t voffset);
/* From the current position to the end of the file: count the records and note the virtual offset of records 0, every,
 * 2*every, ... in voffsets[0..cap) (only block_size fields are read).  What a worker of a multi-GPU run needs to take a
 * contiguous share of the alignments by itself (one process per GPU, each seeks to its own range; north_star: "reads
 * shard embarrassingly across the GPUs") - the reference hands reads out from ONE reader process instead
 * (src/remora/inference.py:488-519).  Leaves the handle at end of file: rmr_bam_seek before reading. */
int rmr_bam_scan(rmr_bam *b, int64_t every, int64_t *voffsets, int64_t cap, int64_t *n_records);
/* The virtual offset of the first record that starts in a BGZF member at or behind byte `file_offset` of the file, found
 * WITHOUT the records in front of it (*voffset = -1: none); the handle is left there.  A position counts as a record start
 * when the record and the records chained behind it pass every structural check of the format (field ranges against
 * the header, printable name, CIGAR codes, a tag region that parses to block_size exactly).  With it N workers split a
 * file by byte ranges in O(1) instead of one O(file) pass; the worker in front verifies each guess for certain (its
 * own chain of records has to end on it).  Same reference counterpart as rmr_bam_scan. */
int rmr_bam_guess_start(rmr_bam *b, int64_t file_offset, int64_t *voffset);

/* ---- N3: the output side of `remora infer` for a batch of reads (host code) ----------------------------- */
/* replaces: util.format_mm_ml_tags (src/remora/util.py:485-537) as inference.post_process_reads calls it per read
 * (src/remora/inference.py:429-459; flagged slow in the source, :55).  Read r has calls call_off[r] .. call_off[r+1]:
 * pos[] = positions in its sequence seq[seq_off[r] .. seq_off[r+1]) (any order: sorted stably here), probs[call][n_mods]
 * = probabilities of the modified bases.  mod_codes = n_mods NUL-terminated short names one after the other ("m\0h\0",
 * ChEBI codes allowed).  Per read and modified base: "<can_base><strand><code>?,<gaps>;" appended to mm (gaps = canonical
 * bases skipped between consecutive calls) and floor(p * 256) clipped to 255 appended to ml (all calls of base 0, then of
 * base 1, ...); mm_off / ml_off [n_reads + 1] delimit the reads' slices; a read without calls gets empty slices. */
int rmr_format_mm_ml(int64_t n_reads, const char *seq, const int64_t *seq_off, const int64_t *pos, const double *probs,
                     const int64_t *call_off, int n_mods, const char *mod_codes, char can_base, char strand, char *mm,
                     int64_t mm_cap, int64_t *mm_off, uint8_t *ml, int64_t ml_cap, int64_t *ml_off);
/* replaces: pysam.AlignedSegment.from_dict(io_read.full_align) + the MM/ML tags, src/remora/inference.py:450, :619-623.
 * raw[r] = record r as stored (without block_size, raw_len[r] bytes), tags_off[r] = offset of its tag region.  Written to
 * `out`, one after the other: int32 block_size + the record with any MM/ML/Mm/Ml tag removed and, when has_tags[r],
 * MM:Z:<mm slice> and ML:B:C<ml slice> appended (has_tags[r] = 0: the record leaves without modified-base tags). */
int rmr_records_with_mod_tags(int64_t n_reads, const uint8_t *const *raw, const int64_t *raw_len, const int64_t *tags_off,
                              const char *mm, const int64_t *mm_off, const uint8_t *ml, const int64_t *ml_off,
                              const uint8_t *has_tags, uint8_t *out, int64_t out_cap, int64_t *out_len);
/* The same for reference-anchored calling (`infer --reference-anchored`): a record that gets tags and owns a non-empty slice
 * ref_seq[ref_off[r] .. ref_off[r+1]) - the reference bases of its alignment, forward strand - is written in the form the
 * reference writes there: CIGAR <len>M, that sequence, qualities 0xff (src/remora/inference.py:452-458); every other record
 * as rmr_records_with_mod_tags writes it.  The caller's `out` needs 4 + (len + 1) / 2 + len more bytes per such record. */
int rmr_records_with_mod_tags_ref(int64_t n_reads, const uint8_t *const *raw, const int64_t *raw_len, const int64_t *tags_off,
                                  const char *mm, const int64_t *mm_off, const uint8_t *ml, const int64_t *ml_off,
                                  const uint8_t *has_tags, const char *ref_seq, const int64_t *ref_off, uint8_t *out, int64_t out_cap,
                                  int64_t *out_len);

/* The signal coordinate of every reference position of one alignment (host code).
 * replaces: compute_ref_to_signal = map_ref_to_signal(make_sequence_coordinate_mapping(cigar)), src/remora/data_chunks.py:60-122
 * (called from io.Read.add_alignment, src/remora/io.py:2070-2080), with np.interp's float64 arithmetic: the same integers.
 * cigar u32[n_ops] as stored in a BAM record (length << 4 | op), taken back to front when `reverse` (a reverse-strand
 * record: the read-oriented CIGAR); query_to_signal i64[n_knots] (the expanded move table, seq_len + 1 entries).
 * ref_to_signal receives *n_out = reference length + 1 entries (RMR_ERR_INVALID and *n_out set when cap is smaller).
 * Errors (RMR_ERR_INVALID, the reference's texts): "Invalid cigar op(s)", "No match operations found in alignment cigar". */
int rmr_ref_to_signal(const uint32_t *cigar, int64_t n_ops, int reverse, const int64_t *query_to_signal, int64_t n_knots,
                      int64_t *ref_to_signal, int64_t cap, int64_t *n_out);
/* The reference-anchored half of a BAM batch's ingest, per record on native threads: io.parse_move_tag
 * (src/remora/io.py:394-407: mv -> query_to_signal, with its two checks) followed by compute_ref_to_signal
 * (src/remora/data_chunks.py:118-122) as Read.add_alignment composes them (io.py:2066-2084).  Record i: move table
 * mv[mv_off[i] .. mv_off[i+1]) (first entry = stride), trimmed signal length sig_len[i], seq_len[i] bases, CIGAR words
 * cigar[cigar_off[i] .. cigar_off[i+1]) in BAM order (reverse[i]: walked backwards, the read's orientation), ref_len[i]
 * reference bases (< 0: no reference sequence - status 9 once its move table has passed the checks).  Its ref_to_signal goes
 * to r2s[r2s_off[i] .. +ref_len[i]+1).
 * status[i]: 0 done; RMR_ERR_INVALID (empty table / stride <= 0), RMR_ERR_DISCORDANT_SEQ, RMR_ERR_DISCORDANT_SIG as
 * rmr_parse_moves_batch; 1 "Discordant ref seq lengths" (io.py:2078-2079); 2 "Invalid cigar op(s)"; 3 "No match operations
 * found in alignment cigar"; 4 a CIGAR with an empty match run; 8 no move table; 9 no reference sequence. */
int rmr_ref_anchor_batch(int64_t n, const int8_t *mv, const int64_t *mv_off, const int64_t *sig_len, const int64_t *seq_len,
                         const uint32_t *cigar, const int64_t *cigar_off, const uint8_t *reverse, const int64_t *ref_len,
                         int64_t *r2s, const int64_t *r2s_off, int32_t *status, int threads);

/* The native BAM reader's own inflater for BGZF members, alone (host code; replaces zlib's inflate under
 * rmr_bam_read_batch, which itself stands in for htslib below pysam, src/remora/io.py:184-358): a raw RFC 1951 stream
 * src[0..n) into exactly out[0..out_len); RMR_ERR_INVALID when the stream is malformed, ends elsewhere or does not fill
 * the output exactly.  Inside the reader every member's CRC32 is checked and zlib takes what this decoder refuses. */
int rmr_inflate_raw(const uint8_t *src, int64_t n, uint8_t *out, int64_t out_len);

/* ---- N3: BGZF members for the output BAM (host code) ------------------------------------------------ */
/* replaces: the deflate step of pysam / htslib below AlignmentFile.write (src/remora/inference.py:619-623), for the
 * writer's fast mode (`--bam-level 1`).  src[0..n) is cut into payloads of 0xFF00 bytes (the last one shorter); each
 * becomes one BGZF member (gzip + BC extra field, SAM spec 4.1) whose deflate stream is a single dynamic-Huffman block
 * over the literals - no LZ77 matches - or a stored block when that would not shrink it; CRC32 and ISIZE as gzip wants
 * them.  Members are written to `out` back to back (out_cap >= 65311 bytes per payload), *out_len = their total size;
 * `n_threads` payloads are coded side by side.  No end-of-file marker is appended. */
int rmr_bgzf_huffman(const uint8_t *src, int64_t n, int n_threads, uint8_t *out, int64_t out_cap, int64_t *out_len);

/* ---- N1: POD5 signal rows, the zstd layer (host code, parallel over rows) ----------------------------- */
/* replaces: the zstd step of pod5's signal reader under io.iter_signal (src/remora/io.py:441-474).  `src[i]`
 * points at row i's compressed bytes (one zstd frame, src_len[i] bytes).  rmr_zstd_frame_sizes reads the
 * decompressed size of every frame from its header; rmr_zstd_rows inflates row i into out[out_off[i] ..
 * out_off[i+1]) (that span must equal the frame's content size) with `n_threads` threads.  libzstd.so.1 is
 * loaded from the system at run time. */
int rmr_zstd_frame_sizes(const uint8_t *const *src, const int64_t *src_len, int64_t n_rows, int64_t *sizes);
int rmr_zstd_rows(const uint8_t *const *src, const int64_t *src_len, int64_t n_rows, uint8_t *out,
                  const int64_t *out_off, int n_threads);

/* ---- N1: POD5 signal decompression (the VBZ layer below zstd) ------------------------------ */
/* replaces: the per-row signal decode that pod5's C++ reader performs for the records consumed by
 * io.iter_signal (src/remora/io.py:441-474) / Read.from_pod5_and_alignment (:2086-2121):
 * streamvbyte16 (ceil(n/8) key bytes, one bit per sample LSB first: 0 = one data byte, 1 = two; then the
 * data bytes) -> zigzag -> running sum in int16.  `svb` holds the zstd-DEcompressed bytes of the rows back to
 * back (row r = svb[row_off[r] : row_off[r+1]]), row_samples[r] its sample count; `out` receives the rows'
 * samples back to back (sum(row_samples) int16).  The delta coding restarts in every row.
 * Errors: RMR_ERR_INVALID "corrupt VBZ signal block" when a row's byte count disagrees with its keys. */
int rmr_vbz_decode(rmr_engine *e, const uint8_t *svb, const int64_t *row_off, const int32_t *row_samples,
                   int64_t n_rows, int16_t *out, int mem);

/* ---- M3: motif scan ---------------------------------------------------------------------- */
/* replaces: Motif.findall (src/remora/util.py:281-297) + find_focus_bases_in_int_sequence (:413-426) as
 * called by RemoraRead.set_motif_focus_bases (src/remora/data_chunks.py:310-317), for a batch of reads:
 * flags[b] = 1 iff base b (index into the concatenated int_seq) is the focus base of a hit of any motif
 * that lies entirely inside its read.  mask[m][j] has bit k set when base k (0..3 = ACGT) is allowed at
 * motif position j; N bases (-1) never match.  The positions come out in ascending order once the
 * flags are compacted (the reference's python-set order is a property of the single-read host API). */
typedef struct {
    int32_t n_motifs;            /* <= 8 */
    int32_t len[8];              /* <= 16 */
    int32_t focus_pos[8];        /* may be negative after leading-N stripping (util.py:208-214) */
    uint8_t mask[8][16];
} rmr_motif_set;
int rmr_motif_flags(rmr_engine *e, const int8_t *int_seq, const int64_t *seq_off, int64_t n_reads,
                    const rmr_motif_set *motifs, uint8_t *flags, int mem);
/* The same scan for a batch of device-resident reads without the flag array (DEVICE pointers only): pass 1 counts the
 * hits of every read (counts i64[n_reads]); the caller turns them into offsets foc_off i64[n_reads+1] (exclusive
 * prefix sum) and sizes `focus`; pass 2 writes the read-local focus positions of read r, ascending, at
 * focus[foc_off[r] .. foc_off[r+1]).  One wavefront per read, ballot + prefix-popcount compaction. */
int rmr_motif_focus_counts(rmr_engine *e, const int8_t *int_seq, const int64_t *seq_off, int64_t n_reads,
                           const rmr_motif_set *motifs, int64_t *counts);
int rmr_motif_focus_fill(rmr_engine *e, const int8_t *int_seq, const int64_t *seq_off, int64_t n_reads,
                         const rmr_motif_set *motifs, const int64_t *foc_off, int64_t *focus);

/* Host-side gather of a batch of reads into the concatenated rmr_reads layout (no GPU involved; native threads): read i
 * contributes sig_n[i] int16 dacs, seq_n[i]+1 int64 mapping entries and seq_n[i] bases (integers of seq_itemsize[i]
 * bytes, narrowed to int8).  dst_* are caller buffers (typically the pinned staging buffer that is uploaded next);
 * sig_off / seq_off i64[n_reads+1] receive the offsets; the mapping of read i starts at dst_maps[seq_off[i] + i].
 * replaces: nothing in the reference (it handles one read at a time, src/remora/inference.py:62-137). */
int rmr_pack_reads(int64_t n_reads, const void *const *dacs, const int64_t *sig_n, const void *const *maps,
                   const void *const *seqs, const int64_t *seq_n, const int32_t *seq_itemsize, int16_t *dst_dacs,
                   int64_t *dst_maps, int8_t *dst_seq, int64_t *sig_off, int64_t *seq_off, int threads);
/* The same gather with the mapping narrowed to int32 (dst_maps32, same offsets) - a seventh less to carry across PCIe; the
 * caller widens it on the device.  *maps_fit = 0 when some mapping value does not fit int32 (the buffer's mapping is then
 * not usable: gather again with rmr_pack_reads). */
int rmr_pack_reads_narrow(int64_t n_reads, const void *const *dacs, const int64_t *sig_n, const void *const *maps,
                          const void *const *seqs, const int64_t *seq_n, const int32_t *seq_itemsize, int16_t *dst_dacs,
                          int32_t *dst_maps32, int8_t *dst_seq, int64_t *sig_off, int64_t *seq_off, int threads, int *maps_fit);

/* The bases of selected records of a BAM batch in read orientation, back to back, with their integer codes (host code, native
 * threads).  Record i: src[start[i] .. start[i] + len[i]); `upper` != 0 folds ASCII lower case first; rev[i] != 0: reversed and
 * mapped through comp[256]; codes[j] = code[256] of the oriented byte; fwd (NULL allowed): the folded bases un-reversed.  Output
 * offset of record i = len[0] + ... + len[i-1] in all three.
 * replaces: per read, revcomp(query_sequence) / revcomp(ref_seq) in io.Read.add_alignment (src/remora/io.py:2023, :2058-2060)
 * and util.seq_to_int (src/remora/util.py:131-142). */
int rmr_orient_bases(const uint8_t *src, const int64_t *start, const int64_t *len, const uint8_t *rev, int64_t n, int upper,
                     const uint8_t *comp, const int8_t *code, uint8_t *fwd, uint8_t *oriented, int8_t *codes, int threads);


/* ---- X1 + X2 + X3 (+X6): chunk extraction for a batch of reads --------------------------- */
/* replaces: RemoraRead.sig (src/remora/data_chunks.py:191-197), iter_chunks (:425-466),
 * extract_chunk (:331-423) and the row packing of CoreRemoraDataset.write_chunk
 * (:1376-1418).  Reads are concatenated; read r owns dacs[sig_off[r]:sig_off[r+1]],
 * int_seq[seq_off[r]:seq_off[r+1]], seq_to_sig_map[seq_off[r]+r : seq_off[r+1]+r+1]
 * (one more entry than bases) and focus_bases[focus_off[r]:focus_off[r+1]] (read-local
 * base indices, in the order the caller wants the chunks).
 *
 * Step 1 normalises the signal and computes the geometry of every chunk:
 *   sig_out      f32[sig_off[n_reads]]            ((dacs - shift)/scale in f64, cast to f32)
 *   geo          i64[n_chunks, 6] = {seq_len, chunk_sig_focus_idx, chunk_focus_base,
 *                                    read_focus_base, seq_start, sig_start(signed, unclipped)}
 *   *max_seq_len max over chunks of seq_len (host int, always written; implies a sync)
 * Step 2 fills the dataset-layout arrays for widths seq_w >= max_seq_len+kb+ka,
 *   map_w >= max_seq_len+1 (padding columns: sequence -1, mapping 0; the reference leaves
 *   them uninitialised). */
typedef struct {
    int64_t n_reads;
    const int16_t *dacs;        /* concatenated */
    const int64_t *sig_off;     /* [n_reads+1] */
    const int64_t *seq_to_sig;  /* concatenated, n_bases+1 per read */
    const int8_t *int_seq;      /* concatenated, values -1..3 */
    const int64_t *seq_off;     /* [n_reads+1] */
    const double *shift;        /* [n_reads] */
    const double *scale;        /* [n_reads] */
    const int64_t *focus_bases; /* concatenated, read-local */
    const int64_t *focus_off;   /* [n_reads+1] */
    int32_t cc_before, cc_after, kb, ka, base_start_justify, offset;
    /* base_start_justify == 2: focus_bases holds SIGNAL indices (the focus_sig_idx argument of RemoraRead.extract_chunk,
     * src/remora/data_chunks.py:331-343) and `offset` is the read_focus_base the chunks report - no clipping, no mapping
     * look-up */
    /* optional (all three or NULL): HOST copies of sig_off / seq_off / focus_off for callers whose arrays are device
     * memory - the library needs the offsets on the host (launch sizes, staging) and otherwise fetches them with three
     * small device-to-host copies per call (about 60 us of a single-read call) */
    const int64_t *host_sig_off, *host_seq_off, *host_focus_off;
} rmr_reads;

int rmr_chunk_geometry(rmr_engine *e, const rmr_reads *reads, float *sig_out, int64_t *geo,
                       int64_t *max_seq_len, int mem);
int rmr_chunk_fill(rmr_engine *e, const rmr_reads *reads, const float *sig, const int64_t *geo,
                   float *signal /* f32[n_chunks, L] */, int8_t *seqs, int seq_w, int16_t *maps,
                   int map_w, int16_t *lens, int64_t *read_focus_bases, int mem);

/* ---- F1 / F2: network forward on materialised inputs ------------------------------------ */
/* replaces: `model(sigs, enc_kmers)` = ConvLSTM_w_ref.network.forward
 * (models/ConvLSTM_w_ref.py:39-58) / Conv_w_ref.network.forward (models/Conv_w_ref.py:44-62)
 * as called from RemoraRead.run_model (src/remora/data_chunks.py:528-533) and
 * run_model_batched (src/remora/inference.py:311-315).  sigs f32[n,1,L]; seqs f32[n,4K,L]
 * (any values, not only one-hot); logits f32[n,num_out]. */
int rmr_forward(rmr_model *m, const float *sigs, const float *seqs, int64_t n, float *logits,
                int mem);

/* ---- fused hot path: chunk arrays -> logits, never materialising the one-hot ------------- */
/* replaces: CoreRemoraDataset.extract_batch -> compute_encoded_kmer_batch
 * (src/remora/data_chunks.py:1652-1676) followed by model(sigs, enc_kmers) — i.e. what
 * RemoraRead.prepare_batches + run_model (:468-540) and prep_nn_input + run_model_batched
 * (src/remora/inference.py:152-168, :277-316) compute between them.
 * label_counts (nullable): i64[num_out], INCREMENTED by the argmax histogram of this call
 * (first maximum wins, as np.argmax in src/remora/validate.py:42-45). */
int rmr_infer_chunks(rmr_model *m, const float *signal, const int8_t *seqs, int seq_w,
                     const int16_t *maps, int map_w, const int16_t *lens, int kb, int ka,
                     int64_t n, float *logits, int64_t *label_counts, int mem);

/* ---- one read, one call ------------------------------------------------------------------- */
/* replaces: what `call_read_mods` does for ONE read between its motif search and its return
 * (src/remora/inference.py:661-712): RemoraRead.prepare_batches -> an in-memory CoreRemoraDataset of the read's chunks
 * (src/remora/data_chunks.py:468-514: sig :191-197, iter_chunks :425-466, extract_chunk :331-423, write_chunk
 * :1376-1418, compute_encoded_kmer_batch) and RemoraRead.run_model (:516-540) - staging, signal normalisation, chunk
 * geometry and rows, and the network in one call with ONE stream synchronisation (the chunk geometry is integer arithmetic on
 * the mapping and is computed on the host while the arrays cross PCIe), no Python in between.
 * All pointers are HOST memory.  focus_bases: read-local base indices in the order the caller wants the chunks (the
 * reference's python-set order is the caller's business); logits f32[n_focus, num_out] and read_focus_bases
 * i64[n_focus] (the focus base after `offset`, clipped into the read, :443-446) are written in that order. */
typedef struct {
    const int16_t *dacs;        /* [n_sig] */
    int64_t n_sig;
    const int64_t *seq_to_sig;  /* [n_bases + 1] */
    const void *int_seq;        /* [n_bases] integers of seq_itemsize bytes (1, 2, 4 or 8), values -1..3 */
    int32_t seq_itemsize, _pad;
    int64_t n_bases;
    double shift, scale;
    const int64_t *focus_bases; /* [n_focus] */
    int64_t n_focus;
    int32_t cc_before, cc_after, kb, ka, base_start_justify, offset;
} rmr_read;
int rmr_call_read(rmr_model *m, const rmr_read *read, float *logits, int64_t *read_focus_bases);

/* argmax histogram of existing logits (same rule), counts[num_out] incremented. */
int rmr_count_labels(rmr_engine *e, const float *logits, int64_t n, int num_out,
                     int64_t *counts, int mem);

/* Validation tally of one batch of logits ON THE DEVICE (every pointer except label_of_column is device memory): what
 * ValidationLogger.run_validation computes on the host per batch and at the end (src/remora/validate.py:208-259) -
 * add_unmodeled_labels (:69-99: the model's num_out columns widened to the dataset's num_labels; label_of_column[c] = the
 * model column of label c, -1 = not modelled -> -1000), util.softmax_axis1 in float32 (util.py:182-186), np.argmax, the
 * confusion counts of compute_metrics (:42-45), CrossEntropyLoss.
 *   confusion i64[num_labels * num_labels] (true x called) and loss_sum f64[1] (sum of the chunks' cross entropies) are
 *   INCREMENTED; win_prob f32[n] (the called class's probability: what the filtered metrics take their quantile of) and
 *   call u8[n] are written.  Asynchronous on the engine's stream. */
int rmr_validation_tally(rmr_engine *e, const float *logits, const int64_t *labels, int64_t n, int num_out, int num_labels,
                         const int32_t *label_of_column, int64_t *confusion, float *win_prob, uint8_t *call, double *loss_sum);

/* ---- (e) multi-GPU: the ONE collective of the path, RCCL over xGMI called from the library ------- */
/* replaces, in distributed form: the per-label tally of a run summed over the workers —
 * validate.compute_metrics' confusion/label tally (src/remora/validate.py:42-45) and get_label_counts
 * (src/remora/data_chunks.py:1074-1082); the reference itself is single-process.  One process per GPU; chunks /
 * reads shard embarrassingly, so this int64[num_out] all-reduce(sum) is the only exchange of a job.
 * Bootstrap as RCCL's own: rank 0 calls rmr_comm_unique_id and hands the 128 bytes to every rank by ANY channel
 * (file, socket, MPI, torch.distributed ...); every rank then calls rmr_comm_init (collective: all ranks must
 * call it) with the same id.  librccl is dlopen'ed on first use: a single-GPU caller never needs it.
 * rmr_allreduce_counts: in place; world 1 (or no communicator) = identity.  RMR_MEM_DEVICE: enqueued on the
 * engine stream (ordered after the kernels that produced the counts; rmr_engine_synchronize to read them on the
 * host); RMR_MEM_HOST: staged through the engine, synchronous. */
#define RMR_COMM_ID_BYTES 128
int rmr_comm_unique_id(uint8_t id[RMR_COMM_ID_BYTES]);
int rmr_comm_init(rmr_engine *e, const uint8_t id[RMR_COMM_ID_BYTES], int rank, int world);
int rmr_comm_destroy(rmr_engine *e);
int rmr_allreduce_counts(rmr_engine *e, int64_t *counts, int n, int mem);

/* ---- N2: signal-mapping refinement (banded dynamic programming) --------------------------- */
/* replaces, for a batch of reads, the body of refine_signal_mapping
 * (src/remora/refine_signal_map.py:780-840) as called by SigMapRefiner.refine_sig_map (:472-497):
 * extract_levels (src/remora/refine_signal_map_core.pyx:87-101), compute_sig_band (.py:631-683),
 * convert_to_seq_band (.py:740-772), adjust_seq_band (core.pyx:31-74), validate_band (.py:686-737),
 * the signal normalisation (dacs - shift) / scale (f64, cast to f32, .py:479), and seq_banded_dp
 * (core.pyx:403-473) with either forward step (Viterbi core.pyx:256-317, dwell penalty :150-253)
 * and the traceback (:119-148).  Scores are float32 with the reference's operation order, so the
 * returned paths equal the reference's. */
enum rmr_refine_algo { RMR_REFINE_VITERBI = 0, RMR_REFINE_DWELL_PENALTY = 1 };
enum rmr_refine_status {   /* per read; > 0 are the RemoraError cases of validate_band */
    RMR_REFINE_OK = 0,
    RMR_REFINE_BAND_START = 1,   /* "Band does not start with 0 coordinate." */
    RMR_REFINE_ZERO_LEN = 2,     /* "Band contains 0-length region" */
    RMR_REFINE_START_ORDER = 3,  /* "Band start positions are not monotonically increasing" */
    RMR_REFINE_END_ORDER = 4,    /* "Band end positions are not monotonically increasing" */
    RMR_REFINE_BAND_END = 5,     /* "Invalid seq_band end coordinate" */
    RMR_REFINE_BAND_LENGTH = 6,  /* "Invalid sig_band length" */
    RMR_REFINE_EMPTY = 7         /* read without bases */
};
typedef struct {
    const float *kmer_levels;   /* host, f32[4^kmer_len], index = sum base_j * 4^(kmer_len-1-j); no NaN */
    int32_t kmer_len, center_idx;
    const float *sd_arr;        /* host, short dwell penalties (SigMapRefiner.sd_arr); dwell_penalty only */
    int32_t sd_len;
    int32_t algo;               /* rmr_refine_algo */
    int32_t half_bandwidth;     /* SigMapRefiner.half_bandwidth */
    int32_t min_step;           /* adjust_band_min_step (2 in the reference) */
} rmr_refine_desc;
typedef struct rmr_refiner rmr_refiner;

int rmr_refiner_create(rmr_engine *e, const rmr_refine_desc *desc, rmr_refiner **out);
void rmr_refiner_destroy(rmr_refiner *r);
const char *rmr_refine_status_message(int status);
/* Reads are concatenated exactly as in rmr_reads (dacs / sig_off / seq_to_sig / int_seq / seq_off /
 * shift / scale).  out_map has the layout of seq_to_sig and receives the refined mapping of every
 * read whose status is 0 (other reads are left untouched); status i32[n_reads]. */
int rmr_refine_signal_maps(rmr_refiner *r, int64_t n_reads, const int16_t *dacs, const int64_t *sig_off,
                           const int64_t *seq_to_sig, const int8_t *int_seq, const int64_t *seq_off,
                           const double *shift, const double *scale, int64_t *out_map, int32_t *status,
                           int mem);
/* The two quantile vectors the rough re-scale fits its line through (SigMapRefiner.rough_rescale,
 * src/remora/refine_signal_map.py:330-420: np.quantile of the normalised centre sample of every base and of the
 * expected k-mer levels, clip_bases dropped at both ends of reads longer than 2 * clip_bases).  Reads laid out as
 * above, DEVICE pointers; quants is a HOST array f64[n_quants] in [0, 1].  sig_q / lvl_q: device f64[n_reads][n_quants],
 * bit-identical to numpy's "linear" quantiles of the float64 / float32 arrays.  status i32[n_reads] (device): 0 ok,
 * 1 = read longer than the in-LDS sort holds (16384 kept bases; max_read_bases, the longest read of the batch or 0
 * if unknown, sizes the sort), 2 = read without bases; rows of such reads are not written. */
int rmr_rescale_quantiles(rmr_refiner *r, int64_t n_reads, const int16_t *dacs, const int64_t *sig_off,
                          const int64_t *seq_to_sig, const int8_t *int_seq, const int64_t *seq_off,
                          const double *shift, const double *scale, int64_t max_read_bases, int clip_bases,
                          int n_quants, const double *quants, double *sig_q, double *lvl_q, int32_t *status);

/* ---- measurement: HIP-event timing of every kernel launch on the engine stream ----------- */
int rmr_profile_enable(rmr_engine *e, int on);
int rmr_profile_reset(rmr_engine *e);
int rmr_profile_num_kernels(void);
const char *rmr_profile_kernel_name(int kernel_id);
/* synchronises, then returns accumulated event time and launch count for one kernel id */
int rmr_profile_get(rmr_engine *e, int kernel_id, double *total_ms, int64_t *launches);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* REMORA_HIP_H */
