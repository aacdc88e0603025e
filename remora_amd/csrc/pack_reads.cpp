// pack_reads.cpp — host side of the batched reads path: the arrays of a batch of reads (each read its own numpy arrays)
// gathered into the concatenated rmr_reads layout (include/remora_hip.h) inside the caller's pinned staging buffer, by a
// few native threads.  The reference has no counterpart (it processes reads one at a time in Python,
// src/remora/inference.py:62-137); in this engine the per-read Python copies of this step were 55 of the 83 ms that a
// batch of 2048 x 5 kb reads took end to end.
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "rmr_internal.h"

namespace {

template <typename T>
void narrow_to_i8(const void *src, int64_t n, int8_t *dst) {
    const T *s = static_cast<const T *>(src);
    for (int64_t i = 0; i < n; ++i) dst[i] = (int8_t)s[i];
}

// int64 bases (what util.seq_to_int and the reference hand over: 8 bytes a base) -> int8, sixteen at a time: the low dwords of
// eight 16-byte loads, then the two saturating packs.  Base codes are -1..3; a value outside int8 must not become a valid
// code: the packs clamp whatever fits in 32 bits to -128 / 127, and a value whose HIGH dword is not the sign extension of its
// low dword (|v| >= 2^31, which the low dword alone would wrap - 2^32 + 1 -> 1) is forced to 127 / -128 by its sign as well; the scalar tail
// clamps the same way.  Everything outside -1..3 is refused downstream (RemoraRead.check's message).  The scalar loop cost
// 5 us per 5 kb read - more than the copy of its 100 KB of signal.
inline int8_t clamp_i64_to_i8(int64_t v) { return (int8_t)(v > 127 ? 127 : (v < -128 ? -128 : v)); }
void narrow_i64_to_i8(const void *src, int64_t n, int8_t *dst) {
    const int64_t *s = static_cast<const int64_t *>(src);
    int64_t i = 0;
#if defined(__SSE2__)
    for (; i + 16 <= n; i += 16) {
        __m128i d[4];
        for (int k = 0; k < 4; ++k) {
            const __m128 a = _mm_castsi128_ps(_mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i + 4 * k)));
            const __m128 b = _mm_castsi128_ps(_mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i + 4 * k + 2)));
            const __m128i lo = _mm_castps_si128(_mm_shuffle_ps(a, b, _MM_SHUFFLE(2, 0, 2, 0)));  // low dwords of four int64
            const __m128i hi = _mm_castps_si128(_mm_shuffle_ps(a, b, _MM_SHUFFLE(3, 1, 3, 1)));  // their high dwords
            const __m128i wide = _mm_xor_si128(hi, _mm_srai_epi32(lo, 31));  // non-zero where hi is not lo's sign extension
            const __m128i fits = _mm_cmpeq_epi32(wide, _mm_setzero_si128());
            const __m128i sat = _mm_xor_si128(_mm_set1_epi32(0x7fff), _mm_srai_epi32(hi, 31));  // 32767 or -32768 by the value's sign
            d[k] = _mm_or_si128(_mm_and_si128(fits, lo), _mm_andnot_si128(fits, sat));
        }
        const __m128i w0 = _mm_packs_epi32(d[0], d[1]), w1 = _mm_packs_epi32(d[2], d[3]);
        _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + i), _mm_packs_epi16(w0, w1));
    }
#endif
    for (; i < n; ++i) dst[i] = clamp_i64_to_i8(s[i]);
}

// Copy into the pinned staging buffer with streaming stores: the destination is written once and next read by the GPU's DMA
// engine, never by this CPU - ordinary stores first READ every destination line into the cache (a third of the memory
// traffic of the gather, which runs at the memory's rate, not the cores': profiles/r05_reads_timeline.md).
void stream_copy(void *dst, const void *src, size_t n) {
    char *d = static_cast<char *>(dst);
    const char *s = static_cast<const char *>(src);
#if !defined(__SSE2__)
    memcpy(d, s, n);  // no streaming stores on this host: the plain copy
    return;
#else
    if (n < 2048) {
        memcpy(d, s, n);
        return;
    }
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15;
    memcpy(d, s, head);
    d += head; s += head; n -= head;
    const size_t blocks = n / 64;
    for (size_t i = 0; i < blocks; ++i, s += 64, d += 64) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s)), b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + 16));
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + 32)), e = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + 48));
        _mm_stream_si128(reinterpret_cast<__m128i *>(d), a);
        _mm_stream_si128(reinterpret_cast<__m128i *>(d + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i *>(d + 32), c);
        _mm_stream_si128(reinterpret_cast<__m128i *>(d + 48), e);
    }
    memcpy(d, s, n - blocks * 64);
#endif
}


// The mapping of a read as int32 (sample indices inside one read's signal: they fit whenever the signal has fewer than 2^31
// samples - always, for int16 samples in memory): half the bytes of the int64 form on the way across PCIe, widened again on
// the device.  Returns false when a value does not survive the narrowing (the caller then ships int64).
bool narrow_i64_to_i32(const void *src, int64_t n, int32_t *dst) {
    const int64_t *s = static_cast<const int64_t *>(src);
    int64_t i = 0;
    bool ok = true;
#if defined(__SSE2__)
    __m128i bad = _mm_setzero_si128();
    for (; i + 4 <= n; i += 4) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i)), b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + i + 2));
        const __m128i lo = _mm_castps_si128(_mm_shuffle_ps(_mm_castsi128_ps(a), _mm_castsi128_ps(b), _MM_SHUFFLE(2, 0, 2, 0)));
        const __m128i hi = _mm_castps_si128(_mm_shuffle_ps(_mm_castsi128_ps(a), _mm_castsi128_ps(b), _MM_SHUFFLE(3, 1, 3, 1)));
        bad = _mm_or_si128(bad, _mm_xor_si128(hi, _mm_srai_epi32(lo, 31)));  // the high dword must be the sign of the low one
        _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + i), lo);
    }
    ok = _mm_movemask_epi8(_mm_cmpeq_epi32(bad, _mm_setzero_si128())) == 0xFFFF;
#endif
    for (; i < n; ++i) {
        dst[i] = (int32_t)s[i];
        ok = ok && (int64_t)dst[i] == s[i];
    }
    return ok;
}

}  // namespace

static int pack_reads_impl(int64_t n_reads, const void *const *dacs, const int64_t *sig_n, const void *const *maps,
                           const void *const *seqs, const int64_t *seq_n, const int32_t *seq_itemsize, int16_t *dst_dacs,
                           int64_t *dst_maps, int32_t *dst_maps32, int *maps_fit, int8_t *dst_seq, int64_t *sig_off, int64_t *seq_off,
                           int threads) {
    if (n_reads < 0 || !dacs || !sig_n || !maps || !seqs || !seq_n || !seq_itemsize || !dst_dacs || (!dst_maps && !dst_maps32) || !dst_seq ||
        !sig_off || !seq_off)
        RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    std::atomic<int> fit{1};
    sig_off[0] = seq_off[0] = 0;
    for (int64_t i = 0; i < n_reads; ++i) {
        if (sig_n[i] < 0 || seq_n[i] < 0) RMR_FAIL(RMR_ERR_INVALID, "read %lld: negative size", (long long)i);
        const int sz = seq_itemsize[i];
        if (sz != 1 && sz != 2 && sz != 4 && sz != 8) RMR_FAIL(RMR_ERR_INVALID, "read %lld: int_seq itemsize %d", (long long)i, sz);
        sig_off[i + 1] = sig_off[i] + sig_n[i];
        seq_off[i + 1] = seq_off[i] + seq_n[i];
    }
    if (threads < 1) threads = 1;
    if (threads > 32) threads = 32;
    // reads are dealt to the threads in contiguous ranges of about equal signal volume
    auto work = [&](int64_t r0, int64_t r1) {
        for (int64_t i = r0; i < r1; ++i) {
            stream_copy(dst_dacs + sig_off[i], dacs[i], (size_t)sig_n[i] * sizeof(int16_t));
            if (dst_maps32) {  // n + 1 entries per read
                if (!narrow_i64_to_i32(maps[i], seq_n[i] + 1, dst_maps32 + seq_off[i] + i)) fit.store(0);
            } else {
                stream_copy(dst_maps + seq_off[i] + i, maps[i], (size_t)(seq_n[i] + 1) * sizeof(int64_t));
            }
            int8_t *d = dst_seq + seq_off[i];
            switch (seq_itemsize[i]) {
                case 1: memcpy(d, seqs[i], (size_t)seq_n[i]); break;
                case 2: narrow_to_i8<int16_t>(seqs[i], seq_n[i], d); break;
                case 4: narrow_to_i8<int32_t>(seqs[i], seq_n[i], d); break;
                default: narrow_i64_to_i8(seqs[i], seq_n[i], d); break;
            }
        }
        _mm_sfence();  // the streamed lines are globally visible before the caller queues the upload
    };
    if (threads == 1 || n_reads < 2 * threads) {
        work(0, n_reads);
        if (maps_fit) *maps_fit = fit.load();
        return 0;
    }
    std::vector<std::thread> pool;
    const int64_t total = sig_off[n_reads];
    int64_t r0 = 0;
    for (int t = 0; t < threads && r0 < n_reads; ++t) {
        int64_t r1 = r0;
        const int64_t want = total * (t + 1) / threads;
        while (r1 < n_reads && (sig_off[r1 + 1] <= want || r1 == r0)) ++r1;
        if (t == threads - 1) r1 = n_reads;
        pool.emplace_back(work, r0, r1);
        r0 = r1;
    }
    for (auto &th : pool) th.join();
    if (maps_fit) *maps_fit = fit.load();
    return 0;
}

extern "C" int rmr_pack_reads(int64_t n_reads, const void *const *dacs, const int64_t *sig_n, const void *const *maps,
                              const void *const *seqs, const int64_t *seq_n, const int32_t *seq_itemsize, int16_t *dst_dacs,
                              int64_t *dst_maps, int8_t *dst_seq, int64_t *sig_off, int64_t *seq_off, int threads) {
    if (!dst_maps) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    return pack_reads_impl(n_reads, dacs, sig_n, maps, seqs, seq_n, seq_itemsize, dst_dacs, dst_maps, nullptr, nullptr, dst_seq, sig_off, seq_off,
                           threads);
}

extern "C" int rmr_pack_reads_narrow(int64_t n_reads, const void *const *dacs, const int64_t *sig_n, const void *const *maps,
                                     const void *const *seqs, const int64_t *seq_n, const int32_t *seq_itemsize, int16_t *dst_dacs,
                                     int32_t *dst_maps32, int8_t *dst_seq, int64_t *sig_off, int64_t *seq_off, int threads, int *maps_fit) {
    if (!dst_maps32 || !maps_fit) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    return pack_reads_impl(n_reads, dacs, sig_n, maps, seqs, seq_n, seq_itemsize, dst_dacs, nullptr, dst_maps32, maps_fit, dst_seq, sig_off,
                           seq_off, threads);
}

// The bases of the selected records of a BAM batch in read orientation, back to back, with their integer codes - what
// Read.add_alignment (seq = revcomp(query_sequence) for reverse-strand records, src/remora/io.py:2023; ref_seq likewise,
// :2058-2060) and util.seq_to_int (src/remora/util.py:131-142) do per read.  Record i: src[start[i] .. start[i] + len[i]);
// `upper`: ASCII lower case folded first (pysam's reference sequence marks mismatches in lower case); rev[i]: reversed and
// mapped through comp[256]; codes = code[256] of the oriented byte.  fwd (may be NULL): the folded, un-reversed bases.
// Output position of record i: the running sum of len.  The tables are the caller's: one definition of the alphabet.
extern "C" int rmr_orient_bases(const uint8_t *src, const int64_t *start, const int64_t *len, const uint8_t *rev, int64_t n, int upper,
                                const uint8_t *comp, const int8_t *code, uint8_t *fwd, uint8_t *oriented, int8_t *codes, int threads) {
    if (n < 0 || (n > 0 && (!src || !start || !len || !rev || !comp || !code || !oriented || !codes))) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    std::vector<int64_t> off((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; ++i) {
        if (len[i] < 0 || start[i] < 0) RMR_FAIL(RMR_ERR_INVALID, "record %lld: negative extent", (long long)i);
        off[i + 1] = off[i] + len[i];
    }
    uint8_t fold[256], comp_fold[256];
    for (int c = 0; c < 256; ++c) {
        fold[c] = (uint8_t)((upper && c >= 'a' && c <= 'z') ? c - 32 : c);
        comp_fold[c] = comp[fold[c]];
    }
    auto work = [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; ++i) {
            const uint8_t *s = src + start[i];
            const int64_t m = len[i];
            uint8_t *o = oriented + off[i];
            int8_t *k = codes + off[i];
            if (fwd) {
                uint8_t *f = fwd + off[i];
                for (int64_t j = 0; j < m; ++j) f[j] = fold[s[j]];
            }
            if (rev[i]) {
                for (int64_t j = 0; j < m; ++j) {
                    const uint8_t b = comp_fold[s[m - 1 - j]];
                    o[j] = b;
                    k[j] = code[b];
                }
            } else {
                for (int64_t j = 0; j < m; ++j) {
                    const uint8_t b = fold[s[j]];
                    o[j] = b;
                    k[j] = code[b];
                }
            }
        }
    };
    if (threads < 1) threads = 1;
    if (threads > 32) threads = 32;
    if (threads == 1 || n < 4 * threads) {
        work(0, n);
        return 0;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(work, n * t / threads, n * (t + 1) / threads);
    for (auto &th : pool) th.join();
    return 0;
}
