// mod_tags.cpp — the output side of `remora infer` for a whole batch of reads, host code:
//   rmr_format_mm_ml        : per-site probabilities -> MM:Z strings and ML:B:C arrays
//                             (util.format_mm_ml_tags, src/remora/util.py:485-537, called per read from
//                              inference.post_process_reads, src/remora/inference.py:429-459)
//   rmr_records_with_mod_tags: stored BAM records with their old modified-base tags removed and the new ones appended
//                             (what pysam.AlignedSegment.from_dict(io_read.full_align) + set_tag does per read,
//                              src/remora/inference.py:450, :619-623)
// The reference flags this stage as slow in its source (inference.py:55); per read it is a handful of numpy calls and string
// joins - 90 us of interpreter time per 7 kb read here, the largest single item on the consumer thread of a file-to-file run.
// One call per batch, no GPU.  Byte-for-byte the output of the Python implementations it replaces (tests/test_host_cpu.py).
#include "../../include/remora_hip.h"
#include "rmr_internal.h"

#include <array>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

using namespace rmr;

namespace {

inline int32_t rd32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }

// bytes of a tag value of type t at p; -1 on error / overrun (as bam_reader.cpp)
int64_t value_size(char t, const uint8_t *p, const uint8_t *end) {
    switch (t) {
        case 'A': case 'c': case 'C': return 1;
        case 's': case 'S': return 2;
        case 'i': case 'I': case 'f': return 4;
        case 'Z': case 'H': {
            const void *z = memchr(p, 0, (size_t)(end - p));
            return z ? (const uint8_t *)z - p + 1 : -1;
        }
        case 'B': {
            if (end - p < 5) return -1;
            const char sub = (char)p[0];
            const int64_t cnt = rd32(p + 1);
            const int w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : -1;
            if (w < 0 || cnt < 0) return -1;
            return 5 + cnt * w;
        }
        default: return -1;
    }
}

bool contains(const uint8_t *hay, size_t n, const char *pat3) {
    if (n < 3) return false;
    for (size_t i = 0; i + 3 <= n; ++i) {
        const void *f = memchr(hay + i, pat3[0], n - i - 2);
        if (!f) return false;
        i = (size_t)((const uint8_t *)f - hay);
        if (hay[i + 1] == (uint8_t)pat3[1] && hay[i + 2] == (uint8_t)pat3[2]) return true;
    }
    return false;
}

inline char *put_int(char *p, long long v) {  // decimal, as Python's str(int)
    char tmp[24];
    int n = 0;
    unsigned long long u = v < 0 ? (unsigned long long)(-(v + 1)) + 1ull : (unsigned long long)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *p++ = '-';
    while (n) *p++ = tmp[--n];
    return p;
}

}  // namespace

extern "C" {

int rmr_format_mm_ml(int64_t n_reads, const char *seq, const int64_t *seq_off, const int64_t *pos, const double *probs,
                     const int64_t *call_off, int n_mods, const char *mod_codes, char can_base, char strand, char *mm,
                     int64_t mm_cap, int64_t *mm_off, uint8_t *ml, int64_t ml_cap, int64_t *ml_off) {
    if (n_reads < 0 || n_mods < 1 || !seq_off || !call_off || !mod_codes || !mm_off || !ml_off || (n_reads > 0 && (!seq || !mm || !ml)))
        RMR_FAIL(RMR_ERR_INVALID, "bad argument");
    std::vector<std::string> codes;
    {
        const char *c = mod_codes;
        for (int m = 0; m < n_mods; ++m) {
            codes.emplace_back(c);
            c += codes.back().size() + 1;
        }
    }
    std::vector<int64_t> order;
    std::vector<long long> gaps;
    std::string gap_str;
    int64_t mm_len = 0, ml_len = 0;
    mm_off[0] = ml_off[0] = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        const int64_t c0 = call_off[r], nc = call_off[r + 1] - c0;
        if (nc < 0) RMR_FAIL(RMR_ERR_INVALID, "call offsets not ascending");
        if (nc > 0) {
            if (!pos || !probs) RMR_FAIL(RMR_ERR_INVALID, "NULL positions / probabilities");
            const int64_t *p = pos + c0;
            order.resize((size_t)nc);
            std::iota(order.begin(), order.end(), (int64_t)0);
            bool sorted = true;
            for (int64_t j = 1; j < nc; ++j) sorted = sorted && p[j - 1] <= p[j];
            if (!sorted) std::stable_sort(order.begin(), order.end(), [p](int64_t a, int64_t b) { return p[a] < p[b]; });
            // index of every called base among the canonical bases of the read: #(canonical bases at positions <= pos) - 1
            const char *s = seq + seq_off[r];
            const int64_t slen = seq_off[r + 1] - seq_off[r];
            gaps.resize((size_t)nc);
            long long count = 0, prev = -1;  // count = canonical bases in s[0 .. cursor)
            int64_t cursor = 0;
            for (int64_t j = 0; j < nc; ++j) {
                const int64_t sp = p[order[(size_t)j]];
                const int64_t upto = sp + 1 < 0 ? 0 : (sp + 1 > slen ? slen : sp + 1);  // positions 0 .. sp
                for (; cursor < upto; ++cursor) count += s[cursor] == can_base;
                const long long idx = count - 1;
                gaps[(size_t)j] = idx - prev - 1;
                prev = idx;
            }
            gap_str.clear();
            gap_str.reserve((size_t)nc * 4);
            char tmp[24];
            for (int64_t j = 0; j < nc; ++j) {
                char *e = put_int(tmp, gaps[(size_t)j]);
                if (j) gap_str.push_back(',');
                gap_str.append(tmp, (size_t)(e - tmp));
            }
            for (int m = 0; m < n_mods; ++m) {
                const int64_t need = 2 + (int64_t)codes[(size_t)m].size() + 2 + (int64_t)gap_str.size() + 1;
                if (mm_len + need > mm_cap || ml_len + nc > ml_cap) RMR_FAIL(RMR_ERR_INVALID, "output buffers too small");
                char *o = mm + mm_len;
                *o++ = can_base;
                *o++ = strand;
                memcpy(o, codes[(size_t)m].data(), codes[(size_t)m].size());
                o += codes[(size_t)m].size();
                *o++ = '?';
                *o++ = ',';
                memcpy(o, gap_str.data(), gap_str.size());
                o += gap_str.size();
                *o++ = ';';
                mm_len += need;
                for (int64_t j = 0; j < nc; ++j) {
                    double v = std::floor(probs[(size_t)(c0 + order[(size_t)j]) * (size_t)n_mods + (size_t)m] * 256.0);
                    if (v == 256.0) v = 255.0;
                    ml[ml_len + j] = (uint8_t)(long long)v;
                }
                ml_len += nc;
            }
        }
        mm_off[r + 1] = mm_len;
        ml_off[r + 1] = ml_len;
    }
    return 0;
}

}  // extern "C"

namespace {

// the records of a batch re-emitted with their new modified-base tags; ref_seq != nullptr: a record that gets tags and owns
// a slice of ref_seq leaves in the reference-anchored form (see rmr_records_with_mod_tags_ref)
int rewrite_records(int64_t n_reads, const uint8_t *const *raw, const int64_t *raw_len, const int64_t *tags_off, const char *mm,
                    const int64_t *mm_off, const uint8_t *ml, const int64_t *ml_off, const uint8_t *has_tags, const char *ref_seq,
                    const int64_t *ref_off, uint8_t *out, int64_t out_cap, int64_t *out_len) {
    if (n_reads < 0 || !out_len || (n_reads > 0 && (!raw || !raw_len || !tags_off || !has_tags || !out)))
        RMR_FAIL(RMR_ERR_INVALID, "bad argument");
    static const std::array<uint8_t, 256> nt16 = [] {  // base letter -> BAM's 4-bit code; anything else -> 15 (N)
        std::array<uint8_t, 256> t;
        t.fill(15);
        const char *letters = "=ACMGRSVTWYHKDBN";
        for (int i = 0; i < 16; ++i) t[(uint8_t)letters[i]] = (uint8_t)i;
        return t;
    }();
    int64_t w = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        const uint8_t *rec = raw[r];
        const int64_t len = raw_len[r], to = tags_off[r];
        if (!rec || to < 32 || to > len) RMR_FAIL(RMR_ERR_INVALID, "record %lld: bad tag offset", (long long)r);
        const int64_t mml = has_tags[r] ? mm_off[r + 1] - mm_off[r] : 0, mll = has_tags[r] ? ml_off[r + 1] - ml_off[r] : 0;
        const int64_t extra = has_tags[r] ? 3 + mml + 1 + 4 + 4 + mll : 0;
        const int64_t nref = (ref_seq && ref_off && has_tags[r]) ? ref_off[r + 1] - ref_off[r] : 0;
        if (w + 4 + len + extra + (nref ? 4 + (nref + 1) / 2 + nref : 0) > out_cap) RMR_FAIL(RMR_ERR_INVALID, "output buffer too small");
        uint8_t *body = out + w + 4;
        int64_t b;
        if (nref > 0) {
            // the reference-anchored record the reference writes (src/remora/inference.py:452-458): CIGAR <n>M, the reference
            // bases of the alignment (forward strand), no qualities; every other fixed field and the name as they are
            const int64_t l_name = rec[8];
            if (32 + l_name > to) RMR_FAIL(RMR_ERR_INVALID, "record %lld: name runs into the tags", (long long)r);
            memcpy(body, rec, (size_t)(32 + l_name));
            const uint16_t one = 1;
            const int32_t l_seq = (int32_t)nref;
            memcpy(body + 12, &one, 2);
            memcpy(body + 16, &l_seq, 4);
            b = 32 + l_name;
            const uint32_t cig = ((uint32_t)nref << 4) | 0u;
            memcpy(body + b, &cig, 4);
            b += 4;
            const uint8_t *q = reinterpret_cast<const uint8_t *>(ref_seq) + ref_off[r];
            for (int64_t k = 0; k + 1 < nref; k += 2) body[b++] = (uint8_t)((nt16[q[k]] << 4) | nt16[q[k + 1]]);
            if (nref & 1) body[b++] = (uint8_t)(nt16[q[nref - 1]] << 4);
            memset(body + b, 0xff, (size_t)nref);
            b += nref;
        } else {
            memcpy(body, rec, (size_t)to);
            b = to;
        }
        const uint8_t *tags = rec + to, *end = rec + len;
        // records that carry no modified-base tag keep their tag bytes as they are; only a record in which one of the four
        // tag headers occurs (as a tag, or by chance inside another tag's data) is walked tag by tag
        const size_t tn = (size_t)(len - to);
        if (contains(tags, tn, "MMZ") || contains(tags, tn, "MLB") || contains(tags, tn, "MmZ") || contains(tags, tn, "MlB")) {
            const uint8_t *p = tags;
            while (p + 3 <= end) {
                const int64_t sz = value_size((char)p[2], p + 3, end);
                if (sz < 0 || p + 3 + sz > end) RMR_FAIL(RMR_ERR_INVALID, "record %lld: corrupt tag region", (long long)r);
                const bool mod = p[0] == 'M' && (p[1] == 'M' || p[1] == 'L' || p[1] == 'm' || p[1] == 'l');
                if (!mod) {
                    memcpy(body + b, p, (size_t)(3 + sz));
                    b += 3 + sz;
                }
                p += 3 + sz;
            }
            if (p != end) RMR_FAIL(RMR_ERR_INVALID, "record %lld: corrupt tag region", (long long)r);
        } else {
            memcpy(body + b, tags, tn);
            b += (int64_t)tn;
        }
        if (has_tags[r]) {
            memcpy(body + b, "MMZ", 3);
            b += 3;
            if (mml) memcpy(body + b, mm + mm_off[r], (size_t)mml);
            b += mml;
            body[b++] = 0;
            memcpy(body + b, "MLBC", 4);
            b += 4;
            const int32_t cnt = (int32_t)mll;
            memcpy(body + b, &cnt, 4);
            b += 4;
            if (mll) memcpy(body + b, ml + ml_off[r], (size_t)mll);
            b += mll;
        }
        const int32_t bs = (int32_t)b;
        memcpy(out + w, &bs, 4);
        w += 4 + b;
    }
    *out_len = w;
    return 0;
}

}  // namespace

extern "C" {

int rmr_records_with_mod_tags(int64_t n_reads, const uint8_t *const *raw, const int64_t *raw_len, const int64_t *tags_off,
                              const char *mm, const int64_t *mm_off, const uint8_t *ml, const int64_t *ml_off,
                              const uint8_t *has_tags, uint8_t *out, int64_t out_cap, int64_t *out_len) {
    return rewrite_records(n_reads, raw, raw_len, tags_off, mm, mm_off, ml, ml_off, has_tags, nullptr, nullptr, out, out_cap, out_len);
}

int rmr_records_with_mod_tags_ref(int64_t n_reads, const uint8_t *const *raw, const int64_t *raw_len, const int64_t *tags_off,
                                  const char *mm, const int64_t *mm_off, const uint8_t *ml, const int64_t *ml_off,
                                  const uint8_t *has_tags, const char *ref_seq, const int64_t *ref_off, uint8_t *out, int64_t out_cap,
                                  int64_t *out_len) {
    if (n_reads > 0 && (!ref_seq || !ref_off)) RMR_FAIL(RMR_ERR_INVALID, "bad argument");
    return rewrite_records(n_reads, raw, raw_len, tags_off, mm, mm_off, ml, ml_off, has_tags, ref_seq, ref_off, out, out_cap, out_len);
}

}  // extern "C"
