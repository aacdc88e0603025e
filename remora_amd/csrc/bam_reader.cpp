// Native BGZF / BAM record reader for the POD5+BAM ingest.
//
// The reference reads alignments through pysam / htslib (src/remora/io.py:184-358, ReadIndexedBam, and
// Read.add_alignment :1972-2084 for what it takes from a record).  Here the same fields are produced a batch
// of records at a time: BGZF members are inflated with zlib, records are split and their fixed fields, name,
// CIGAR, 4-bit sequence and the tags the hot path needs (mv, ts, ns, sp, sm, sd, pi, MD) are decoded into flat
// arrays the Python host wraps without a per-record parse; optionally the reference bases of the alignment
// are rebuilt from query + CIGAR + MD (what pysam's get_reference_sequence returns, mismatches in lower case).
//
// Host code only (no device work): plain C++17 + zlib, part of libremora_hip.so.
#include <array>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <memory>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/remora_hip.h"
#include "rmr_internal.h"
#include "fast_inflate.h"
#include "crc32_fast.h"

using rmr::set_error;

struct rmr_bam {
    FILE *fh = nullptr;
    std::vector<uint8_t> cbuf;    // compressed block
    std::vector<uint8_t> ubuf;    // inflated, not yet consumed bytes
    size_t upos = 0;              // consumed prefix of ubuf
    struct Seg { size_t begin; int64_t file_off; uint32_t isize; };  // ubuf[begin, begin+isize) came from the member at file_off
    std::vector<Seg> segs;
    std::vector<int64_t> voff;    // per record of the batch: BGZF virtual offset (file_off << 16 | offset in block)
    bool eof = false;
    // BGZF members are independent deflate streams: kSlots of them are read ahead and inflated by as many threads
    static constexpr int kSlots = 32;  // upper bound; `nslots` of them are in use
    bool check_crc = true;
    int nslots = 8;                    // RMR_BAM_INFLATE_THREADS at open (a launcher that scans for all of its ranks asks for more)
    struct Slot {
        std::vector<uint8_t> cbuf, out;
        uint32_t crc = 0, isize = 0;
        int clen = 0;
        int64_t file_off = 0;
        z_stream zs{};
        bool zs_init = false;
        std::unique_ptr<rmr_inflate::Tables> tables;  // decode tables of the one-shot inflater (fast_inflate.h)
        bool check = true;  // verify the member's CRC32 (off during rmr_bam_scan: whoever reads the records verifies them)
        int rc = 0;
    } slot[kSlots];
    // persistent inflate workers (creating threads per batch of members costs more than the inflate itself in a
    // process that carries the HIP runtime's thread-local state)
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    uint64_t generation = 0;
    int n_active = 0, n_pending = 0;
    bool stop = false;
    // one generation of members is inflated AHEAD of the parser: dispatched when the previous one has been appended to ubuf,
    // collected when the parser runs out of bytes (the parse of a batch - copies, base decoding, tag walk - is serial and
    // costs as much as the inflate; before, the two took turns)
    bool inflight = false;
    int inflight_n = 0;
    int dispatch_err = 0;
    std::string dispatch_msg;
    std::vector<uint8_t> header;  // everything before the first record (magic, text, references)
    int64_t first_voffset = -1;   // of the first record (-1: the file holds none)
    std::vector<std::string> refs;
    // batch arenas (valid until the next read_batch call)
    std::vector<int32_t> flag, ref_id, pos, mapq, l_seq, n_cigar, ts, ns, sp;
    std::vector<float> sm, sd;
    std::vector<uint8_t> has;  // bit0 mv, 1 ts, 2 ns, 3 sp, 4 sm, 5 sd, 6 pi, 7 MD
    std::vector<uint8_t> ref_ok;
    std::vector<int64_t> raw_off, name_off, seq_off, cigar_off, tags_off, mv_off, pi_off, md_off, refseq_off;
    std::vector<uint8_t> raw;
    std::vector<char> names, seq, pi, md, refseq;
    std::vector<uint32_t> cigar;
    std::vector<int8_t> mv;
};

namespace {

// ---- BGZF ---------------------------------------------------------------------------------
// reads the next BGZF member (compressed) into slot k; returns 1 = read, 0 = clean EOF, negative = error
int read_member(rmr_bam *b, rmr_bam::Slot &sl) {
    uint8_t hd[12];
    sl.file_off = (int64_t)ftello(b->fh);
    const size_t got = fread(hd, 1, 12, b->fh);
    if (got == 0) return 0;
    if (got != 12 || hd[0] != 0x1f || hd[1] != 0x8b || hd[2] != 8 || !(hd[3] & 4)) {
        set_error("not a BAM file: not a BGZF block (truncated or plain gzip)");
        return RMR_ERR_INVALID;
    }
    const int xlen = hd[10] | (hd[11] << 8);
    uint8_t extra[65536];
    if (fread(extra, 1, (size_t)xlen, b->fh) != (size_t)xlen) {
        set_error("truncated BAM file");
        return RMR_ERR_INVALID;
    }
    int bsize = -1;
    for (int p = 0; p + 4 <= xlen;) {
        const int slen = extra[p + 2] | (extra[p + 3] << 8);
        if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2 && p + 6 <= xlen) bsize = extra[p + 4] | (extra[p + 5] << 8);
        p += 4 + slen;
    }
    if (bsize < 0) {
        set_error("BGZF block without BC field");
        return RMR_ERR_INVALID;
    }
    sl.clen = bsize - xlen - 19;  // deflate payload
    if (sl.clen < 0) {
        set_error("corrupt BGZF block size");
        return RMR_ERR_INVALID;
    }
    sl.cbuf.resize((size_t)sl.clen + 8 + 8);  // + the trailer, + padding the one-shot inflater may read into (fast_inflate.h: 16 bytes)
    if (fread(sl.cbuf.data(), 1, (size_t)sl.clen + 8, b->fh) != (size_t)sl.clen + 8) {
        set_error("truncated BAM file");
        return RMR_ERR_INVALID;
    }
    const uint8_t *tail = sl.cbuf.data() + sl.clen;
    sl.crc = tail[0] | (tail[1] << 8) | (tail[2] << 16) | ((uint32_t)tail[3] << 24);
    sl.isize = tail[4] | (tail[5] << 8) | (tail[6] << 16) | ((uint32_t)tail[7] << 24);
    return 1;
}

// inflates slot's member into slot.out (thread-safe: touches the slot only); rc in slot.rc: 0 ok, 1 inflate, 2 crc
void inflate_member(rmr_bam::Slot &sl) {
    sl.out.resize(sl.isize);
    sl.rc = 0;
    if (sl.isize == 0) return;  // empty member (e.g. the EOF marker)
    // the one-shot decoder first (2-3x zlib on BAM records); only where the member's CRC32 is verified anyway, so that a
    // stream it refuses - or ever got wrong - goes through zlib below
    static const bool fast = !(getenv("RMR_FAST_INFLATE") && atoi(getenv("RMR_FAST_INFLATE")) == 0);
    if (fast && sl.check) {
        if (!sl.tables) sl.tables.reset(new rmr_inflate::Tables);
        if (rmr_inflate::inflate_raw(sl.cbuf.data(), (size_t)sl.clen, sl.out.data(), sl.isize, *sl.tables) &&
            rmr_crc::crc32(sl.out.data(), sl.isize) == sl.crc)
            return;
    }
    if (!sl.zs_init) {
        if (inflateInit2(&sl.zs, -15) != Z_OK) { sl.rc = 1; return; }
        sl.zs_init = true;
    } else {
        inflateReset(&sl.zs);
    }
    sl.zs.next_in = sl.cbuf.data();
    sl.zs.avail_in = (uInt)sl.clen;
    sl.zs.next_out = sl.out.data();
    sl.zs.avail_out = sl.isize;
    if (inflate(&sl.zs, Z_FINISH) != Z_STREAM_END || sl.zs.avail_out != 0) { sl.rc = 1; return; }
    if (sl.check && rmr_crc::crc32(sl.out.data(), sl.isize) != sl.crc) sl.rc = 2;
}

// worker w inflates slot w of every generation that has that many members
void worker_loop(rmr_bam *b, int w) {
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(b->mu);
            b->cv_work.wait(lk, [&] { return b->stop || b->generation != seen; });
            if (b->stop) return;
            seen = b->generation;
            if (w >= b->n_active) continue;
        }
        inflate_member(b->slot[w]);
        {
            std::lock_guard<std::mutex> lk(b->mu);
            --b->n_pending;
        }
        b->cv_done.notify_one();
    }
}

// reads up to nslots members and hands them to the workers; returns their number (0 = clean EOF) or a negative error
int dispatch(rmr_bam *b) {
    int n = 0, rc = 1;
    while (n < b->nslots) {
        rc = read_member(b, b->slot[n]);
        if (rc <= 0) break;
        b->slot[n].check = b->check_crc;
        ++n;
    }
    if (rc < 0) return rc;
    if (n == 0) return 0;
    if (b->workers.empty())
        for (int w = 0; w < b->nslots; ++w) b->workers.emplace_back(worker_loop, b, w);
    {
        std::lock_guard<std::mutex> lk(b->mu);
        b->n_active = n;
        b->n_pending = n;
        ++b->generation;
    }
    b->cv_work.notify_all();
    b->inflight = true;
    b->inflight_n = n;
    return n;
}

// waits for the generation in flight (if any) and forgets it: in front of every reposition of the file
void drop_inflight(rmr_bam *b) {
    if (b->inflight) {
        std::unique_lock<std::mutex> lk(b->mu);
        b->cv_done.wait(lk, [&] { return b->n_pending == 0; });
        b->inflight = false;
    }
    b->dispatch_err = 0;
}

// appends the next generation of inflated members to b->ubuf in file order and sends the one after it on its way;
// returns the number of members appended (0 = clean EOF) or a negative error
int next_blocks(rmr_bam *b) {
    if (b->dispatch_err) {  // the read-ahead failed last time: reported now that the parser has come this far
        const int rc = b->dispatch_err;
        b->dispatch_err = 0;
        set_error("%s", b->dispatch_msg.c_str());
        return rc;
    }
    if (!b->inflight) {
        const int rc = dispatch(b);
        if (rc <= 0) return rc;
    }
    {
        std::unique_lock<std::mutex> lk(b->mu);
        b->cv_done.wait(lk, [&] { return b->n_pending == 0; });
    }
    b->inflight = false;
    const int n = b->inflight_n;
    if (b->upos > 0 && b->upos >= b->ubuf.size() / 2) {  // compact the consumed prefix before growing
        const size_t cut = b->upos;
        b->ubuf.erase(b->ubuf.begin(), b->ubuf.begin() + (ptrdiff_t)cut);
        b->upos = 0;
        size_t keep = 0;
        for (auto &sg : b->segs) {  // drop fully consumed members, shift the rest (begin may become "negative")
            if (sg.begin + sg.isize <= cut) continue;
            rmr_bam::Seg t = sg;
            t.begin = sg.begin - cut;  // size_t wrap-around is fine: only begin + offset sums are used
            b->segs[keep++] = t;
        }
        b->segs.resize(keep);
    }
    for (int k = 0; k < n; ++k) {
        if (b->slot[k].rc != 0) {
            set_error(b->slot[k].rc == 1 ? "corrupt BGZF block (inflate)" : "corrupt BGZF block (crc)");
            return RMR_ERR_INVALID;
        }
        if (b->slot[k].isize) b->segs.push_back({b->ubuf.size(), b->slot[k].file_off, b->slot[k].isize});
        b->ubuf.insert(b->ubuf.end(), b->slot[k].out.begin(), b->slot[k].out.end());
    }
    const int ahead = dispatch(b);  // the slots are free again: the next generation inflates while the caller parses this one
    if (ahead < 0) {
        b->dispatch_err = ahead;
        b->dispatch_msg = rmr_last_error();
    }
    return n;
}

// makes at least n unread bytes available; returns 1, 0 (EOF before n bytes; *avail says how many there are) or <0
int ensure(rmr_bam *b, size_t n) {
    while (b->ubuf.size() - b->upos < n) {
        if (b->eof) return 0;
        const int rc = next_blocks(b);
        if (rc < 0) return rc;
        if (rc == 0) b->eof = true;
    }
    return 1;
}

inline int32_t rd_i32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }
inline uint32_t rd_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

// BGZF virtual offset (member's file offset << 16 | offset inside the inflated member) of byte `pos` of ubuf
int64_t voffset_at(const rmr_bam *b, size_t pos) {
    for (const auto &sg : b->segs) {
        const size_t rel = pos - sg.begin;  // (wraps for members in front of pos: then rel >= isize)
        if (rel < sg.isize) return (sg.file_off << 16) | (int64_t)rel;
    }
    return -1;
}

// ---- where does a record start?  (rmr_bam_guess_start) ---------------------------------------------------
// A worker that begins in the middle of the file needs a record boundary without walking the block_size chain from
// the first record.  A position is accepted when a record AND the records chained behind it (kGuessChain of them, or
// up to the end of the file) pass every check the format allows: field ranges against the header's reference count,
// a NUL-terminated name without control characters, CIGAR operation codes, and a tag region that parses tag by tag and ends exactly
// where block_size says.  The worker in front of this one verifies the guess for certain: its own chain of records
// must END on it (io.py: a share that runs past its end mark is an error, never a silent overlap).
constexpr int kGuessChain = 8;

// length (4 + block_size) of a valid record at ubuf[q] (upos is 0 during the search), 0 = not a record here, < 0 = I/O error
int64_t plausible_record(rmr_bam *b, size_t q) {
    int rc = ensure(b, q + 36);
    if (rc < 0) return rc;
    if (rc == 0) return 0;
    const uint8_t *r = b->ubuf.data() + q;
    const int64_t bs = rd_i32(r), n_ref = (int64_t)b->refs.size();
    const int64_t ref_id = rd_i32(r + 4), pos = rd_i32(r + 8), l_name = r[12], n_cig = r[16] | (r[17] << 8), l_seq = rd_i32(r + 20);
    const int64_t nref_id = rd_i32(r + 24), npos = rd_i32(r + 28);
    if (bs < 32 + 1 || bs > (1 << 28) || ref_id < -1 || ref_id >= n_ref || pos < -1 || l_name < 1 || l_seq < 0) return 0;
    if (nref_id < -1 || nref_id >= n_ref || npos < -1) return 0;
    const int64_t fixed = 32 + l_name + 4 * n_cig + (l_seq + 1) / 2 + l_seq;
    if (fixed > bs) return 0;
    rc = ensure(b, q + 36 + (size_t)l_name);
    if (rc < 0) return rc;
    if (rc == 0) return 0;
    r = b->ubuf.data() + q;
    if (r[36 + l_name - 1] != 0) return 0;
    for (int64_t i = 0; i + 1 < l_name; ++i)
        if (r[36 + i] < 33 || r[36 + i] == 127) return 0;  // (bytes >= 0x80 pass: names in UTF-8 occur in the wild)
    rc = ensure(b, q + 4 + (size_t)bs);  // (only a position that looks like a record so far makes the window grow)
    if (rc < 0) return rc;
    if (rc == 0) return 0;
    r = b->ubuf.data() + q;
    const uint8_t *cig = r + 36 + l_name;
    for (int64_t i = 0; i < n_cig; ++i)
        if ((rd_u32(cig + 4 * i) & 0xF) > 8) return 0;
    const uint8_t *t = r + 4 + fixed, *end = r + 4 + bs;
    while (t < end) {
        if (end - t < 4 || !isalpha(t[0]) || !isalnum(t[1])) return 0;
        const char ty = (char)t[2];
        t += 3;
        int64_t sz;
        switch (ty) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': {
                const void *z = memchr(t, 0, (size_t)(end - t));
                if (!z) return 0;
                sz = (const uint8_t *)z - t + 1;
                break;
            }
            case 'B': {
                if (end - t < 5) return 0;
                int64_t el;
                switch ((char)t[0]) {
                    case 'c': case 'C': el = 1; break;
                    case 's': case 'S': el = 2; break;
                    case 'i': case 'I': case 'f': el = 4; break;
                    default: return 0;
                }
                const int64_t cnt = rd_i32(t + 1);
                if (cnt < 0) return 0;
                sz = 5 + el * cnt;
                break;
            }
            default: return 0;
        }
        if (sz > end - t) return 0;
        t += sz;
    }
    return 4 + bs;
}

// 1 = a chain of records starts at ubuf[p], 0 = not, < 0 = I/O error
int record_chain_at(rmr_bam *b, size_t p) {
    size_t q = p;
    for (int k = 0; k < kGuessChain; ++k) {
        const int rc = ensure(b, q + 1);
        if (rc < 0) return rc;
        if (rc == 0) return (k > 0 && q == b->ubuf.size()) ? 1 : 0;  // the file ends behind a record: as good as a chain
        const int64_t len = plausible_record(b, q);
        if (len < 0) return (int)len;
        if (len == 0) return 0;
        q += (size_t)len;
    }
    return 1;
}

// a BGZF member header at buf[i] (n bytes available): its total size, or 0
int member_size_at(const uint8_t *buf, size_t n, size_t i) {
    if (i + 18 > n || buf[i] != 0x1f || buf[i + 1] != 0x8b || buf[i + 2] != 8 || !(buf[i + 3] & 4)) return 0;
    const int xlen = buf[i + 10] | (buf[i + 11] << 8);
    if (i + 12 + (size_t)xlen > n) return 0;
    for (int p = 0; p + 4 <= xlen;) {
        const uint8_t *e = buf + i + 12 + p;
        const int slen = e[2] | (e[3] << 8);
        if (e[0] == 'B' && e[1] == 'C' && slen == 2 && p + 6 <= xlen) {
            const int bsize = (e[4] | (e[5] << 8)) + 1;
            return bsize >= xlen + 20 ? bsize : 0;
        }
        p += 4 + slen;
    }
    return 0;
}


const char NT16[] = "=ACMGRSVTWYHKDBN";

// size in bytes of a tag value at p (type t); -1 on error / overrun
int64_t tag_value_size(char t, const uint8_t *p, const uint8_t *end) {
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
            const int64_t cnt = rd_i32(p + 1);
            int w = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : -1;
            if (w < 0 || cnt < 0) return -1;
            return 5 + cnt * w;
        }
        default: return -1;
    }
}

bool tag_int(char t, const uint8_t *p, int32_t *out) {
    switch (t) {
        case 'c': *out = (int8_t)p[0]; return true;
        case 'C': *out = p[0]; return true;
        case 's': { int16_t v; memcpy(&v, p, 2); *out = v; return true; }
        case 'S': { uint16_t v; memcpy(&v, p, 2); *out = v; return true; }
        case 'i': *out = rd_i32(p); return true;
        case 'I': *out = (int32_t)rd_u32(p); return true;
        default: return false;
    }
}

// reference bases of the alignment from query + CIGAR + MD; false when MD and CIGAR disagree / MD malformed
// (also when the CIGAR consumes more query bases than SEQ holds: SEQ '*' on a secondary record, or a corrupt record)
bool rebuild_reference(const char *query, size_t l_seq, const uint32_t *cig, size_t n_cig, const char *md, size_t md_len,
                       std::string &cols, std::vector<char> &out) {
    cols.clear();
    size_t q = 0;
    for (size_t k = 0; k < n_cig; ++k) {
        const uint32_t op = cig[k] & 0xF;
        const size_t ln = cig[k] >> 4;
        if (op == 0 || op == 7 || op == 8) {
            if (ln > l_seq - q) return false;  // q <= l_seq is an invariant of this loop
            cols.append(query + q, ln);
            q += ln;
        } else if (op == 1 || op == 4) {
            if (ln > l_seq - q) return false;
            q += ln;
        } else if (op == 2 || op == 3) {
            cols.append(ln, '-');
        }
    }
    const size_t start = out.size();
    size_t i = 0, p = 0;
    while (p < md_len) {
        const unsigned char c = (unsigned char)md[p];
        if (isdigit(c)) {
            size_t run = 0;
            while (p < md_len && isdigit((unsigned char)md[p])) { run = run * 10 + (size_t)(md[p] - '0'); ++p; }
            const size_t take = (i < cols.size()) ? ((run < cols.size() - i) ? run : cols.size() - i) : 0;
            if (take) out.insert(out.end(), cols.begin() + (ptrdiff_t)i, cols.begin() + (ptrdiff_t)(i + take));
            i += run;
        } else if (c == '^') {
            size_t e = p + 1;
            while (e < md_len && isalpha((unsigned char)md[e])) ++e;
            if (e == p + 1) { out.resize(start); return false; }
            for (size_t k = p + 1; k < e; ++k) out.push_back((char)toupper((unsigned char)md[k]));
            i += e - p - 1;
            p = e;
        } else if (isalpha(c)) {
            out.push_back((char)tolower(c));
            ++i;
            ++p;
        } else {
            out.resize(start);
            return false;
        }
    }
    if (i != cols.size()) { out.resize(start); return false; }
    return true;
}

void stop_workers(rmr_bam *b) {
    drop_inflight(b);
    {
        std::lock_guard<std::mutex> lk(b->mu);
        b->stop = true;
    }
    b->cv_work.notify_all();
    for (auto &t : b->workers) t.join();
    b->workers.clear();
}

}  // namespace

extern "C" {

int rmr_bam_open(const char *path, rmr_bam **out) { return rmr_bam_open_threads(path, 0, out); }

int rmr_bam_open_threads(const char *path, int inflate_threads, rmr_bam **out) {
    if (!path || !out) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    std::unique_ptr<rmr_bam> b(new rmr_bam());
    b->fh = fopen(path, "rb");
    if (!b->fh) RMR_FAIL(RMR_ERR_INVALID, "cannot open %s", path);
    if (inflate_threads >= 1) {
        b->nslots = inflate_threads > rmr_bam::kSlots ? rmr_bam::kSlots : inflate_threads;
    } else if (const char *ev = getenv("RMR_BAM_INFLATE_THREADS")) {
        const int v = atoi(ev);
        if (v >= 1) b->nslots = v > rmr_bam::kSlots ? rmr_bam::kSlots : v;
    }
    auto fail = [&](int rc) {
        stop_workers(b.get());
        fclose(b->fh);
        b->fh = nullptr;
        for (auto &sl : b->slot) if (sl.zs_init) { inflateEnd(&sl.zs); sl.zs_init = false; }
        return rc;
    };
    int rc = ensure(b.get(), 12);
    if (rc < 0) return fail(rc);
    if (rc == 0 || memcmp(b->ubuf.data(), "BAM\x01", 4) != 0) { set_error("%s is not a BAM file", path); return fail(RMR_ERR_INVALID); }
    const int64_t l_text = rd_i32(b->ubuf.data() + 4);
    if (l_text < 0) { set_error("corrupt BAM header"); return fail(RMR_ERR_INVALID); }
    rc = ensure(b.get(), 12 + (size_t)l_text);
    if (rc <= 0) { if (rc == 0) set_error("truncated BAM file"); return fail(rc < 0 ? rc : RMR_ERR_INVALID); }
    size_t p = 8 + (size_t)l_text;
    const int64_t n_ref = rd_i32(b->ubuf.data() + p);
    p += 4;
    for (int64_t r = 0; r < n_ref; ++r) {
        rc = ensure(b.get(), p + 4);
        if (rc <= 0) { if (rc == 0) set_error("truncated BAM file"); return fail(rc < 0 ? rc : RMR_ERR_INVALID); }
        const int64_t l_name = rd_i32(b->ubuf.data() + p);
        rc = ensure(b.get(), p + 4 + (size_t)l_name + 4);
        if (rc <= 0 || l_name < 1) { if (rc >= 0) set_error("truncated BAM file"); return fail(rc < 0 ? rc : RMR_ERR_INVALID); }
        b->refs.emplace_back(reinterpret_cast<const char *>(b->ubuf.data() + p + 4), (size_t)l_name - 1);
        p += 4 + (size_t)l_name + 4;
    }
    b->header.assign(b->ubuf.begin(), b->ubuf.begin() + (ptrdiff_t)p);
    b->upos = p;
    rc = ensure(b.get(), 1);
    if (rc < 0) return fail(rc);
    b->first_voffset = rc == 0 ? -1 : voffset_at(b.get(), b->upos);
    *out = b.release();
    return 0;
}

void rmr_bam_close(rmr_bam *b) {
    if (!b) return;
    stop_workers(b);
    if (b->fh) fclose(b->fh);
    for (auto &sl : b->slot) if (sl.zs_init) inflateEnd(&sl.zs);
    delete b;
}

int rmr_bam_header(rmr_bam *b, const uint8_t **bytes, int64_t *n_bytes, int64_t *n_refs) {
    if (!b || !bytes || !n_bytes) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    *bytes = b->header.data();
    *n_bytes = (int64_t)b->header.size();
    if (n_refs) *n_refs = (int64_t)b->refs.size();
    return 0;
}

const char *rmr_bam_ref_name(rmr_bam *b, int64_t ref_id) {
    if (!b || ref_id < 0 || ref_id >= (int64_t)b->refs.size()) return nullptr;
    return b->refs[(size_t)ref_id].c_str();
}

int rmr_bam_read_batch(rmr_bam *b, int64_t max_records, int want_ref, rmr_bam_batch *out) {
    if (!b || !out || max_records < 0) RMR_FAIL(RMR_ERR_INVALID, "bad argument");
    auto clr = [](auto &v) { v.clear(); };
    clr(b->flag); clr(b->ref_id); clr(b->pos); clr(b->mapq); clr(b->l_seq); clr(b->n_cigar); clr(b->ts); clr(b->ns);
    clr(b->sp); clr(b->sm); clr(b->sd); clr(b->has); clr(b->ref_ok); clr(b->raw); clr(b->names); clr(b->seq); clr(b->pi);
    clr(b->md); clr(b->refseq); clr(b->cigar); clr(b->mv); clr(b->voff);
    for (auto *v : {&b->raw_off, &b->name_off, &b->seq_off, &b->cigar_off, &b->tags_off, &b->mv_off, &b->pi_off, &b->md_off,
                    &b->refseq_off}) { v->clear(); }
    for (auto *v : {&b->raw_off, &b->name_off, &b->seq_off, &b->cigar_off, &b->mv_off, &b->pi_off, &b->md_off, &b->refseq_off})
        v->push_back(0);
    std::string cols;
    int64_t n = 0;
    while (n < max_records) {
        int rc = ensure(b, 4);
        if (rc < 0) return rc;
        if (rc == 0) {
            if (b->ubuf.size() - b->upos != 0) RMR_FAIL(RMR_ERR_INVALID, "truncated BAM file");
            break;
        }
        const int64_t bs = rd_i32(b->ubuf.data() + b->upos);
        if (bs < 32) RMR_FAIL(RMR_ERR_INVALID, "corrupt BAM record");
        int64_t vo = -1;
        for (const auto &sg : b->segs) {  // few entries: the members currently buffered
            const size_t rel = b->upos - sg.begin;  // wraps for members that start before the buffer
            if (rel < sg.isize) { vo = (sg.file_off << 16) | (int64_t)rel; break; }
        }
        rc = ensure(b, 4 + (size_t)bs);
        if (rc < 0) return rc;
        if (rc == 0) RMR_FAIL(RMR_ERR_INVALID, "truncated BAM file");
        const uint8_t *rec = b->ubuf.data() + b->upos + 4;
        const uint8_t *end = rec + bs;
        const int32_t ref_id = rd_i32(rec), pos = rd_i32(rec + 4);
        const int l_read_name = rec[8], mapq = rec[9];
        const int n_cig = rec[12] | (rec[13] << 8), flag = rec[14] | (rec[15] << 8);
        const int64_t l_seq = rd_i32(rec + 16);
        const uint8_t *p = rec + 32;
        if (l_seq < 0 || l_read_name < 1 || p + l_read_name + 4 * (int64_t)n_cig + (l_seq + 1) / 2 + l_seq > end)
            RMR_FAIL(RMR_ERR_INVALID, "corrupt BAM record");
        b->voff.push_back(vo);
        b->flag.push_back(flag); b->ref_id.push_back(ref_id); b->pos.push_back(pos); b->mapq.push_back(mapq);
        b->l_seq.push_back((int32_t)l_seq); b->n_cigar.push_back(n_cig);
        const bool ids_only = (want_ref & 2) != 0;  // flags, names and tags scalars; no record bytes, bases, CIGAR, move table
        if (!ids_only) b->raw.insert(b->raw.end(), rec, end);
        b->raw_off.push_back((int64_t)b->raw.size());
        b->names.insert(b->names.end(), (const char *)p, (const char *)p + l_read_name - 1);
        b->name_off.push_back((int64_t)b->names.size());
        p += l_read_name;
        const size_t cig0 = b->cigar.size();
        if (!ids_only) {
            b->cigar.resize(cig0 + (size_t)n_cig);
            if (n_cig) memcpy(b->cigar.data() + cig0, p, 4 * (size_t)n_cig);  // (BAM is little-endian, as every host this builds for)
        }
        p += 4 * (size_t)n_cig;
        const size_t seq0 = b->seq.size();
        if (!ids_only) b->seq.resize(seq0 + (size_t)l_seq);
        if (!ids_only) {   // two bases per packed byte through a 256-entry table of character pairs
            static const std::array<uint16_t, 256> pair = [] {
                std::array<uint16_t, 256> t{};
                for (int v = 0; v < 256; ++v) t[(size_t)v] = (uint16_t)((uint8_t)NT16[v >> 4] | ((uint16_t)(uint8_t)NT16[v & 0xF] << 8));
                return t;
            }();
            char *dst = b->seq.data() + seq0;
            const int64_t full = l_seq >> 1;
            for (int64_t k = 0; k < full; ++k) memcpy(dst + 2 * k, &pair[p[k]], 2);
            if (l_seq & 1) dst[l_seq - 1] = NT16[p[full] >> 4];
        }
        b->seq_off.push_back((int64_t)b->seq.size());
        p += (l_seq + 1) / 2 + l_seq;  // packed bases + qualities
        b->tags_off.push_back((int64_t)(p - rec));
        // ---- tags ----
        uint8_t has = 0;
        int32_t ts = 0, ns = 0, sp = 0;
        float sm = 0.f, sd = 0.f;
        const char *md = nullptr;
        size_t md_len = 0;
        const uint8_t *cg = nullptr;  // CG:B,I — the real CIGAR of a record with more than 65535 operations
        int64_t cg_n = 0;
        while (p + 3 <= end) {
            const char t0 = (char)p[0], t1 = (char)p[1], ty = (char)p[2];
            const uint8_t *val = p + 3;
            const int64_t sz = tag_value_size(ty, val, end);
            if (sz < 0 || val + sz > end) RMR_FAIL(RMR_ERR_INVALID, "corrupt BAM tag %c%c", t0, t1);
            if (t0 == 'm' && t1 == 'v' && ty == 'B' && (val[0] == 'c' || val[0] == 'C')) {
                const int64_t cnt = rd_i32(val + 1);
                if (!ids_only) b->mv.insert(b->mv.end(), (const int8_t *)val + 5, (const int8_t *)val + 5 + cnt);
                has |= 1;
            } else if (t0 == 't' && t1 == 's' && tag_int(ty, val, &ts)) has |= 2;
            else if (t0 == 'n' && t1 == 's' && tag_int(ty, val, &ns)) has |= 4;
            else if (t0 == 's' && t1 == 'p' && tag_int(ty, val, &sp)) has |= 8;
            else if (t0 == 's' && t1 == 'm' && ty == 'f') { memcpy(&sm, val, 4); has |= 16; }
            else if (t0 == 's' && t1 == 'd' && ty == 'f') { memcpy(&sd, val, 4); has |= 32; }
            else if (t0 == 'p' && t1 == 'i' && ty == 'Z') { b->pi.insert(b->pi.end(), (const char *)val, (const char *)val + sz - 1); has |= 64; }
            else if (t0 == 'M' && t1 == 'D' && ty == 'Z') {
                md = (const char *)val; md_len = (size_t)sz - 1;
                if (!ids_only) b->md.insert(b->md.end(), md, md + md_len);
                has |= 128;
            } else if (t0 == 'C' && t1 == 'G' && ty == 'B' && val[0] == 'I') {
                cg = val + 5;
                cg_n = rd_i32(val + 1);
            }
            p = val + sz;
        }
        if (p != end) RMR_FAIL(RMR_ERR_INVALID, "corrupt BAM tag region");
        // SAM spec 4.2.2: a CIGAR that does not fit the 16-bit count is stored in CG and the record carries the
        // placeholder <l_seq>S<ref_len>N (htslib / pysam resolve this transparently)
        // (htslib's bam_tag2cigar looks at the first operation only, on mapped records)
        if (!ids_only && cg && cg_n > 0 && n_cig >= 1 && ref_id >= 0 && pos >= 0 && (b->cigar[cig0] & 0xF) == 4 &&
            (int64_t)(b->cigar[cig0] >> 4) == l_seq) {
            b->cigar.resize(cig0);
            for (int64_t k = 0; k < cg_n; ++k) b->cigar.push_back(rd_u32(cg + 4 * k));  // n_cigar keeps the record's own count (byte offsets)
        }
        b->cigar_off.push_back((int64_t)b->cigar.size());
        b->mv_off.push_back((int64_t)b->mv.size());
        b->pi_off.push_back((int64_t)b->pi.size());
        b->md_off.push_back((int64_t)b->md.size());
        b->ts.push_back(ts); b->ns.push_back(ns); b->sp.push_back(sp); b->sm.push_back(sm); b->sd.push_back(sd);
        b->has.push_back(has);
        uint8_t ok = 0;
        if ((want_ref & 1) && !ids_only && md && !(flag & 4))
            ok = rebuild_reference(b->seq.data() + seq0, (size_t)l_seq, b->cigar.data() + cig0, b->cigar.size() - cig0, md,
                                   md_len, cols, b->refseq) ? 1 : 0;
        b->ref_ok.push_back(ok);
        b->refseq_off.push_back((int64_t)b->refseq.size());
        b->upos += 4 + (size_t)bs;
        ++n;
    }
    out->n_records = n;
    out->flag = b->flag.data(); out->ref_id = b->ref_id.data(); out->pos = b->pos.data(); out->mapq = b->mapq.data();
    out->l_seq = b->l_seq.data(); out->n_cigar = b->n_cigar.data();
    out->raw_off = b->raw_off.data(); out->raw = b->raw.data();
    out->name_off = b->name_off.data(); out->names = b->names.data();
    out->seq_off = b->seq_off.data(); out->seq = b->seq.data();
    out->cigar_off = b->cigar_off.data(); out->cigar = b->cigar.data();
    out->tags_off = b->tags_off.data();
    out->has = b->has.data();
    out->mv_off = b->mv_off.data(); out->mv = b->mv.data();
    out->ts = b->ts.data(); out->ns = b->ns.data(); out->sp = b->sp.data(); out->sm = b->sm.data(); out->sd = b->sd.data();
    out->pi_off = b->pi_off.data(); out->pi = b->pi.data();
    out->md_off = b->md_off.data(); out->md = b->md.data();
    out->ref_ok = b->ref_ok.data(); out->refseq_off = b->refseq_off.data(); out->refseq = b->refseq.data();
    out->voffset = b->voff.data();
    return 0;
}

// Walk the rest of the file reading only the block_size fields: the number of records and the virtual offset of every
// `every`-th one (records 0, every, 2 every, ...), so that N workers can each seek to a contiguous share of the
// records without a coordinator handing them out (no tag, name or base is decoded: BGZF inflate + one add per record).
int rmr_bam_scan(rmr_bam *b, int64_t every, int64_t *voffsets, int64_t cap, int64_t *n_records) {
    if (!b || !n_records || every < 1 || cap < 0 || (cap > 0 && !voffsets)) RMR_FAIL(RMR_ERR_INVALID, "bad argument");
    int64_t n = 0;
    // boundaries only: the record bytes are verified by whoever seeks here and reads them (this handle included, later)
    struct NoCrc { rmr_bam *b; explicit NoCrc(rmr_bam *x) : b(x) { b->check_crc = false; } ~NoCrc() { b->check_crc = true; } } no_crc(b);
    for (;;) {
        int rc = ensure(b, 4);
        if (rc < 0) return rc;
        if (rc == 0) {
            if (b->ubuf.size() - b->upos != 0) RMR_FAIL(RMR_ERR_INVALID, "truncated BAM file");
            break;
        }
        const int64_t bs = rd_i32(b->ubuf.data() + b->upos);
        if (bs < 32) RMR_FAIL(RMR_ERR_INVALID, "corrupt BAM record");
        if (n % every == 0 && n / every < cap) {
            int64_t vo = -1;
            for (const auto &sg : b->segs) {
                const size_t rel = b->upos - sg.begin;
                if (rel < sg.isize) { vo = (sg.file_off << 16) | (int64_t)rel; break; }
            }
            voffsets[n / every] = vo;
        }
        rc = ensure(b, 4 + (size_t)bs);
        if (rc < 0) return rc;
        if (rc == 0) RMR_FAIL(RMR_ERR_INVALID, "truncated BAM file");
        b->upos += 4 + (size_t)bs;
        ++n;
    }
    *n_records = n;
    return 0;
}

// The virtual offset of the first record that starts in a BGZF member at or behind byte `file_offset` of the file
// (-1: none), found without the records in front of it: see plausible_record above.  Leaves the handle there.
int rmr_bam_guess_start(rmr_bam *b, int64_t file_offset, int64_t *voffset) {
    if (!b || !voffset || file_offset < 0) RMR_FAIL(RMR_ERR_INVALID, "bad argument");
    *voffset = -1;
    if (b->first_voffset < 0) return 0;
    if (file_offset <= (b->first_voffset >> 16)) {
        *voffset = b->first_voffset;
        return rmr_bam_seek(b, b->first_voffset);
    }
    drop_inflight(b);
    if (fseeko(b->fh, 0, SEEK_END) != 0) RMR_FAIL(RMR_ERR_INVALID, "seek failed");
    const int64_t fsize = (int64_t)ftello(b->fh);
    if (file_offset >= fsize) return 0;
    // the next member boundary: a header whose size leads to another header (or to the end of the file)
    std::vector<uint8_t> buf((size_t)std::min<int64_t>(fsize - file_offset, 3 * 65536 + 64));
    if (fseeko(b->fh, (off_t)file_offset, SEEK_SET) != 0 || fread(buf.data(), 1, buf.size(), b->fh) != buf.size())
        RMR_FAIL(RMR_ERR_INVALID, "read failed");
    int64_t member = -1;
    for (size_t i = 0; i < buf.size() && i <= 65536; ++i) {
        const int sz = member_size_at(buf.data(), buf.size(), i);
        if (!sz) continue;
        const size_t nx = i + (size_t)sz;
        if (file_offset + (int64_t)nx == fsize || member_size_at(buf.data(), buf.size(), nx)) { member = file_offset + (int64_t)i; break; }
    }
    if (member < 0) {
        if (file_offset + (int64_t)buf.size() == fsize && buf.size() <= 65536) return 0;  // inside the last member: nothing starts behind it
        RMR_FAIL(RMR_ERR_INVALID, "no BGZF block boundary within 64 KiB of offset %lld", (long long)file_offset);
    }
    if (fseeko(b->fh, (off_t)member, SEEK_SET) != 0) RMR_FAIL(RMR_ERR_INVALID, "seek failed");
    b->ubuf.clear();
    b->segs.clear();
    b->upos = 0;
    b->eof = false;
    const size_t limit = (size_t)1 << 29;  // a record and its chain within 512 MiB of inflated data, or the file is something else
    for (size_t p = 0;; ++p) {
        int rc = ensure(b, p + 1);
        if (rc < 0) return rc;
        if (rc == 0) return 0;  // nothing starts behind file_offset
        rc = record_chain_at(b, p);
        if (rc < 0) return rc;
        if (rc == 1) {
            *voffset = voffset_at(b, p);
            b->upos = p;
            return 0;
        }
        if (p > limit) RMR_FAIL(RMR_ERR_INVALID, "no alignment record found behind offset %lld", (long long)file_offset);
    }
}

int rmr_bam_seek(rmr_bam *b, int64_t voffset) {
    if (!b || voffset < 0) RMR_FAIL(RMR_ERR_INVALID, "bad argument");
    drop_inflight(b);
    if (fseeko(b->fh, (off_t)(voffset >> 16), SEEK_SET) != 0) RMR_FAIL(RMR_ERR_INVALID, "seek failed");
    b->ubuf.clear();
    b->segs.clear();
    b->upos = 0;
    b->eof = false;
    const size_t within = (size_t)(voffset & 0xFFFF);
    const int rc = ensure(b, within + 1);
    if (rc < 0) return rc;
    if (rc == 0) RMR_FAIL(RMR_ERR_INVALID, "virtual offset beyond the end of the file");
    b->upos = within;
    return 0;
}

}  // extern "C"

// one-shot inflate of a raw deflate stream (tests: the decoder alone, no zlib behind it)
extern "C" int rmr_inflate_raw(const uint8_t *src, int64_t n, uint8_t *out, int64_t out_len) {
    if ((!src && n > 0) || (!out && out_len > 0) || n < 0 || out_len < 0) return RMR_ERR_INVALID;
    std::vector<uint8_t> padded((size_t)n + 16, 0);
    if (n) memcpy(padded.data(), src, (size_t)n);
    std::unique_ptr<rmr_inflate::Tables> tb(new rmr_inflate::Tables);
    return rmr_inflate::inflate_raw(padded.data(), (size_t)n, out, (size_t)out_len, *tb) ? 0 : RMR_ERR_INVALID;
}
