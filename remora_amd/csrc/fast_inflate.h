// fast_inflate.h — raw DEFLATE (RFC 1951) decoder for BGZF members, host code: the inflate step of the native BAM reader
// (bam_reader.cpp; pysam / htslib do this for the reference, src/remora/io.py:184-358).
//
// Why: zlib 1.2.11 inflates BAM records at 180-230 MB/s, a quarter of the host time a record costs the file-to-file
// pipeline (profiles/NOTES_r04.md section 6).  A BGZF member is a complete stream of at most 64 KiB whose inflated size
// and CRC32 stand in its trailer, so a decoder for it can be one-shot: whole input and whole output in memory, a 64-bit
// bit buffer refilled eight bytes at a time, one table lookup per literal (11 primary bits, second-level tables for the
// longer codes), matches copied eight bytes at a time.  The reader checks the CRC32 of every member it inflates and hands
// any member this decoder refuses (or gets wrong) to zlib - so it only has to be fast on valid streams and safe on all.
//
// Supports everything the format has: stored, fixed and dynamic blocks, any number of blocks per member.
#pragma once
#include <cstdint>
#include <cstring>

namespace rmr_inflate {

constexpr int LIT_PRIMARY = 11, DIST_PRIMARY = 8;
constexpr int LIT_TABLE = (1 << LIT_PRIMARY) + 288 * 16, DIST_TABLE = (1 << DIST_PRIMARY) + 32 * 128;
enum Kind : uint32_t { LITERAL = 0, LENGTH = 1, END = 2, LINK = 3, INVALID = 4, DISTANCE = 5 };

// entry: bits 0-7 code bits consumed by this lookup, 8-11 kind, 12-15 extra bits (lengths / distances; LINK: bits of the
// second-level table), 16-31 payload (literal, base length, base distance, or first index of the second-level table;
// LITERAL: `extra` = how many literals the entry decodes - one, or two whose codes fit the primary bits together: the first in
// bits 16-23, the second in 24-31 - so that the decoder's loop has no branch on which of the two it met)
inline uint32_t entry(uint32_t nbits, uint32_t kind, uint32_t extra, uint32_t payload) {
    return nbits | (kind << 8) | (extra << 12) | (payload << 16);
}

struct Tables {
    uint32_t lit[LIT_TABLE];
    uint32_t dist[DIST_TABLE];
};

inline uint32_t reverse_bits(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

// decode table of a canonical Huffman code given its code lengths; `what` = what symbol s decodes to (an entry without its
// nbits).  Returns false for an over-subscribed code, or an incomplete one with more than one symbol.
template <typename F>
inline bool build_table(const uint8_t *len, int nsym, int primary, uint32_t *table, int table_cap, F what) {
    int count[16] = {0};
    for (int s = 0; s < nsym; ++s) ++count[len[s]];
    count[0] = 0;
    int used = 0, left = 1, maxlen = 0;
    for (int b = 1; b <= 15; ++b) {
        left = (left << 1) - count[b];
        if (left < 0) return false;  // over-subscribed
        used += count[b];
        if (count[b]) maxlen = b;
    }
    const int nprim = 1 << primary;
    for (int i = 0; i < nprim; ++i) table[i] = entry(0, INVALID, 0, 0);
    if (used == 0) return true;                 // no codes at all (a block of literals has no distance code): every lookup invalid
    if (left > 0 && used != 1) return false;    // incomplete (zlib accepts exactly one code of one bit)
    uint32_t next[16], code = 0;
    next[0] = 0;
    for (int b = 1; b <= 15; ++b) {
        code = (code + (uint32_t)count[b - 1]) << 1;
        next[b] = code;
    }
    // second-level tables: for every primary prefix the longest code below it decides the size
    int sub_bits[1 << LIT_PRIMARY];
    uint32_t rev[320];
    if (maxlen > primary) {
        for (int i = 0; i < nprim; ++i) sub_bits[i] = 0;
        uint32_t nx[16];
        memcpy(nx, next, sizeof(nx));
        for (int s = 0; s < nsym; ++s) {
            if (!len[s]) continue;
            const uint32_t r = reverse_bits(nx[len[s]]++, len[s]);
            rev[s] = r;
            if (len[s] > primary) {
                const int p = (int)(r & (uint32_t)(nprim - 1));
                if (len[s] - primary > sub_bits[p]) sub_bits[p] = len[s] - primary;
            }
        }
        int at = nprim;
        for (int p = 0; p < nprim; ++p)
            if (sub_bits[p]) {
                const int size = 1 << sub_bits[p];
                if (at + size > table_cap) return false;
                table[p] = entry((uint32_t)primary, LINK, (uint32_t)sub_bits[p], (uint32_t)at);
                for (int i = 0; i < size; ++i) table[at + i] = entry(0, INVALID, 0, 0);
                at += size;
            }
    } else {
        uint32_t nx[16];
        memcpy(nx, next, sizeof(nx));
        for (int s = 0; s < nsym; ++s)
            if (len[s]) rev[s] = reverse_bits(nx[len[s]]++, len[s]);
    }
    for (int s = 0; s < nsym; ++s) {
        const int L = len[s];
        if (!L) continue;
        const uint32_t r = rev[s], e = what(s);
        if (L <= primary) {
            for (uint32_t i = r; i < (uint32_t)nprim; i += 1u << L) table[i] = e | (uint32_t)L;
        } else {
            const uint32_t link = table[r & (uint32_t)(nprim - 1)];
            const uint32_t start = link >> 16, sb = (link >> 12) & 15u, hi = r >> primary;
            for (uint32_t i = hi; i < (1u << sb); i += 1u << (L - primary)) table[start + i] = e | (uint32_t)(L - primary);
        }
    }
    return true;
}

// Two literals per lookup where both codes fit the primary bits: BAM records code to under four bits per byte (qualities,
// move tables), and a decoder's speed is the latency of lookup -> shift -> lookup, so halving the lookups nearly halves the
// time.  An entry of a code of n2 bits is replicated over all values of the bits above it: lit[idx >> n1] IS the decode of
// what follows a first literal of n1 bits whenever n2 <= primary - n1.
inline void add_double_literals(uint32_t *lit) {
    uint32_t single[1 << LIT_PRIMARY];
    memcpy(single, lit, sizeof(single));
    for (uint32_t idx = 0; idx < (1u << LIT_PRIMARY); ++idx) {
        const uint32_t e1 = single[idx], n1 = e1 & 255;
        if (((e1 >> 8) & 15) != LITERAL || n1 == 0 || n1 >= (uint32_t)LIT_PRIMARY) continue;
        const uint32_t e2 = single[idx >> n1], n2 = e2 & 255;
        if (((e2 >> 8) & 15) != LITERAL || n2 == 0 || n1 + n2 > (uint32_t)LIT_PRIMARY) continue;
        lit[idx] = entry(n1 + n2, LITERAL, 2, ((e1 >> 16) & 255) | ((e2 >> 16) & 255) << 8);
    }
}

inline uint32_t litlen_entry(int s) {
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    if (s < 256) return entry(0, LITERAL, 1, (uint32_t)s);
    if (s == 256) return entry(0, END, 0, 0);
    if (s > 285) return entry(0, INVALID, 0, 0);
    return entry(0, LENGTH, extra[s - 257], base[s - 257]);
}
inline uint32_t dist_entry(int s) {
    static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (s > 29) return entry(0, INVALID, 0, 0);
    return entry(0, DISTANCE, extra[s], base[s]);
}

// Inflate src[0..n) into exactly out[0..out_len).  `src` must be readable for 16 bytes beyond n (padding, any content).
// Returns true when the stream is valid, ends inside src and fills the output exactly; false otherwise (the output may then
// hold anything up to out_len bytes - never more).
inline bool inflate_raw(const uint8_t *src, size_t n, uint8_t *out, size_t out_len, Tables &tb) {
    const uint8_t *in = src, *const in_end = src + n;
    uint8_t *o = out, *const o_end = out + out_len;
    uint64_t bits = 0;
    unsigned cnt = 0;
    // after refill at least 56 bits are in the buffer (zeros beyond the padding); a symbol needs at most 48
    auto refill = [&]() {
        uint64_t w;
        memcpy(&w, in, 8);
        bits |= w << cnt;
        in += (63 - cnt) >> 3;
        cnt |= 56;
    };
    // bytes actually consumed passed the end of the stream?  Checked in front of every refill: then `in` is at most 7 bytes
    // behind the end and the eight bytes a refill reads stay inside the padding
    auto overrun = [&]() { return in - (cnt >> 3) > in_end; };
    for (;;) {
        if (overrun()) return false;
        refill();
        const unsigned last = (unsigned)(bits & 1), type = (unsigned)((bits >> 1) & 3);
        bits >>= 3;
        cnt -= 3;
        if (type == 0) {  // stored: to the byte boundary, LEN, NLEN, bytes
            const unsigned drop = cnt & 7;
            bits >>= drop;
            cnt -= drop;
            in -= cnt >> 3;  // hand the whole bytes still in the buffer back
            bits = 0;
            cnt = 0;
            if (in + 4 > in_end) return false;
            const unsigned len = in[0] | (in[1] << 8), nlen = in[2] | (in[3] << 8);
            in += 4;
            if ((len ^ nlen) != 0xFFFFu || in + len > in_end || o + len > o_end) return false;
            if (len) memcpy(o, in, len);
            in += len;
            o += len;
        } else if (type == 1 || type == 2) {
            if (type == 1) {  // fixed code
                uint8_t ll[288], dl[32];
                for (int s = 0; s < 288; ++s) ll[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
                for (int s = 0; s < 32; ++s) dl[s] = 5;
                if (!build_table(ll, 288, LIT_PRIMARY, tb.lit, LIT_TABLE, litlen_entry)) return false;
                add_double_literals(tb.lit);
                if (!build_table(dl, 32, DIST_PRIMARY, tb.dist, DIST_TABLE, dist_entry)) return false;
            } else {          // dynamic code
                const unsigned hlit = (unsigned)(bits & 31) + 257, hdist = (unsigned)((bits >> 5) & 31) + 1, hclen = (unsigned)((bits >> 10) & 15) + 4;
                bits >>= 14;
                cnt -= 14;
                if (hlit > 286 || hdist > 30) return false;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (unsigned i = 0; i < hclen; ++i) {
                    if (cnt < 3) {
                        if (overrun()) return false;
                        refill();
                    }
                    cl[order[i]] = (uint8_t)(bits & 7);
                    bits >>= 3;
                    cnt -= 3;
                }
                uint32_t cltab[128];
                if (!build_table(cl, 19, 7, cltab, 128, [](int s) { return entry(0, LITERAL, 0, (uint32_t)s); })) return false;
                uint8_t lens[286 + 30 + 16] = {0};
                unsigned i = 0;
                while (i < hlit + hdist) {
                    if (overrun()) return false;
                    refill();
                    const uint32_t e = cltab[bits & 127];
                    if (((e >> 8) & 15) != LITERAL || (e & 255) == 0) return false;
                    bits >>= (e & 255);
                    cnt -= (e & 255);
                    const unsigned sym = e >> 16;
                    if (sym < 16) {
                        lens[i++] = (uint8_t)sym;
                    } else {
                        unsigned rep, val = 0;
                        if (sym == 16) {
                            if (i == 0) return false;
                            val = lens[i - 1];
                            rep = 3 + (unsigned)(bits & 3);
                            bits >>= 2;
                            cnt -= 2;
                        } else if (sym == 17) {
                            rep = 3 + (unsigned)(bits & 7);
                            bits >>= 3;
                            cnt -= 3;
                        } else {
                            rep = 11 + (unsigned)(bits & 127);
                            bits >>= 7;
                            cnt -= 7;
                        }
                        if (i + rep > hlit + hdist) return false;
                        while (rep--) lens[i++] = (uint8_t)val;
                    }
                }
                if (lens[256] == 0) return false;  // no end-of-block code
                if (!build_table(lens, (int)hlit, LIT_PRIMARY, tb.lit, LIT_TABLE, litlen_entry)) return false;
                add_double_literals(tb.lit);
                if (!build_table(lens + hlit, (int)hdist, DIST_PRIMARY, tb.dist, DIST_TABLE, dist_entry)) return false;
            }
            // ---- the block's symbols ----
            for (;;) {
                if (overrun()) return false;
                refill();
                // up to four lookups of literals out of one refill (each consumes at most the primary bits); one or two
                // literals per lookup, written as two bytes either way (no branch on the data)
                uint32_t e = tb.lit[bits & ((1u << LIT_PRIMARY) - 1)];
                int quick = 0;
                for (; quick < 4; ++quick) {
                    if (((e >> 8) & 15) != LITERAL || (e & 255) == 0) break;
                    const unsigned nlit = (e >> 12) & 15;
                    if (o_end - o >= 2) {
                        o[0] = (uint8_t)(e >> 16);
                        o[1] = (uint8_t)(e >> 24);
                    } else {  // the last byte of the member
                        if (nlit > (unsigned)(o_end - o)) return false;
                        o[0] = (uint8_t)(e >> 16);
                    }
                    o += nlit;
                    bits >>= (e & 255);
                    cnt -= (e & 255);
                    e = tb.lit[bits & ((1u << LIT_PRIMARY) - 1)];
                }
                if (quick == 4) continue;
                if (quick) {  // something else than a literal follows: it may need 48 bits
                    if (overrun()) return false;
                    refill();
                    e = tb.lit[bits & ((1u << LIT_PRIMARY) - 1)];
                }
                if (((e >> 8) & 15) == LINK) {
                    bits >>= LIT_PRIMARY;
                    cnt -= LIT_PRIMARY;
                    e = tb.lit[(e >> 16) + (bits & ((1u << ((e >> 12) & 15)) - 1))];
                }
                const unsigned kind = (e >> 8) & 15;
                bits >>= (e & 255);
                cnt -= (e & 255);
                if (kind == LITERAL) {  // (a literal with a long code, out of a second-level table)
                    if (o >= o_end || (e & 255) == 0) return false;
                    *o++ = (uint8_t)(e >> 16);
                    continue;
                }
                if (kind == END) break;
                if (kind != LENGTH || (e & 255) == 0) return false;
                const unsigned lx = (e >> 12) & 15;
                const unsigned length = (e >> 16) + (unsigned)(bits & ((1u << lx) - 1));
                bits >>= lx;
                cnt -= lx;
                uint32_t d = tb.dist[bits & ((1u << DIST_PRIMARY) - 1)];
                if (((d >> 8) & 15) == LINK) {
                    bits >>= DIST_PRIMARY;
                    cnt -= DIST_PRIMARY;
                    d = tb.dist[(d >> 16) + (bits & ((1u << ((d >> 12) & 15)) - 1))];
                }
                if (((d >> 8) & 15) != DISTANCE || (d & 255) == 0) return false;
                bits >>= (d & 255);
                cnt -= (d & 255);
                const unsigned dx = (d >> 12) & 15;
                const size_t dist = (d >> 16) + (size_t)(bits & ((1u << dx) - 1));
                bits >>= dx;
                cnt -= dx;
                if (dist > (size_t)(o - out) || length > (size_t)(o_end - o)) return false;
                const uint8_t *from = o - dist;
                if (dist >= 8 && (size_t)(o_end - o) >= length + 8) {  // eight bytes at a time (may write up to 7 bytes beyond: inside the output)
                    uint8_t *to = o;
                    for (unsigned k = 0; k < length; k += 8) {
                        uint64_t w;
                        memcpy(&w, from + k, 8);
                        memcpy(to + k, &w, 8);
                    }
                } else {
                    for (unsigned k = 0; k < length; ++k) o[k] = from[k];  // overlapping runs replicate byte by byte
                }
                o += length;
            }
        } else {
            return false;  // block type 3
        }
        if (last) break;
    }
    if (overrun()) return false;
    return o == o_end;
}

}  // namespace rmr_inflate
