// bgzf_deflate.cpp — BGZF members with Huffman-only deflate, host code (N3: the output side of `infer from_pod5_and_bam`).
// replaces: what pysam / htslib do below AlignmentFile.write for the reference (src/remora/inference.py:619-623); here the
// writer's deflate pool (remora_amd/io.py, BamWriter) at `--bam-level 1`.
//
// Why its own encoder: file-to-file inference is CPU-bound on 16 cores, and of the 0.3 ms a record costs the host two
// thirds are zlib (1.2.11, no SIMD): inflate of the input, deflate of the output.  At level 1 the writer uses Huffman
// coding only (no LZ77 matches: move tables and qualities barely repeat) - for which zlib still runs its generic deflate
// loop at ~120 MB/s.  A Huffman-only encoder is a histogram, a code construction and a table-driven bit packer.
//
// Format (RFC 1951 / SAM spec 4.1): per <= 0xFF00 bytes of payload one gzip member with the BC extra field; its deflate
// stream is ONE final block: dynamic Huffman (BTYPE 10) over the 256 literals + end-of-block, no distance codes (HDIST = 0
// with one code of zero bits = "all literals"), code lengths sent with the plain 4-bit code-length code (symbols 0..15,
// complete; no run-length symbols: 129 bytes of header per block, 0.2 %); a block that would not shrink is stored
// (BTYPE 00).  Any inflater reads it; the bytes differ from zlib's Z_HUFFMAN_ONLY output (another, equally valid code).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/remora_hip.h"
#include "crc32_fast.h"

namespace {

constexpr int NSYM = 257;       // literals 0..255 + end of block (256)
constexpr int MAXBITS = 15;
constexpr size_t BLOCK = 0xFF00;  // payload per member, as the Python writer cuts it
constexpr size_t MEMBER_MAX = 18 + 5 + BLOCK + 8;  // a stored block: the largest member

// code lengths (<= MAXBITS) of a Huffman code for freq[0..NSYM); symbols of frequency 0 get length 0
void code_lengths(const uint32_t *freq_in, uint8_t *len) {
    uint32_t freq[NSYM];
    memcpy(freq, freq_in, sizeof(freq));
    for (;;) {
        // two-queue construction over the symbols sorted by frequency
        struct Node { uint64_t w; int left, right; };
        std::vector<int> order;
        for (int s = 0; s < NSYM; ++s)
            if (freq[s]) order.push_back(s);
        if (order.size() == 1) {  // a code needs two leaves to be complete: the lone symbol gets one bit, so does a dummy
            memset(len, 0, NSYM);
            len[order[0]] = 1;
            len[order[0] == 0 ? 1 : 0] = 1;
            return;
        }
        std::sort(order.begin(), order.end(), [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
        const int n = (int)order.size();
        std::vector<Node> nodes(2 * n - 1);
        for (int i = 0; i < n; ++i) nodes[i] = {freq[order[i]], -1, -1};
        int leaf = 0, inner = n, made = n;
        auto take = [&]() {
            if (leaf < n && (inner >= made || nodes[leaf].w <= nodes[inner].w)) return leaf++;
            return inner++;
        };
        while (made < 2 * n - 1) {
            const int a = take(), b = take();
            nodes[made] = {nodes[a].w + nodes[b].w, a, b};
            ++made;
        }
        // depths: children are created before their parents, so one backward pass from the root
        std::vector<int> depth(2 * n - 1, 0);
        int maxd = 0;
        for (int i = 2 * n - 2; i >= n; --i) {
            depth[nodes[i].left] = depth[nodes[i].right] = depth[i] + 1;
        }
        memset(len, 0, NSYM);
        for (int i = 0; i < n; ++i) {
            len[order[i]] = (uint8_t)depth[i];
            maxd = std::max(maxd, depth[i]);
        }
        if (maxd <= MAXBITS) return;
        for (int s = 0; s < NSYM; ++s)  // too deep (a very skewed block): flatten the histogram and build again
            if (freq[s]) freq[s] = std::max<uint32_t>(1, freq[s] >> 2);
    }
}

inline uint32_t bit_reverse(uint32_t v, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
    return r;
}

struct BitWriter {
    uint8_t *p;
    uint64_t acc = 0;
    int n = 0;
    explicit BitWriter(uint8_t *out) : p(out) {}
    inline void put(uint32_t bits, int count) {  // LSB first
        acc |= (uint64_t)bits << n;
        n += count;
        if (n >= 32) {
            memcpy(p, &acc, 4);
            p += 4;
            acc >>= 32;
            n -= 32;
        }
    }
    uint8_t *finish() {
        while (n > 0) {
            *p++ = (uint8_t)acc;
            acc >>= 8;
            n -= 8;
        }
        return p;
    }
};

// one BGZF member for src[0..n) (n <= BLOCK) into out (>= MEMBER_MAX bytes); returns its size
size_t bgzf_member(const uint8_t *src, size_t n, uint8_t *out) {
    static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(out, head, 16);
    uint8_t *body = out + 18, *end = nullptr;
    uint32_t freq[NSYM] = {0};
    {   // four histograms side by side: runs of equal bytes (qualities, move tables) would otherwise serialise on one counter
        uint32_t h[4][256] = {{0}};
        size_t i = 0;
        for (; i + 4 <= n; i += 4) {
            ++h[0][src[i]];
            ++h[1][src[i + 1]];
            ++h[2][src[i + 2]];
            ++h[3][src[i + 3]];
        }
        for (; i < n; ++i) ++h[0][src[i]];
        for (int s = 0; s < 256; ++s) freq[s] = h[0][s] + h[1][s] + h[2][s] + h[3][s];
    }
    freq[256] = 1;
    uint8_t len[NSYM];
    code_lengths(freq, len);
    // estimated size of the dynamic block: 17 bits of block header + 19 x 3 + 258 x 4 bits of lengths + the payload
    uint64_t bits = 3 + 5 + 5 + 4 + 19 * 3 + 258 * 4;
    for (int s = 0; s < NSYM; ++s) bits += (uint64_t)freq[s] * len[s];
    if ((bits + 7) / 8 >= n + 5) {  // would not shrink: stored block
        body[0] = 1;                // BFINAL = 1, BTYPE = 00 (rest of the byte is padding)
        body[1] = (uint8_t)(n & 0xff);
        body[2] = (uint8_t)(n >> 8);
        body[3] = (uint8_t)(~n & 0xff);
        body[4] = (uint8_t)((~n >> 8) & 0xff);
        if (n) memcpy(body + 5, src, n);
        end = body + 5 + n;
    } else {
        // canonical codes (RFC 1951 3.2.2), stored bit-reversed: deflate packs Huffman codes most significant bit first
        uint32_t count[MAXBITS + 1] = {0};
        for (int s = 0; s < NSYM; ++s) ++count[len[s]];
        count[0] = 0;
        uint32_t code = 0;
        uint32_t first[MAXBITS + 1] = {0};
        for (int b = 1; b <= MAXBITS; ++b) {
            code = (code + count[b - 1]) << 1;
            first[b] = code;
        }
        uint32_t cw[NSYM];
        for (int s = 0; s < NSYM; ++s) cw[s] = len[s] ? bit_reverse(first[len[s]]++, len[s]) : 0;
        BitWriter bw(body);
        bw.put(1, 1);        // BFINAL
        bw.put(2, 2);        // BTYPE = 10: dynamic Huffman
        bw.put(0, 5);        // HLIT: 257 literal / length codes
        bw.put(0, 5);        // HDIST: 1 distance code (of zero bits: no distances)
        bw.put(15, 4);       // HCLEN: all 19 code-length code lengths follow
        static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int i = 0; i < 19; ++i) bw.put(order[i] < 16 ? 4 : 0, 3);  // symbols 0..15: 4 bits each (a complete code), 16-18 unused
        for (int s = 0; s < NSYM; ++s) bw.put(bit_reverse(len[s], 4), 4);  // the code of length value v is v itself
        bw.put(bit_reverse(0, 4), 4);                                      // the one distance code: length 0
        // code and length of a literal in one word; the bit buffer takes two literals (<= 30 bits) per flush test
        uint32_t lit[256];
        for (int s = 0; s < 256; ++s) lit[s] = cw[s] | ((uint32_t)len[s] << 16);
        size_t i = 0;
        for (; i + 2 <= n; i += 2) {
            const uint32_t a = lit[src[i]], b = lit[src[i + 1]];
            const unsigned la = a >> 16, lb = b >> 16;
            bw.put((a & 0xFFFFu) | ((b & 0xFFFFu) << la), la + lb);
        }
        if (i < n) bw.put(cw[src[i]], len[src[i]]);
        bw.put(cw[256], len[256]);
        end = bw.finish();
    }
    const uint32_t crc = rmr_crc::crc32(src, n), isize = (uint32_t)n;
    memcpy(end, &crc, 4);
    memcpy(end + 4, &isize, 4);
    const size_t total = (size_t)(end + 8 - out);
    const uint16_t bsize = (uint16_t)(total - 1);
    memcpy(out + 16, &bsize, 2);
    return total;
}

}  // namespace

extern "C" {

int rmr_bgzf_huffman(const uint8_t *src, int64_t n, int n_threads, uint8_t *out, int64_t out_cap, int64_t *out_len) {
    if ((!src && n > 0) || !out || !out_len || n < 0) return RMR_ERR_INVALID;
    const int64_t n_blocks = (n + (int64_t)BLOCK - 1) / (int64_t)BLOCK;
    if (out_cap < n_blocks * (int64_t)MEMBER_MAX) return RMR_ERR_INVALID;
    *out_len = 0;
    if (n_blocks == 0) return 0;
    // every block into its own MEMBER_MAX slot (threads share nothing), then the members are moved together
    std::vector<size_t> size((size_t)n_blocks);
    auto work = [&](int64_t b0, int64_t b1) {
        for (int64_t b = b0; b < b1; ++b) {
            const size_t off = (size_t)b * BLOCK, len = std::min(BLOCK, (size_t)n - off);
            size[(size_t)b] = bgzf_member(src + off, len, out + (size_t)b * MEMBER_MAX);
        }
    };
    int nt = std::max(1, std::min<int>(n_threads, (int)std::min<int64_t>(n_blocks, 64)));
    if (nt == 1) {
        work(0, n_blocks);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(work, n_blocks * t / nt, n_blocks * (t + 1) / nt);
        for (auto &x : th) x.join();
    }
    size_t pos = size[0];
    for (int64_t b = 1; b < n_blocks; ++b) {
        memmove(out + pos, out + (size_t)b * MEMBER_MAX, size[(size_t)b]);
        pos += size[(size_t)b];
    }
    *out_len = (int64_t)pos;
    return 0;
}

}  // extern "C"
