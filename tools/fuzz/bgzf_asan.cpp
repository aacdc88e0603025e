#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cstring>
#include <zlib.h>
extern "C" int rmr_bgzf_huffman(const uint8_t *src, int64_t n, int n_threads, uint8_t *out, int64_t out_cap, int64_t *out_len);
int main() {
    std::mt19937 rng(3);
    for (int round = 0; round < 300; ++round) {
        size_t n = rng() % 200000; if (round < 5) n = round;
        std::vector<uint8_t> raw(n ? n : 1);
        int kind = rng() % 5;
        for (size_t i = 0; i < n; ++i) raw[i] = kind == 0 ? rng() : kind == 1 ? rng() % 3 : kind == 2 ? 7 : kind == 3 ? (uint8_t)(__builtin_ctz(rng() | 0x1000000) * 11) : (uint8_t)(i * i >> 5);
        size_t nb = (n + 0xFEFF) / 0xFF00; std::vector<uint8_t> out(nb ? nb * 65311 : 1); int64_t ol = -1;
        if (rmr_bgzf_huffman(raw.data(), (int64_t)n, 1 + rng() % 4, out.data(), (int64_t)(nb * 65311), &ol) != 0) { printf("rc\n"); return 1; }
        // inflate every member with zlib and compare
        size_t pos = 0, got = 0;
        while (pos < (size_t)ol) {
            size_t bs = (out[pos + 16] | (out[pos + 17] << 8)) + 1; if (bs > 65536) { printf("member too large\n"); return 1; }
            std::vector<uint8_t> o(0xFF00 + 1); z_stream s{}; inflateInit2(&s, -15); s.next_in = out.data() + pos + 18; s.avail_in = bs - 26; s.next_out = o.data(); s.avail_out = o.size();
            int rc = inflate(&s, Z_FINISH); size_t len = s.total_out; inflateEnd(&s);
            uint32_t crc, isz; memcpy(&crc, out.data() + pos + bs - 8, 4); memcpy(&isz, out.data() + pos + bs - 4, 4);
            if (rc != Z_STREAM_END || isz != len || crc != (uint32_t)crc32(crc32(0, Z_NULL, 0), o.data(), len) || memcmp(o.data(), raw.data() + got, len)) { printf("round %d bad member at %zu rc %d\n", round, pos, rc); return 1; }
            got += len; pos += bs;
        }
        if (got != n) { printf("size mismatch\n"); return 1; }
    }
    printf("300 payloads: every member inflates with zlib to the input, CRC32 and ISIZE right\n");
}
