#include "../../remora_amd/csrc/fast_inflate.h"
#include <zlib.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <memory>
#include <random>
#include <algorithm>
// exact-size heap buffers (ASAN catches any read beyond src + n + 16 and any write beyond out + out_len)
int main() {
    std::mt19937 rng(7);
    std::unique_ptr<rmr_inflate::Tables> tb(new rmr_inflate::Tables);
    long ok = 0, refused = 0, total = 0;
    for (int round = 0; round < 400; ++round) {
        size_t n = 1 + rng() % 70000;
        std::vector<uint8_t> raw(n);
        int kind = rng() % 5;
        for (size_t i = 0; i < n; ++i) raw[i] = kind == 0 ? rng() : kind == 1 ? rng() % 4 : kind == 2 ? (uint8_t)(i / 7) : kind == 3 ? (uint8_t)std::min<unsigned>(255, __builtin_ctz(rng() | 0x10000) * 9) : (uint8_t)"ACGT"[rng() % 4];
        int level = rng() % 10, strat = rng() % 5;
        z_stream s{}; deflateInit2(&s, level, Z_DEFLATED, -15, 9, strat);
        std::vector<uint8_t> z(n + n / 2 + 1024); s.next_in = raw.data(); s.avail_in = n; s.next_out = z.data(); s.avail_out = z.size();
        deflate(&s, Z_FINISH); size_t zn = s.total_out; deflateEnd(&s);
        for (int m = 0; m < 30; ++m) {
            size_t use = zn; std::vector<uint8_t> src(z.begin(), z.begin() + zn);
            if (m > 0) {
                int what = rng() % 4;
                if (what == 0) src[rng() % zn] ^= 1u << (rng() % 8);
                else if (what == 1) { use = rng() % (zn + 1); src.resize(use); }
                else if (what == 2) for (int k = 0; k < 8; ++k) src[rng() % zn] = rng();
                else { size_t a = rng() % zn; std::fill(src.begin() + a, src.end(), (uint8_t)rng()); }
            }
            std::vector<uint8_t> padded(use + 16, 0xAA); std::copy(src.begin(), src.begin() + use, padded.begin());
            size_t out_len = m % 5 == 4 ? (n ? n - 1 : 0) : n;
            std::vector<uint8_t> out(out_len ? out_len : 1);
            bool r = rmr_inflate::inflate_raw(padded.data(), use, out.data(), out_len, *tb);
            ++total;
            if (m == 0) { if (!r || !std::equal(raw.begin(), raw.end(), out.begin())) { printf("VALID STREAM FAILED round %d\n", round); return 1; } ++ok; }
            else if (!r) ++refused;
        }
    }
    printf("%ld calls: %ld valid streams decoded, %ld malformed refused, the rest decoded to the right size (CRC would catch them)\n", total, ok, refused);
}
