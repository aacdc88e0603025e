// crc32_fast.h — the gzip CRC-32 (polynomial 0xEDB88320, as zlib's crc32) sixteen bytes per step with sixteen 256-entry tables
// ("slicing"): zlib 1.2.11 does about 1 GB/s here, a BGZF member is checksummed once when it is read and once when it is
// written, and that was a sixth of the host time a record costs (profiles/NOTES_r04.md section 6).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace rmr_crc {

struct Tables {
    uint32_t t[16][256];
    Tables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 16; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
    }
};

inline uint32_t crc32(const uint8_t *p, size_t n, uint32_t crc = 0) {
    static const Tables T;
    const uint32_t(*t)[256] = T.t;
    uint32_t c = ~crc;
    while (n >= 16) {
        uint32_t a, b, d, e;
        memcpy(&a, p, 4);
        memcpy(&b, p + 4, 4);
        memcpy(&d, p + 8, 4);
        memcpy(&e, p + 12, 4);
        a ^= c;
        c = t[15][a & 0xFF] ^ t[14][(a >> 8) & 0xFF] ^ t[13][(a >> 16) & 0xFF] ^ t[12][a >> 24] ^
            t[11][b & 0xFF] ^ t[10][(b >> 8) & 0xFF] ^ t[9][(b >> 16) & 0xFF] ^ t[8][b >> 24] ^
            t[7][d & 0xFF] ^ t[6][(d >> 8) & 0xFF] ^ t[5][(d >> 16) & 0xFF] ^ t[4][d >> 24] ^
            t[3][e & 0xFF] ^ t[2][(e >> 8) & 0xFF] ^ t[1][(e >> 16) & 0xFF] ^ t[0][e >> 24];
        p += 16;
        n -= 16;
    }
    while (n--) c = t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return ~c;
}

}  // namespace rmr_crc
