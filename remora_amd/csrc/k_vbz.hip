// N1: POD5 signal decompression on the GPU — the VBZ layer below zstd:
// streamvbyte16 (one key bit per value, keys first: 0 = one data byte, 1 = two) -> zigzag -> running sum (int16).
// replaces the per-row decode that pod5's C++ library performs for `pod5.Reader` / `ReadRecord.signal`, which the
// reference consumes in io.iter_signal (src/remora/io.py:441-474) and Read.from_pod5_and_alignment (:2086-2121).
// zstd itself stays on the host (libzstd); its output - 1..2 bytes per sample - is what crosses PCIe.
//
// One 64-lane wave per signal row (4 rows per workgroup), steps of 1024 samples: lane l owns two key bytes of the
// step (16 samples).
//   bytes consumed by a lane = 16 + popcount(keys)  -> wave exclusive scan (shuffles) -> its offset in the data stream
//   the step's data bytes are staged in the wave's LDS slice as aligned dwords (coalesced), decoded per lane from LDS
//   per-lane running sum of its 16 deltas -> wave exclusive scan (mod 2^16) + carry from the previous step
// No workgroup barrier anywhere: a row advances at the pace of its own wave.
// HBM traffic: 1.0-2.1 B read and 2 B written per sample; no other bound applies.
#include "rmr_internal.h"

namespace rmr {
namespace {

constexpr int kVbzWaves = 4;          // rows per workgroup
constexpr int kVbzPer = 16;           // samples per lane and step (a multiple of 8, at most 32)
constexpr int kVbzStep = 64 * kVbzPer;

// exclusive scan over the 64 lanes with DPP row shifts / row broadcasts (no LDS round trips); *total receives the sum
__device__ __forceinline__ int wave_exscan(int v, int *total) {
    int inc = v;
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, false);  // row_shr:1
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, false);  // row_shr:2
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, false);  // row_shr:4
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, false);  // row_shr:8
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    *total = __builtin_amdgcn_readlane(inc, 63);
    return inc - v;
}

__global__ __launch_bounds__(64 * kVbzWaves) void vbz_decode_kernel(const uint8_t *__restrict__ svb,
                                                                    const int64_t *__restrict__ row_off,
                                                                    const int32_t *__restrict__ row_n,
                                                                    const int64_t *__restrict__ out_off, int64_t n_rows,
                                                                    int16_t *__restrict__ out, int32_t *__restrict__ status) {
    __shared__ uint32_t sdata_all[kVbzWaves][kVbzStep * 2 / 4 + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * kVbzWaves + wave;
    if (row >= n_rows) return;
    uint32_t *sdata = sdata_all[wave];
    const int n = row_n[row];
    if (n <= 0) return;
    const int64_t nkeys = ((int64_t)n + 7) / 8;
    const uint8_t *keys = svb + row_off[row];
    const uint8_t *data = keys + nkeys;
    const int64_t avail = row_off[row + 1] - row_off[row] - nkeys;  // data bytes present
    int16_t *dst = out + out_off[row];
    int64_t data_pos = 0;
    uint32_t carry = 0;  // running sum of the previous steps, mod 2^16
    constexpr int kStage = (kVbzStep * 2 / 4 + 4 + 63) / 64;  // dwords a lane stages per step

    // keys of a step for this lane, and what they imply (number of samples, data bytes)
    auto load_keys = [&](int64_t s0, uint32_t *key, int *nv) {
        const int64_t first = s0 + (int64_t)lane * kVbzPer;
        *nv = (int)min((int64_t)kVbzPer, max((int64_t)0, (int64_t)n - first));
        uint32_t k = 0;
#pragma unroll
        for (int b = 0; b < kVbzPer / 8; ++b)
            if (*nv > 8 * b) k |= (uint32_t)keys[(first >> 3) + b] << (8 * b);
        k &= (*nv >= 32) ? 0xffffffffu : ((1u << *nv) - 1u);
        *key = k;
    };
    // the data bytes of a step, as aligned dwords in registers (they go to LDS when the slice is free)
    auto load_data = [&](int64_t pos, int total, uint32_t *regs, int *mis) {
        const uint8_t *src = data + pos;
        *mis = (int)((uintptr_t)src & 3);
        const uint32_t *src4 = reinterpret_cast<const uint32_t *>(src - *mis);
        const int nwords = (*mis + total + 3) >> 2;
#pragma unroll
        for (int i = 0; i < kStage; ++i) regs[i] = (lane + 64 * i < nwords) ? src4[lane + 64 * i] : 0u;
    };

    // software pipeline: while step k is decoded from LDS, the keys of step k+2 and the data of step k+1 are in flight
    uint32_t key_c, key_n = 0;
    int nv_c, nv_n = 0, off_c, total_c, mis_c, off_n = 0, total_n = 0, mis_n = 0;
    uint32_t regs[kStage];
    load_keys(0, &key_c, &nv_c);
    off_c = wave_exscan(nv_c + __popc(key_c), &total_c);
    if (total_c > avail) { if (lane == 0) status[row] = 1; return; }
    load_data(0, total_c, regs, &mis_c);
    if (kVbzStep < n) load_keys(kVbzStep, &key_n, &nv_n);
    for (int64_t s0 = 0; s0 < n; s0 += kVbzStep) {
        // data of the current step: registers -> LDS
#pragma unroll
        for (int i = 0; i < kStage; ++i)
            if (lane + 64 * i < kVbzStep * 2 / 4 + 4) sdata[lane + 64 * i] = regs[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // next step: its offsets are known from its keys alone, so its data can be requested now
        const bool more = s0 + kVbzStep < n;
        uint32_t key_nn = 0;
        int nv_nn = 0;
        if (more) {
            off_n = wave_exscan(nv_n + __popc(key_n), &total_n);
            if (data_pos + total_c + total_n > avail) { if (lane == 0) status[row] = 1; return; }
            load_data(data_pos + total_c, total_n, regs, &mis_n);
            if (s0 + 2 * kVbzStep < n) load_keys(s0 + 2 * kVbzStep, &key_nn, &nv_nn);
        }
        const int64_t first = s0 + (int64_t)lane * kVbzPer;
        const uint8_t *sb = reinterpret_cast<const uint8_t *>(sdata) + mis_c + off_c;
        uint32_t acc = 0;
        uint16_t vals[kVbzPer];
        uint32_t raw[kVbzPer];
#pragma unroll
        for (int j = 0; j < kVbzPer; ++j) {  // byte offset of sample j = j + (two-byte samples before it): independent reads
            const int pj = j + __popc(key_c & ((1u << j) - 1u));
            const uint32_t b0 = sb[pj], b1 = sb[pj + 1];  // byte reads: unaligned 16-bit LDS reads measured 2x slower
            raw[j] = ((key_c >> j) & 1u) ? (b0 | (b1 << 8)) : b0;
        }
#pragma unroll
        for (int j = 0; j < kVbzPer; ++j) {
            const uint32_t v = (j < nv_c) ? raw[j] : 0u;
            acc += (v >> 1) ^ (0u - (v & 1u));  // zigzag
            vals[j] = (uint16_t)acc;
        }
        int tsum;
        const int base = wave_exscan((int)(acc & 0xffffu), &tsum);
        const uint32_t add = carry + (uint32_t)base;
        int16_t *o = dst + first;
        if (nv_c == kVbzPer && ((uintptr_t)o & 15) == 0) {
            uint4 w[kVbzPer / 8];
            uint32_t *wp = reinterpret_cast<uint32_t *>(w);
#pragma unroll
            for (int j = 0; j < kVbzPer; j += 2)
                wp[j >> 1] = ((uint32_t)(uint16_t)(vals[j] + add)) | ((uint32_t)(uint16_t)(vals[j + 1] + add) << 16);
#pragma unroll
            for (int q = 0; q < kVbzPer / 8; ++q) reinterpret_cast<uint4 *>(o)[q] = w[q];
        } else {
#pragma unroll
            for (int j = 0; j < kVbzPer; ++j)
                if (j < nv_c) o[j] = (int16_t)(uint16_t)(vals[j] + add);
        }
        carry = (carry + (uint32_t)tsum) & 0xffffu;
        data_pos += total_c;
        key_c = key_n; nv_c = nv_n; off_c = off_n; total_c = total_n; mis_c = mis_n;
        key_n = key_nn; nv_n = nv_nn;
        __builtin_amdgcn_wave_barrier();  // the LDS slice is refilled by the next step
    }
    if (lane == 0 && data_pos != avail) status[row] = 1;  // trailing garbage
}

}  // namespace

int launch_vbz(rmr_engine *e, const uint8_t *svb, const int64_t *row_off, const int32_t *row_n, const int64_t *out_off,
               int64_t n_rows, int16_t *out, int32_t *status) {
    if (n_rows <= 0) return 0;
    ProfScope ps(e, K_VBZ);
    hipLaunchKernelGGL(vbz_decode_kernel, dim3((unsigned)((n_rows + kVbzWaves - 1) / kVbzWaves)), dim3(64 * kVbzWaves), 0,
                       e->stream, svb, row_off, row_n, out_off, n_rows, out, status);
    RMR_HIP(hipGetLastError());
    return 0;
}

}  // namespace rmr
