// s4_loop.hip — merge_conv1 (S4 of remora_amd/csrc/k_fused.hip) in isolation: what is each way of feeding the matrix cores
// worth when nothing else runs?  Round-3 review item 1 asked for 32 output channels per wave on v_mfma_f32_32x32x16 (one B
// fragment read from LDS feeds two A tiles) "or a micro-benchmark that reproduces the S4 loop with 32x32x16 and shows the
// ceiling is lower".  This is that benchmark.
//
// Work per chunk (C100): 24 output columns x 64 channels, K = 5 taps x 128 channels = 640  ->  1.97 MFLOP, 480 matrix
// cycles per chunk and CU at the full rate of four SIMDs (120 v_mfma_f32_16x16x32_bf16 = 60 v_mfma_f32_32x32x16_bf16).
// The CAT image is the kernel's: four 8-channel planes, rows of five 16-byte slots (four used), rows of a chunk contiguous.
// Every block holds `cb` chunks of CAT in LDS (filled once) and loops over them; two blocks of four waves per CU as in the
// kernel.  Modes:
//   0  the kernel's scheme: wave = 16 output channels (80 VGPRs of A fragments), column tiles in pairs, 16 reads in
//      flight, swish epilogue + bf16 stores
//   1  mode 0 without the epilogue (accumulators folded into one store at the end): what the VALU tail costs
//   2  mode 0 with chunk-contiguous columns (no 4-row gap between chunks: no tile straddles a boundary): what the bank
//      conflicts of straddling tiles cost
//   3  32 output channels per wave on 16x16x32 (two A tiles per B fragment: half the LDS reads), A fragments resident
//      (160 VGPRs), wave (mh, cg) takes half of the column tiles in groups of three
//   4  32 output channels per wave on 32x32x16 (A resident, 160 VGPRs), wave (mh, cg), 32-column tiles
//   5  mode 3 with the A fragments streamed from L2 through a ring (what the fused kernel would have to do: it has no
//      160 spare VGPRs)
// Build:  hipcc -O3 --offload-arch=gfx950 -o bin/s4_loop s4_loop.hip ;  run: bin/s4_loop [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

__device__ __forceinline__ f32x4 mfma16(const uint4 a, const uint4 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(const uint4 a, const uint4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the kernel's epilogue: four activations as two register pairs, z / (1 + 2^-z), packed to bf16
__device__ __forceinline__ uint2 swish_pack(const f32x4 acc) {
    f32x2 lo = {acc[0], acc[1]}, hi = {acc[2], acc[3]};
    const f32x2 elo = {__builtin_amdgcn_exp2f(-lo.x), __builtin_amdgcn_exp2f(-lo.y)};
    const f32x2 ehi = {__builtin_amdgcn_exp2f(-hi.x), __builtin_amdgcn_exp2f(-hi.y)};
    const f32x2 dlo = elo + 1.0f, dhi = ehi + 1.0f;
    lo = lo * f32x2{__builtin_amdgcn_rcpf(dlo.x), __builtin_amdgcn_rcpf(dlo.y)};
    hi = hi * f32x2{__builtin_amdgcn_rcpf(dhi.x), __builtin_amdgcn_rcpf(dhi.y)};
    const bf16x4 o = {(__bf16)lo.x, (__bf16)lo.y, (__bf16)hi.x, (__bf16)hi.y};
    return __builtin_bit_cast(uint2, o);
}

struct Args {
    const uint4 *afrag;  // [4 oc tiles][20 k-steps][64 lanes] (16x16x32) or [2][40][64] (32x32x16): any finite bf16
    uint16_t *x;         // [blocks][cb * 24][64]
    int iters, cb, P3, T, cat_plane;
};

template <bool TWO>
__device__ __forceinline__ void s4_pair(const uint4 (&A)[20], const unsigned char *r0, const unsigned char *r1, f32x4 &acc0, f32x4 &acc1) {
    uint4 ba0[4], ba1[4], bb0[4], bb1[4];
    auto ld = [&](uint4(&b0)[4], uint4(&b1)[4], int off) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            b0[s] = *reinterpret_cast<const uint4 *>(r0 + off + 16 * s);
            if (TWO) b1[s] = *reinterpret_cast<const uint4 *>(r1 + off + 16 * s);
        }
    };
    auto mm = [&](const uint4(&b0)[4], const uint4(&b1)[4], int a0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc0 = mfma16(A[a0 + s], b0[s], acc0);
            if (TWO) acc1 = mfma16(A[a0 + s], b1[s], acc1);
        }
    };
    ld(ba0, ba1, 0); ld(bb0, bb1, 80);
    mm(ba0, ba1, 0); ld(ba0, ba1, 160);
    mm(bb0, bb1, 4); ld(bb0, bb1, 240);
    mm(ba0, ba1, 8); ld(ba0, ba1, 320);
    mm(bb0, bb1, 12);
    mm(ba0, ba1, 16);
    constexpr int T = TWO ? 2 : 1;
    __builtin_amdgcn_sched_group_barrier(0x100, 8 * T, 0);
#pragma unroll
    for (int i = 0; i < 12 * T; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8 * T, 0);
}

__device__ __forceinline__ void fill_cat(unsigned char *smem, int bytes, int tid) {
    // finite bf16 around 1.0 with varying mantissas
    for (int i = tid; i < bytes / 4; i += 256) reinterpret_cast<unsigned *>(smem)[i] = 0x3F803F00u + ((unsigned)(i * 2654435761u) & 0x007F007Fu);
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256, MODE >= 3 && MODE != 5 ? 2 : 2) void s4_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, nn = lane & 15;
    fill_cat(smem, 4 * a.cat_plane, tid);
    uint16_t *xb = a.x + (size_t)blockIdx.x * a.cb * a.T * 64;
    const int ncols = a.cb * a.T;
    auto row_of = [&](int col) {  // byte offset of a column's first CAT row
        if (MODE == 2) return col * 80;
        const int ch = col / a.T;
        return (ch * a.P3 + (col - ch * a.T)) * 80;
    };
    if constexpr (MODE <= 2) {
        uint4 A[20];
#pragma unroll
        for (int s = 0; s < 20; ++s) A[s] = a.afrag[(w * 20 + s) * 64 + lane];
        const unsigned char *cat_r = smem + (size_t)q * a.cat_plane;
        const int ntiles = (ncols + 15) >> 4;
        f32x4 sink = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < a.iters; ++it) {
            for (int tile = 0; tile < ntiles; tile += 2) {
                const int col0 = tile * 16 + nn, col1 = col0 + 16;
                const bool v0 = col0 < ncols, v1 = col1 < ncols;
                const unsigned char *r0 = cat_r + row_of(v0 ? col0 : ncols - 1), *r1 = cat_r + row_of(v1 ? col1 : ncols - 1);
                f32x4 acc0 = {0.1f, 0.1f, 0.1f, 0.1f}, acc1 = acc0;
                if (tile + 1 < ntiles) s4_pair<true>(A, r0, r1, acc0, acc1);
                else s4_pair<false>(A, r0, r1, acc0, acc1);
                if (MODE == 1) {
                    sink += acc0 + acc1;
                } else {
                    if (v0) *reinterpret_cast<uint2 *>(xb + (size_t)col0 * 64 + 16 * w + 4 * q) = swish_pack(acc0);
                    if (v1) *reinterpret_cast<uint2 *>(xb + (size_t)col1 * 64 + 16 * w + 4 * q) = swish_pack(acc1);
                }
            }
            __syncthreads();  // (the kernel has a barrier per stage; one per iteration here)
        }
        if (MODE == 1) *reinterpret_cast<uint2 *>(xb + 16 * w + 4 * q + (size_t)nn * 64) = swish_pack(sink);
    } else if constexpr (MODE == 3 || MODE == 5) {
        const int mh = w >> 1, cg = w & 1;
        constexpr int RING = MODE == 5 ? 5 : 20;
        uint4 A[RING][2];
        auto frag = [&](int i, int s) { return a.afrag[((2 * mh + i) * 20 + s) * 64 + lane]; };
        if (MODE == 3) {
#pragma unroll
            for (int s = 0; s < 20; ++s) { A[s % RING][0] = frag(0, s); A[s % RING][1] = frag(1, s); }
        }
        const unsigned char *cat_r = smem + (size_t)q * a.cat_plane;
        const int ntiles = (ncols + 15) >> 4, split = (ntiles + 1) >> 1;
        const int lo = cg ? split : 0, hi = cg ? ntiles : split;
        auto boff = [](int s) { return (s >> 2) * 80 + (s & 3) * 16; };
        for (int it = 0; it < a.iters; ++it) {
            for (int t0 = lo; t0 < hi; t0 += 3) {  // groups of three column tiles (the last may repeat a tile: same cost)
                const unsigned char *r[3];
                f32x4 acc[3][2];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int col = (t0 + j < hi ? t0 + j : hi - 1) * 16 + nn;
                    r[j] = cat_r + row_of(col < ncols ? col : ncols - 1);
                    acc[j][0] = acc[j][1] = f32x4{0.1f, 0.1f, 0.1f, 0.1f};
                }
                if (MODE == 5) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) { A[s][0] = frag(0, s); A[s][1] = frag(1, s); }
                }
                uint4 B[3][3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    B[0][j] = *reinterpret_cast<const uint4 *>(r[j] + boff(0));
                    B[1][j] = *reinterpret_cast<const uint4 *>(r[j] + boff(1));
                }
#pragma unroll
                for (int s = 0; s < 20; ++s) {
                    if (MODE == 5 && s + 4 < 20) { A[(s + 4) % RING][0] = frag(0, s + 4); A[(s + 4) % RING][1] = frag(1, s + 4); }
                    if (s + 2 < 20) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) B[(s + 2) % 3][j] = *reinterpret_cast<const uint4 *>(r[j] + boff(s + 2));
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        acc[j][0] = mfma16(A[s % RING][0], B[s % 3][j], acc[j][0]);
                        acc[j][1] = mfma16(A[s % RING][1], B[s % 3][j], acc[j][1]);
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
                for (int s = 0; s < 20; ++s) {
                    if (MODE == 5 && s + 4 < 20) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        if (s + 2 < 20) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int col = (t0 + j) * 16 + nn;
                    if (t0 + j < hi && col < ncols) {
                        *reinterpret_cast<uint2 *>(xb + (size_t)col * 64 + 16 * (2 * mh) + 4 * q) = swish_pack(acc[j][0]);
                        *reinterpret_cast<uint2 *>(xb + (size_t)col * 64 + 16 * (2 * mh + 1) + 4 * q) = swish_pack(acc[j][1]);
                    }
                }
            }
            __syncthreads();
        }
    } else {  // MODE 4: 32x32x16
        const int mh = w >> 1, cg = w & 1, half = lane >> 5, j32 = lane & 31;
        uint4 A[40];
#pragma unroll
        for (int s = 0; s < 40; ++s) A[s] = a.afrag[(mh * 40 + s) * 64 + lane];
        const int ntiles = (ncols + 31) >> 5, split = (ntiles + 1) >> 1;
        const int lo = cg ? split : 0, hi = cg ? ntiles : split;
        // k-step s (16 k): tap s / 8, channels 16 (s % 8) + 8 half .. +7  ->  plane ((16 (s%8) + 8 half) % 32) / 8, slot (16 (s%8)) / 32
        auto boff = [&](int s) {
            const int c = 16 * (s & 7) + 8 * half;
            return ((c & 31) >> 3) * a.cat_plane + (s >> 3) * 80 + (c >> 5) * 16;
        };
        for (int it = 0; it < a.iters; ++it) {
            for (int t0 = lo; t0 < hi; ++t0) {
                const int col = t0 * 32 + j32;
                const unsigned char *r = smem + row_of(col < ncols ? col : ncols - 1);
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.1f;
                uint4 B[4];
#pragma unroll
                for (int s = 0; s < 3; ++s) B[s] = *reinterpret_cast<const uint4 *>(r + boff(s));
#pragma unroll
                for (int s = 0; s < 40; ++s) {
                    if (s + 3 < 40) B[(s + 3) & 3] = *reinterpret_cast<const uint4 *>(r + boff(s + 3));
                    acc = mfma32(A[s], B[s & 3], acc);
                }
                if (col < ncols) {
                    // D layout of 32x32: lane (half, j) holds rows 8 i + 4 half + r (i = 0..3, r = 0..3) of column j
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *reinterpret_cast<uint2 *>(xb + (size_t)col * 64 + 32 * mh + 8 * i + 4 * half) =
                            swish_pack(f32x4{acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]});
                }
            }
            __syncthreads();
        }
    }
}

template <int MODE>
static double run(const Args &a0, int cb, int grid, const char *what, double base) {
    Args a = a0;
    a.cb = cb;
    a.P3 = 28; a.T = 24;
    a.cat_plane = (((cb * a.P3 + 1) * 5 + 15) & ~15) * 16;
    const size_t lds = (size_t)4 * a.cat_plane;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(s4_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    Args warm = a;
    warm.iters = 20;
    hipLaunchKernelGGL(s4_kernel<MODE>, dim3(grid), dim3(256), lds, 0, warm);
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(s4_kernel<MODE>, dim3(grid), dim3(256), lds, 0, a);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    hipFuncAttributes at;
    CHECK(hipFuncGetAttributes(&at, reinterpret_cast<const void *>(s4_kernel<MODE>)));
    const double chunks = (double)grid * a.iters * cb;
    const double ns = best * 1e6 / chunks;
    const double tf = 1.96608e6 / ns * 1e-3;  // FLOP per chunk / ns -> TFLOP/s
    printf("mode %d  cb %d  %-62s %6.3f ns/chunk  %6.0f TF (%.2f of 2.5 PF)  %3d VGPRs%s\n", MODE, cb, what, ns, tf, tf / 2500.0, at.numRegs,
           base > 0 ? (ns < base ? "  faster" : "  slower") : "");
    fflush(stdout);
    return ns;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 3000;
    int dev_cus = 256;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    dev_cus = prop.multiProcessorCount;
    const int grid = 2 * dev_cus;
    std::vector<uint32_t> h(4 * 40 * 64 * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3C003C00u + ((uint32_t)(i * 2246822519u) & 0x007F007Fu);  // bf16 ~ 0.0078 .. 0.0157
    uint4 *afrag;
    uint16_t *x;
    CHECK(hipMalloc(reinterpret_cast<void **>(&afrag), h.size() * 4));
    CHECK(hipMemcpy(afrag, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(reinterpret_cast<void **>(&x), (size_t)grid * 8 * 24 * 64 * 2));
    Args a;
    a.afrag = afrag; a.x = x; a.iters = iters;
    printf("S4 (merge_conv1) alone, %d blocks of 4 waves (2 per CU), %d iterations; 480 matrix cycles per chunk and CU = 0.78 ns/chunk at the full rate\n", grid, iters);
    const double b4 = run<0>(a, 4, grid, "kernel's scheme: 16 channels per wave, tile pairs, epilogue", 0);
    run<1>(a, 4, grid, "  without the epilogue", b4);
    run<2>(a, 4, grid, "  chunk-contiguous columns (no straddling tiles)", b4);
    run<3>(a, 4, grid, "32 channels per wave, 16x16x32, A resident (160 VGPRs)", b4);
    run<5>(a, 4, grid, "32 channels per wave, 16x16x32, A streamed from L2 (ring of 5)", b4);
    const double b8 = run<0>(a, 8, grid, "kernel's scheme at 8 chunks per iteration", 0);
    run<3>(a, 8, grid, "32 channels per wave, 16x16x32, A resident", b8);
    run<4>(a, 8, grid, "32 channels per wave, 32x32x16, A resident (160 VGPRs)", b8);
    run<5>(a, 8, grid, "32 channels per wave, 16x16x32, A streamed from L2", b8);
    return 0;
}
