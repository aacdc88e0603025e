// k_fused.hip — the whole convolutional front of ConvLSTM_w_ref in ONE kernel on the bf16 matrix cores:
//   chunk arrays (signal f32[n][L], sequence i8, mapping i16, lengths i16; 472 B/chunk @C100)
//     -> sig_conv1 -> sig_conv2 -> sig_conv3 \
//     -> k-mer one-hot -> seq_conv1 -> seq_conv2 -> cat -> merge_conv1 -> x bf16[n][T][64]   (3 KB/chunk)
// Replaces, fused: encoded_kmers.compute_encoded_kmer_batch (src/remora/encoded_kmers.pyx:13-45) and
// models/ConvLSTM_w_ref.py:41-50 (five Conv1d + BatchNorm1d(eval, folded) + swish triples and the cat).
// Every intermediate lives in LDS as bf16; HBM sees the chunk arrays once and x once (the unfused pipeline moved
// 68.9 KB/chunk through HBM in fp32, profiles/r01; this one 3.5 KB/chunk).
//
// All five convolutions are implicit GEMMs on v_mfma_f32_16x16x32_bf16 (fp32 accumulate):
//   D[oc][col] = sum_k A[oc][k] * B[k][col],  col = (chunk, output position), k = tap * C + channel
// With channel-last activations [row = position][C channels] a stride-S convolution needs no im2col: the K
// extent of column (chunk, pos) is the CONTIGUOUS run that starts at row chunk*Pin + S*pos — lane (q, n) of
// k-step s reads the 8 bf16 at flat index row*C + 32 s + 8 q with one ds_read_b128.  K is padded to a multiple of
// 32 with zero weights; the padded k read finite bf16 of the following rows (never NaN: every activation buffer
// is zeroed once and only ever holds finite activations).
//   layer        C     K (padded)   k-steps   M (oc)   where the operand lives
//   sig_conv1    -     5            VALU      4        signal f32 in LDS           -> SIG1 [row][4]
//   sig_conv2    4     20 (32)      1         16       SIG1                        -> SIG2 [row][16]
//   seq_conv1    40    200 (224)    7         16       one-hot OH, 5 planes        -> SEQ1 [row][16]
//   sig_conv3    16    144 (160)    5         64       SIG2 (stride 3)             -> CAT channels 0..63
//   seq_conv2    16    208 (224)    7         64       SEQ1 (stride 3)             -> CAT channels 64..127
//   merge_conv1  128   640          20        64       CAT, 4 planes               -> x (global, bf16)
// The k-mer one-hot (36 channels, padded to 40) is built in LDS from the 3-bit base codes of the base that
// covers each signal position (the gather form of the reference's scatter loops) and consumed by the matrix
// cores — exact in bf16, and it moves seq_conv1's 69 k gather-adds per chunk off the VALU.
//
// Wave roles: 4 waves per block, CB chunks per block iteration.  M = 16 layers: the waves split the column
// tiles; M = 64 layers: wave w owns output channels 16w..16w+15 and walks all column tiles.  Every wave keeps
// the A fragments of merge_conv1 in registers for the lifetime of the block (80 VGPRs); the fragments of the other
// layers are re-fetched from L2 where their stage begins (their VGPRs hold B fragments in flight during the other stages).
// LDS bank behaviour of the B reads (ds_read_b128, 16-lane service groups that mix two q values):
//   SIG2 / SEQ1 (32-byte rows, stride 3): slot = 6 n + 4 s + q  — distinct over a group;
//   CAT: plane q (8-channel group q of every 32), rows of 4 slots padded to 5 — distinct over a group;
//   OH: plane per 8-channel group, one slot per row — conflict-free except where the two q of a group fall on
//   different taps (2 of 7 k-steps, one extra LDS cycle).
#include "rmr_internal.h"
#include "rmr_math.h"

#include <type_traits>

namespace rmr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct FusedArgs {
    const float *signal;   // [n][L]
    const int8_t *seqs;    // [n][seq_w]
    const int16_t *maps;   // [n][map_w]
    const int16_t *lens;   // [n]
    const uint4 *a_sig2, *a_seq1, *a_sig3, *a_seq2, *a_merge1;  // bf16 A fragments [oc/16][k-steps][64 lanes]
    const float *w_sig1, *b_sig1;                               // [5][4], [4]  (VALU layer), scaled by log2(e)
    const float *b_sig2, *b_seq1, *b_sig3, *b_seq2, *b_merge1;  // folded biases, scaled by log2(e)
    uint16_t *x;           // bf16 [n][T][64]
    int64_t n;
    int L, P1, P2, P3, T, seq_w, map_w, maxlen, cb;
    // LDS carve, byte offsets (all multiples of 16)
    int o_sig, o_seq, o_map, o_len, o_tab, o_col3, o_col4, o_sig1, o_sig2, o_seq1, o_oh;
    int oh_plane, cat_plane;  // bytes
    int lds_bytes;
    FastDiv d_L, d_P1, d_P3, d_T, d_maxlen;
    unsigned mg_ps2, mg_pq1;  // ceil(2^32 / tile pairs per chunk) of sig_conv2 / seq_conv1: S2's item -> (chunk, pair) on the SALU
    int abl;  // experiment builds only (-DRMR_TIMING_ABLATIONS): bit mask of stages to skip, see ABL() below
    // Position windows (WIN kernels: chunks too long for a CU's LDS, e.g. chunk_context (500, 500)).  The kernel then works on
    // VIRTUAL chunks: window `win` of chunk `c` covers output positions [win * Tw, (win + 1) * Tw) of the chunk's T_total and
    // reads L samples from sample 3 * Tw * win on (L, P1 .. T above are the WINDOW's geometry, T >= Tw); the k-mer one-hot looks
    // positions up in the whole chunk's mapping row.  Per virtual chunk of an iteration, LDS at o_win: {first sample, valid
    // outputs, x row of its first output (two words)}.
    int nwin, Tw, L_total, T_total, o_win;
    FastDiv d_nwin, d_L4, d_seqw, d_mapw;
};

// Timing ablations (which stage costs what): compiled in only with -DRMR_TIMING_ABLATIONS (make abl ->
// libremora_hip_abl.so, selected with REMORA_HIP_LIB); the shipped library has no way to skip work.
// bits: 1 S0 loads, 2 S1a (sig_conv1 / covering base / codes), 4 S1b one-hot, 8 S2, 16 S3, 32 S4, 64 swish -> identity
#ifdef RMR_TIMING_ABLATIONS
#define ABL(bit) (a.abl & (bit))
// stage clock of the experiment build (RMR_FUSED_STAGE_CLOCK=1): shader-clock time every wave spends between the
// marks of an iteration, summed over iterations and blocks per wave index -> g_stage_clock[wave][mark]
__device__ unsigned long long g_stage_clock[4][16];
#define TS_DECL unsigned long long ts_prev = __builtin_readcyclecounter(), ts_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define TS(i) do { if (a.abl & 0x1000) { const unsigned long long t_ = __builtin_readcyclecounter(); ts_acc[i] += t_ - ts_prev; ts_prev = t_; } } while (0)
#define TS_FLUSH do { if ((a.abl & 0x1000) && lane == 0) for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&g_stage_clock[w][i_], ts_acc[i_]); } while (0)
#else
#define ABL(bit) 0
#define TS_DECL
#define TS(i)
#define TS_FLUSH
#endif

__device__ __forceinline__ int fdiv(int x, FastDiv d) { return (int)(((float)x + 0.5f) * d.inv); }

// The 16-bit operand type is a template parameter of everything below: F16 = false -> bf16 (8 exponent / 7 mantissa bits;
// BASELINE configs[3]/[4] name it), true -> IEEE half (5 / 10 bits: eight times finer rounding at the same matrix rate
// - v_mfma_f32_16x16x32_f16 and _bf16 are both 16 cycles; activations here are O(1..10), far from half's 65504).
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const uint4 a, const uint4 b, const f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Activations are carried SCALED by log2(e): every layer's weights/bias are prepared on the host so that the
// accumulator holds z = log2(e) * y (y = the reference's pre-activation), and the stored activation is
//   a' = z / (1 + 2^-z) = log2(e) * swish(y)          (src/remora/activations.py:4-18: swish(y) = y * sigmoid(y))
// which saves the multiply by -log2(e) in front of every v_exp_f32; the next layer's weights are unchanged (its bias
// is scaled by log2(e)), and the last layer multiplies by POST = 1 / log2(e) to hand over the true activation.
// Four accumulator rows (consecutive output channels) -> four bf16: 8 bytes.
// The VALU is what this kernel is short of, so the four activations of an accumulator are worked as two
// register pairs: 4 v_exp + 2 v_pk_add_f32 + 4 v_rcp + 2 v_pk_mul_f32 + 2 packs = 14 instructions; left to itself hipcc
// paired elements (1, 2), moved them into an aligned register pair and re-assembled the result with v_perm / v_alignbit
// (22).  Packed and scalar fp32 add / mul round alike: same bits.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef RMR_SWISH_PAIRS
#define RMR_SWISH_PAIRS 1
#endif
template <bool F16>
__device__ __forceinline__ uint2 swish_pack(const f32x4 acc, const int no_swish = 0, const float post = 1.0f) {
    f32x2 lo = {acc[0], acc[1]}, hi = {acc[2], acc[3]};
    if (!RMR_SWISH_PAIRS) {  // the element-by-element form, kept for the A/B (tools/ab_variants.py)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = acc[r];
            const float y = no_swish ? z : z * fast_rcp(1.0f + __builtin_amdgcn_exp2f(-z)) * post;
            if (r < 2) lo[r] = y;
            else hi[r - 2] = y;
        }
    } else if (!no_swish) {
        const f32x2 elo = {__builtin_amdgcn_exp2f(-lo.x), __builtin_amdgcn_exp2f(-lo.y)};
        const f32x2 ehi = {__builtin_amdgcn_exp2f(-hi.x), __builtin_amdgcn_exp2f(-hi.y)};
        const f32x2 dlo = elo + 1.0f, dhi = ehi + 1.0f;
        const f32x2 rlo = {fast_rcp(dlo.x), fast_rcp(dlo.y)}, rhi = {fast_rcp(dhi.x), fast_rcp(dhi.y)};
        lo = lo * rlo * post;
        hi = hi * rhi * post;
    }
    if constexpr (F16) {
        const f16x4 o = {(_Float16)lo.x, (_Float16)lo.y, (_Float16)hi.x, (_Float16)hi.y};
        return __builtin_bit_cast(uint2, o);
    } else {
        const bf16x4 o = {(__bf16)lo.x, (__bf16)lo.y, (__bf16)hi.x, (__bf16)hi.y};
        return __builtin_bit_cast(uint2, o);
    }
}

// B fragments (8 bf16 per lane) of NS k-steps of one or two 16-column tiles, read ahead of the MFMAs that use them
template <int NS, bool TWO, typename Off>
__device__ __forceinline__ void load_b(uint4 (&b0)[NS], uint4 (&b1)[NS], const unsigned char *r0, const unsigned char *r1, Off off) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        b0[s] = *reinterpret_cast<const uint4 *>(r0 + off(s));
        if (TWO) b1[s] = *reinterpret_cast<const uint4 *>(r1 + off(s));
    }
}
template <bool F16, int NS, bool TWO, int A0 = 0, int NA>
__device__ __forceinline__ void mma_b(const uint4 (&A)[NA], const uint4 (&b0)[NS], const uint4 (&b1)[NS], f32x4 &acc0, f32x4 &acc1) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        acc0 = mfma16<F16>(A[A0 + s], b0[s], acc0);
        if (TWO) acc1 = mfma16<F16>(A[A0 + s], b1[s], acc1);
    }
}

// sig_conv3 + seq_conv2 of one or two column tiles: the B fragments of sig_conv3 are all in flight before its first
// MFMA; those of seq_conv2 are read one per MFMA from then on, so that ~10 reads stay ahead of the matrix pipe
template <bool F16, bool TWO>
__device__ __forceinline__ void s3_pair(const uint4 (&Asig3)[5], const uint4 (&Aseq2)[7], const unsigned char *g0, const unsigned char *g1,
                                        const unsigned char *q0, const unsigned char *q1, f32x4 &as0, f32x4 &as1, f32x4 &aq0,
                                        f32x4 &aq1) {
    uint4 bs0[5], bs1[5], bq0[7], bq1[7];
    load_b<5, TWO>(bs0, bs1, g0, g1, [](int s) { return 64 * s; });
    load_b<7, TWO>(bq0, bq1, q0, q1, [](int s) { return 64 * s; });
    mma_b<F16, 5, TWO, 0>(Asig3, bs0, bs1, as0, as1);
    mma_b<F16, 7, TWO, 0>(Aseq2, bq0, bq1, aq0, aq1);
    // pin the issue order (hipcc otherwise sinks every ds_read next to its MFMA: no read in flight under the MFMAs)
    constexpr int T = TWO ? 2 : 1;
    __builtin_amdgcn_sched_group_barrier(0x100, 5 * T, 0);
#pragma unroll
    for (int i = 0; i < 7 * T; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 5 * T, 0);
}

// one stride-3 convolution (NS k-steps) of one or two column tiles, every B fragment in flight before the first MFMA
template <bool F16, int NS, bool TWO>
__device__ __forceinline__ void s3_one(const uint4 (&A)[NS], const unsigned char *g0, const unsigned char *g1, f32x4 &a0, f32x4 &a1) {
    uint4 b0[NS], b1[NS];
    load_b<NS, TWO>(b0, b1, g0, g1, [](int s) { return 64 * s; });
    mma_b<F16, NS, TWO, 0>(A, b0, b1, a0, a1);
    constexpr int T = TWO ? 2 : 1;
    __builtin_amdgcn_sched_group_barrier(0x100, NS * T, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NS * T, 0);
}

// merge_conv1 of one or two column tiles: the four 32-channel slots of tap t+1 are read while the MFMAs of tap t run
template <bool F16, bool TWO>
__device__ __forceinline__ void s4_pair(const uint4 (&Am1)[20], const unsigned char *r0, const unsigned char *r1, f32x4 &acc0, f32x4 &acc1) {
    uint4 ba0[4], ba1[4], bb0[4], bb1[4];
    load_b<4, TWO>(ba0, ba1, r0, r1, [](int s) { return 16 * s; });
    load_b<4, TWO>(bb0, bb1, r0, r1, [](int s) { return 80 + 16 * s; });
    mma_b<F16, 4, TWO, 0>(Am1, ba0, ba1, acc0, acc1);
    load_b<4, TWO>(ba0, ba1, r0, r1, [](int s) { return 160 + 16 * s; });
    mma_b<F16, 4, TWO, 4>(Am1, bb0, bb1, acc0, acc1);
    load_b<4, TWO>(bb0, bb1, r0, r1, [](int s) { return 240 + 16 * s; });
    mma_b<F16, 4, TWO, 8>(Am1, ba0, ba1, acc0, acc1);
    load_b<4, TWO>(ba0, ba1, r0, r1, [](int s) { return 320 + 16 * s; });
    mma_b<F16, 4, TWO, 12>(Am1, bb0, bb1, acc0, acc1);
    mma_b<F16, 4, TWO, 16>(Am1, ba0, ba1, acc0, acc1);
    // pin: two taps of reads up front, then one read issued per MFMA, so that a tap's reads are a tap ahead of its MFMAs
    constexpr int T = TWO ? 2 : 1;
    __builtin_amdgcn_sched_group_barrier(0x100, 8 * T, 0);
#pragma unroll
    for (int i = 0; i < 12 * T; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8 * T, 0);
}

// merge_conv1 with HALF the fragments in flight (one tap = 4 slots per tile): for the three-waves-per-SIMD build, where
// the other two waves of the SIMD cover the wait between a tap's reads and its MFMAs
template <bool F16, bool TWO>
__device__ __forceinline__ void s4_pair_lean(const uint4 (&Am1)[20], const unsigned char *r0, const unsigned char *r1, f32x4 &acc0,
                                             f32x4 &acc1) {
    uint4 ba0[4], ba1[4], bb0[2], bb1[2];
    load_b<4, TWO>(ba0, ba1, r0, r1, [](int s) { return 16 * s; });
#pragma unroll
    for (int tap = 0; tap < 5; ++tap) {
        // first half of the next tap's slots requested under this tap's MFMAs
        if (tap < 4) load_b<2, TWO>(bb0, bb1, r0, r1, [tap](int s) { return 80 * (tap + 1) + 16 * s; });
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc0 = mfma16<F16>(Am1[4 * tap + s], ba0[s], acc0);
            if (TWO) acc1 = mfma16<F16>(Am1[4 * tap + s], ba1[s], acc1);
        }
        if (tap < 4) {
            ba0[0] = bb0[0]; ba0[1] = bb0[1];
            if (TWO) { ba1[0] = bb1[0]; ba1[1] = bb1[1]; }
#pragma unroll
            for (int s = 2; s < 4; ++s) {
                ba0[s] = *reinterpret_cast<const uint4 *>(r0 + 80 * (tap + 1) + 16 * s);
                if (TWO) ba1[s] = *reinterpret_cast<const uint4 *>(r1 + 80 * (tap + 1) + 16 * s);
            }
        }
    }
}

// the chunk arrays of one block iteration, one element per thread and array, on their way from HBM to LDS
struct InRegs {
    float4 sig;
    int8_t seq;
    int16_t map, len;
};

// which A fragments stay in registers for the lifetime of the block (the rest is re-fetched from L2 per iteration)
#ifndef RMR_FUSED_RES_SMALL
#define RMR_FUSED_RES_SMALL 0  // sig_conv2 + seq_conv1 (32 VGPRs)
#endif
#ifndef RMR_FUSED_RES_MID
#define RMR_FUSED_RES_MID 0    // sig_conv3 + seq_conv2 (48 VGPRs)
#endif
#ifndef RMR_FUSED_S3_SPLIT
#define RMR_FUSED_S3_SPLIT 0   // 1: S3 as two passes (sig_conv3, seq_conv2) instead of both convolutions per tile pair
#endif
#ifndef RMR_FUSED_S4_LEAN
#define RMR_FUSED_S4_LEAN 0    // 1: merge_conv1 with 12 instead of 16 fragments in flight per tile pair
#endif
#ifndef RMR_FUSED_STREAM_M1
#define RMR_FUSED_STREAM_M1 0  // 1: merge_conv1's fragments (80 VGPRs) fetched per iteration after S3 instead of living in registers
#endif

template <int K, bool F16, bool WIN>
#ifndef RMR_FUSED_WAVES_EU
// Waves per SIMD the register budget is set for.  2 (254 VGPRs, merge_conv1 fragments resident, two blocks per CU) is the
// shipped build.  The three-waves build (-DRMR_FUSED_WAVES_EU=3 -DRMR_FUSED_STREAM_M1=1 -DRMR_FUSED_S3_SPLIT=1
// -DRMR_FUSED_S4_LEAN=1: 166 VGPRs, no spills, three blocks of three chunks per CU) measured 7.28 ns/chunk against 6.72
// (round 3, tools/ab_variants.py): the extra wave buys 7 %, streaming the fragments, the leaner read-ahead and the
// smaller tiles cost 17 % - the kernel is bound by LDS read bandwidth (one ds_read_b128 per MFMA = 256 B/clk at full
// matrix rate), not by latency.
#define RMR_FUSED_WAVES_EU 2
#endif
__global__ __launch_bounds__(256, RMR_FUSED_WAVES_EU) void fused_front_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CG = (4 * K + 7) / 8;                // 8-channel groups of a one-hot row
    constexpr int KS_SEQ1 = (5 * CG * 8 + 31) / 32;    // k-steps of seq_conv1
    constexpr int KS_SIG3 = 5, KS_SEQ2 = 7, KS_M1 = 20;
    static_assert(K <= 10, "base codes of a k-mer are packed 3 bits each into 32 bits");
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15;

    // ---- register-resident A fragments of merge_conv1 (80 VGPRs); the other layers' fragments (L2-resident, 60 KB in
    //      all) are fetched where their stage begins, so that each stage has room for its B fragments in flight ----
    // fragment loads go through buffer descriptors: wave-uniform base + scalar fragment offset + ONE per-lane offset
    // VGPR (lane * 16), instead of one 64-bit address VGPR pair per fragment that the compiler hoists and spills
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int lane16 = lane * 16;
    auto frag = [&](const uint4 *base, int nfrag, int idx) -> uint4 {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(base), 0, nfrag * 1024, 0x00020000);
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, idx * 1024, 0);
        return __builtin_bit_cast(uint4, v);
    };
    uint4 Am1[KS_M1];
    auto load_m1 = [&]() {
#pragma unroll
        for (int s = 0; s < KS_M1; ++s) Am1[s] = frag(a.a_merge1, 4 * KS_M1, wu * KS_M1 + s);
    };
    if (!RMR_FUSED_STREAM_M1) load_m1();
    uint4 Asig2, Aseq1[KS_SEQ1], Asig3[KS_SIG3], Aseq2[KS_SEQ2];
    auto load_small = [&]() {
        Asig2 = frag(a.a_sig2, 1, 0);
#pragma unroll
        for (int s = 0; s < KS_SEQ1; ++s) Aseq1[s] = frag(a.a_seq1, KS_SEQ1, s);
    };
    auto load_mid = [&]() {
#pragma unroll
        for (int s = 0; s < KS_SIG3; ++s) Asig3[s] = frag(a.a_sig3, 4 * KS_SIG3, wu * KS_SIG3 + s);
#pragma unroll
        for (int s = 0; s < KS_SEQ2; ++s) Aseq2[s] = frag(a.a_seq2, 4 * KS_SEQ2, wu * KS_SEQ2 + s);
    };
    if (RMR_FUSED_RES_SMALL) load_small();
    if (RMR_FUSED_RES_MID) load_mid();
    float w1[5][4];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int o = 0; o < 4; ++o) w1[t][o] = a.w_sig1[t * 4 + o];
    const float4 b1 = *reinterpret_cast<const float4 *>(a.b_sig1);

    float *s_sig = reinterpret_cast<float *>(smem + a.o_sig);
    int8_t *s_seq = reinterpret_cast<int8_t *>(smem + a.o_seq);
    int16_t *s_map = reinterpret_cast<int16_t *>(smem + a.o_map);
    int16_t *s_len = reinterpret_cast<int16_t *>(smem + a.o_len);
    unsigned char *s_sig1 = smem + a.o_sig1;  // [row][4] bf16, 8 B rows
    unsigned char *s_sig2 = smem + a.o_sig2;  // [row][16] bf16, 32 B rows
    unsigned char *s_seq1 = smem + a.o_seq1;  // [row][16] bf16
    uint4 *s_tab = reinterpret_cast<uint4 *>(smem + a.o_tab);  // [64] one-hot pieces by base-code pair
    int2 *s_col3 = reinterpret_cast<int2 *>(smem + a.o_col3);  // [cb * P3] byte offsets of a column's first SIG2 / SEQ1 row
    int *s_col4 = reinterpret_cast<int *>(smem + a.o_col4);    // [cb * T]  byte offset of a column's first CAT row
    unsigned char *s_oh = smem + a.o_oh;      // CG planes x [row] x 16 B;  aliased by
    unsigned char *s_cat = smem + a.o_oh;     // 4 planes x [row][5 slots] x 16 B

    // every byte finite from the start (K padding reads rows that no stage of this iteration wrote)
    for (int i = tid * 16; i < a.lds_bytes; i += 256 * 16) *reinterpret_cast<uint4 *>(smem + i) = make_uint4(0, 0, 0, 0);

    const int tiles_sig2 = (a.P2 + 15) >> 4, tiles_seq1 = (a.P1 + 15) >> 4;
    const int pairs_sig2 = (tiles_sig2 + 1) >> 1, pairs_seq1 = (tiles_seq1 + 1) >> 1;
    const int pairs_chunk = pairs_sig2 + pairs_seq1;
    int nsearch = 1;  // bisection steps that cover maxlen + 1 mapping entries
    while ((1 << nsearch) < a.maxlen + 2) ++nsearch;

    const int64_t n_items = WIN ? a.n * a.nwin : a.n;  // (virtual) chunks
    const int64_t n_iters = (n_items + a.cb - 1) / a.cb;
    int *s_win = reinterpret_cast<int *>(smem + a.o_win);  // WIN: [cb][4]
    // S0: each array of the chunks of an iteration is ONE contiguous run in HBM; the launcher keeps every run within
    // 256 elements (float4 / byte / int16), so a thread carries one element of each from HBM to LDS.  (WIN: the virtual chunks
    // of an iteration are windows of possibly different chunks - every thread finds its own chunk and window)
    auto fetch_inputs = [&](int64_t it) -> InRegs {
        InRegs r;
        r.sig = make_float4(0.f, 0.f, 0.f, 0.f); r.seq = 0; r.map = 0; r.len = 0;
        if (it >= n_iters || ABL(1)) return r;
        const int64_t chunk0 = it * a.cb;
        const int nch = (int)((n_items - chunk0) < a.cb ? (n_items - chunk0) : a.cb);
        if constexpr (WIN) {
            auto chunk_win = [&](int ci, int &win) {  // virtual chunk chunk0 + ci -> (chunk, window); exact for n * nwin < 2^24
                const int v = (int)chunk0 + ci, c = fdiv(v, a.d_nwin);
                win = v - c * a.nwin;
                return c;
            };
            int win;
            if (tid < ((nch * a.L) >> 2)) {
                const int ci = fdiv(tid, a.d_L4), j = tid - ci * (a.L >> 2), c = chunk_win(ci, win);
                const int smp = 3 * a.Tw * win + 4 * j;  // (a multiple of 4: Tw is)
                if (smp < a.L_total) r.sig = *reinterpret_cast<const float4 *>(a.signal + (size_t)c * a.L_total + smp);
            }
            if (tid < nch * a.seq_w) {
                const int ci = fdiv(tid, a.d_seqw), c = chunk_win(ci, win);
                r.seq = a.seqs[(size_t)c * a.seq_w + (tid - ci * a.seq_w)];
            }
            if (tid < nch * a.map_w) {
                const int ci = fdiv(tid, a.d_mapw), c = chunk_win(ci, win);
                r.map = a.maps[(size_t)c * a.map_w + (tid - ci * a.map_w)];
            }
            if (tid < nch) r.len = a.lens[chunk_win(tid, win)];
        } else {
            if (tid < ((nch * a.L) >> 2)) r.sig = reinterpret_cast<const float4 *>(a.signal + (size_t)chunk0 * a.L)[tid];
            if (tid < nch * a.seq_w) r.seq = a.seqs[(size_t)chunk0 * a.seq_w + tid];
            if (tid < nch * a.map_w) r.map = a.maps[(size_t)chunk0 * a.map_w + tid];
            if (tid < nch) r.len = a.lens[chunk0 + tid];
        }
        return r;
    };
    auto store_inputs = [&](const InRegs &r, int64_t it) {
        reinterpret_cast<float4 *>(s_sig)[tid] = r.sig;  // the regions are 256 elements wide (launcher)
        s_seq[tid] = (unsigned char)r.seq < 4 ? r.seq : (int8_t)4;  // base codes 0..3, everything else (N, padding) = 4: missing
        s_map[tid] = r.map;
        if (tid < a.cb) s_len[tid] = (int16_t)(r.len < 0 ? 0 : (r.len > a.maxlen ? a.maxlen : r.len));
        if constexpr (WIN) {
            if (tid < a.cb) {
                const int64_t v = it * a.cb + tid;
                const int c = fdiv((int)(v < n_items ? v : 0), a.d_nwin), win = (int)(v < n_items ? v : 0) - c * a.nwin;
                const int left = a.T_total - win * a.Tw;
                const long long row = (long long)c * a.T_total + (long long)win * a.Tw;
                s_win[4 * tid + 0] = 3 * a.Tw * win;
                s_win[4 * tid + 1] = v < n_items ? (left < a.Tw ? left : a.Tw) : 0;
                s_win[4 * tid + 2] = (int)(row & 0xFFFFFFFFll);
                s_win[4 * tid + 3] = (int)(row >> 32);
            }
        }
    };
    RMR_SYNC();  // the zero fill is done before the first inputs land
    // column (chunk, position) -> operand row, once per block instead of a division and four multiply-adds per lane and
    // tile pair: the column tiles of the M = 64 layers run across chunk boundaries
    for (int col = tid; col < a.cb * a.P3; col += 256) {
        const int ch = fdiv(col, a.d_P3), p3 = col - ch * a.P3;
        s_col3[col] = make_int2((ch * a.P2 + 3 * p3) * 32, (ch * a.P1 + 3 * p3) * 32);
    }
    for (int col = tid; col < a.cb * a.T; col += 256) {
        const int ch = fdiv(col, a.d_T);
        s_col4[col] = (ch * a.P3 + (col - ch * a.T)) * 80;
    }
    // one-hot pieces: entry (b0 | b1 << 3) = the 8 channels of two neighbouring k-mer slots holding base codes b0, b1
    // (0..3 = A, C, G, T -> 1.0 in that channel, 4 = missing -> zeros); 12 instructions per piece when computed in place
    // (read from the first S1 on, behind the barrier at the top of the iteration)
    if (tid < 64) {
        constexpr unsigned ONE = F16 ? 0x3C00u : 0x3F80u;  // 1.0 in half / bf16
        const unsigned b0 = tid & 7u, b1c = tid >> 3;
        const unsigned one0 = ONE << ((b0 & 1u) * 16), one1 = ONE << ((b1c & 1u) * 16);
        uint4 v;
        v.x = (b0 >> 1) == 0 ? one0 : 0u;
        v.y = (b0 >> 1) == 1 ? one0 : 0u;
        v.z = (b1c >> 1) == 0 ? one1 : 0u;
        v.w = (b1c >> 1) == 1 ? one1 : 0u;
        s_tab[tid] = v;
    }
    store_inputs(fetch_inputs(blockIdx.x), blockIdx.x);

    TS_DECL;
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        const int64_t chunk0 = it * a.cb;
        const int nch = (int)((n_items - chunk0) < a.cb ? (n_items - chunk0) : a.cb);
        TS(9);
        RMR_SYNC();  // inputs of this iteration are in LDS; merge_conv1 of the previous one has read CAT (OH aliases it)
        // A fragments of the two M = 16 layers: fetched (L2-resident, 8 KB) at the top of every iteration and dead
        // after S2, so that they do not occupy 32 VGPRs while merge_conv1 runs
        TS(0);
        if (!RMR_FUSED_RES_SMALL) load_small();
        // ---- S1: sig_conv1 (VALU, fp32) -> SIG1;  k-mer one-hot rows -> OH ----
        const int n_s1a = ABL(2) ? 0 : nch;
        for (int i = tid; i < n_s1a * a.P1; i += 256) {
            const int ci = fdiv(i, a.d_P1), pos = i - ci * a.P1;
            const float *xs = s_sig + ci * a.L + pos;
            float4 acc = b1;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const float xv = xs[t];
                acc.x = fmaf(w1[t][0], xv, acc.x); acc.y = fmaf(w1[t][1], xv, acc.y);
                acc.z = fmaf(w1[t][2], xv, acc.z); acc.w = fmaf(w1[t][3], xv, acc.w);
            }
            const f32x4 v = {acc.x, acc.y, acc.z, acc.w};
            *reinterpret_cast<uint2 *>(s_sig1 + (size_t)i * 8) = swish_pack<F16>(v, ABL(64));
        }
        // one thread per signal position: the base p that covers it (the gather form of the reference's scatter loops:
        // p = #{mapping entries map[0..len] <= s} - 1, valid when 0 <= p < len), the K bases p..p+K-1 as 3-bit codes
        // (4 = missing), and the row of CG 16-byte pieces (bf16 1.0 = 0x3F80; piece cg = k-mer slots 2cg, 2cg+1)
        for (int row = tid; row < (ABL(4) ? 0 : nch) * a.L; row += 256) {
            const int ci = fdiv(row, a.d_L);
            int s = row - ci * a.L;
            if constexpr (WIN) s += s_win[4 * ci];  // the position within the whole chunk
            const int16_t *mp = s_map + ci * a.map_w;
            const int len = s_len[ci];
            // the count by binary steps, in arithmetic: a v_cmp + v_cndmask pair through VCC costs 25 cycles on gfx950 against
            // 4.5 per plain instruction (tools/ubench/valu_cycles.hip).  Entries behind map[len] read as map[len]: if that
            // one is <= s the count may overshoot len + 1 - the position has no base either way (p >= len).
            int cnt = 0;
            for (int step = 1 << (nsearch - 1); step; step >>= 1) {
                const int idx = cnt + step - 1;  // the entry that would bring the count to cnt + step
                int d = s - (int)mp[idx < len ? idx : len], m;
                asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m) : "v"(d));  // 0 when the entry is <= s, else -1 (in assembly: in
                cnt += step & ~m;                                   // C++ the optimiser makes a compare + select of it)
            }
            const int p = cnt - 1;
            const bool valid = p >= 0 && p < len;
            // the K base codes (one byte each, 0..4) squeezed to 3 bits each: two shift-or-mask steps per four bytes
            const unsigned char *sq = reinterpret_cast<const unsigned char *>(s_seq) + ci * a.seq_w + (valid ? p : 0);
            unsigned w0, w1;
            __builtin_memcpy(&w0, sq, 4);
            __builtin_memcpy(&w1, sq + 4, 4);  // (K < 8: bytes of the following bases, masked off below)
            auto squeeze = [](unsigned wv) {
                const unsigned x = (wv | (wv >> 5)) & 0x003F003Fu;
                return (x | (x >> 10)) & 0xFFFu;
            };
            unsigned code = squeeze(w0) | (squeeze(w1) << 12);
            if (K > 8) code |= (unsigned)sq[8] << 24;
            constexpr unsigned KMASK = (1u << (3 * K)) - 1u, ALL_MISSING = 0x24924924u & KMASK;  // 4 in every 3-bit slot
            code = valid ? (code & KMASK) : ALL_MISSING;
            if (K & 1) code |= 4u << (3 * K);  // the odd k-mer's last piece has no second base
            // slots 2cg, 2cg+1 of the k-mer by their two codes; all pieces are read before the first is written (the
            // compiler cannot tell that table and rows never overlap, and would wait for every read in turn)
            static_assert(CG <= 5, "pieces p0..p4");
            auto piece = [&](int cg) { return s_tab[(code >> (6 * cg)) & 63u]; };
            const uint4 p0 = piece(0), p1 = piece(CG > 1 ? 1 : 0), p2 = piece(CG > 2 ? 2 : 0), p3 = piece(CG > 3 ? 3 : 0),
                        p4 = piece(CG > 4 ? 4 : 0);
            unsigned char *dst = s_oh + (size_t)row * 16;
            auto put = [&](int cg, const uint4 &v) { *reinterpret_cast<uint4 *>(dst + (size_t)cg * a.oh_plane) = v; };
            put(0, p0);
            if (CG > 1) put(1, p1);
            if (CG > 2) put(2, p2);
            if (CG > 3) put(3, p3);
            if (CG > 4) put(4, p4);
        }
        TS(1);
        RMR_SYNC();
        TS(2);
        // ---- S2: sig_conv2 and seq_conv1 (M = 16): the waves split the column-tile pairs ----
        {
            // (biases are fetched per stage: L1-resident, and not worth 20 VGPRs for the whole block lifetime)
            const f32x4 b_sig2 = *reinterpret_cast<const f32x4 *>(a.b_sig2 + 4 * q);
            const f32x4 b_seq1 = *reinterpret_cast<const f32x4 *>(a.b_seq1 + 4 * q);
            // one-hot operand of seq_conv1: 8-group k8 = 4 s + q of k-step s sits at tap k8 / CG, channel group k8 % CG
            int oh_off[KS_SEQ1];
#pragma unroll
            for (int s = 0; s < KS_SEQ1; ++s) {
                const int k8 = 4 * s + q, tap = k8 / CG, cg = k8 - tap * CG;
                oh_off[s] = cg * a.oh_plane + tap * 16;
            }
            // the expensive items (seq_conv1 pairs: 14 MFMAs) first, striped over the waves, then the cheap ones (sig_conv2
            // pairs: 2 MFMAs): every wave gets the same number of each (+-1).  Striping the chunk-major list instead gave
            // waves 1 and 3 twice the seq_conv1 pairs of waves 0 and 2 (6 items per chunk against a stride of 4).
            const int n_seq_items = (ABL(8) ? 0 : nch) * pairs_seq1, n_items = (ABL(8) ? 0 : nch) * pairs_chunk;
            // (the item and what follows from it are wave-uniform: scalar registers, a multiply-high for the division)
            for (int item = wu; item < n_items; item += 4) {
                const bool is_sig = item >= n_seq_items;
                const int j = is_sig ? item - n_seq_items : item;
                const int ci = (int)__umulhi((unsigned)j, is_sig ? a.mg_ps2 : a.mg_pq1), r = j - ci * (is_sig ? pairs_sig2 : pairs_seq1);
                if (is_sig) {
                    int pos0 = 32 * r + nn, pos1 = pos0 + 16;
                    const bool v0 = pos0 < a.P2, v1 = pos1 < a.P2;
                    pos0 = v0 ? pos0 : a.P2 - 1;
                    pos1 = v1 ? pos1 : a.P2 - 1;
                    const unsigned char *r0 = s_sig1 + (size_t)(ci * a.P1 + pos0) * 8 + 16 * q;
                    const unsigned char *r1 = s_sig1 + (size_t)(ci * a.P1 + pos1) * 8 + 16 * q;
                    const uint2 x00 = *reinterpret_cast<const uint2 *>(r0), x01 = *reinterpret_cast<const uint2 *>(r0 + 8);
                    const uint2 x10 = *reinterpret_cast<const uint2 *>(r1), x11 = *reinterpret_cast<const uint2 *>(r1 + 8);
                    const f32x4 acc0 = mfma16<F16>(Asig2, make_uint4(x00.x, x00.y, x01.x, x01.y), b_sig2);
                    const f32x4 acc1 = mfma16<F16>(Asig2, make_uint4(x10.x, x10.y, x11.x, x11.y), b_sig2);
                    if (v0) *reinterpret_cast<uint2 *>(s_sig2 + (size_t)(ci * a.P2 + pos0) * 32 + 8 * q) = swish_pack<F16>(acc0, ABL(64));
                    if (v1) *reinterpret_cast<uint2 *>(s_sig2 + (size_t)(ci * a.P2 + pos1) * 32 + 8 * q) = swish_pack<F16>(acc1, ABL(64));
                } else {
                    int pos0 = 32 * r + nn, pos1 = pos0 + 16;
                    const bool v0 = pos0 < a.P1, v1 = pos1 < a.P1;
                    pos0 = v0 ? pos0 : a.P1 - 1;
                    pos1 = v1 ? pos1 : a.P1 - 1;
                    const unsigned char *r0 = s_oh + (size_t)(ci * a.L + pos0) * 16;
                    const unsigned char *r1 = s_oh + (size_t)(ci * a.L + pos1) * 16;
                    f32x4 acc0 = b_seq1, acc1 = b_seq1;
                    uint4 b0[KS_SEQ1], b1v[KS_SEQ1];  // all 14 B reads in flight before the first MFMA
#pragma unroll
                    for (int s = 0; s < KS_SEQ1; ++s) {
                        b0[s] = *reinterpret_cast<const uint4 *>(r0 + oh_off[s]);
                        b1v[s] = *reinterpret_cast<const uint4 *>(r1 + oh_off[s]);
                    }
                    mma_b<F16, KS_SEQ1, true, 0>(Aseq1, b0, b1v, acc0, acc1);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * KS_SEQ1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * KS_SEQ1, 0);
                    if (v0) *reinterpret_cast<uint2 *>(s_seq1 + (size_t)(ci * a.P1 + pos0) * 32 + 8 * q) = swish_pack<F16>(acc0, ABL(64));
                    if (v1) *reinterpret_cast<uint2 *>(s_seq1 + (size_t)(ci * a.P1 + pos1) * 32 + 8 * q) = swish_pack<F16>(acc1, ABL(64));
                }
            }
        }
        // A fragments of sig_conv3 / seq_conv2 (this wave's 16 output channels): on their way while the block gathers
        TS(3);
        if (!RMR_FUSED_RES_MID) load_mid();
        RMR_SYNC();
        TS(4);
        // ---- S3: sig_conv3 and seq_conv2 (stride 3, M = 64: wave w = channels 16w..16w+15) -> CAT ----
        {
            const int ncols = nch * a.P3;
            const int ntiles = ABL(16) ? 0 : (ncols + 15) >> 4;
            const f32x4 b_sig3 = *reinterpret_cast<const f32x4 *>(a.b_sig3 + 16 * w + 4 * q);
            const f32x4 b_seq2 = *reinterpret_cast<const f32x4 *>(a.b_seq2 + 16 * w + 4 * q);
            // CAT position of this lane's 4 output channels c0 = 16w + 4q (+64 for the sequence half):
            // plane (c0 % 32) / 8, slot c0 / 32, half (c0 % 8) / 4
            unsigned char *cat_w = s_cat + (size_t)((w & 1) * 2 + (q >> 1)) * a.cat_plane + (w >> 1) * 16 + (q & 1) * 8;
#if RMR_FUSED_S3_SPLIT
            // two passes (sig_conv3, then seq_conv2): half the fragments in flight, so that three waves fit a SIMD
            for (int pass = 0; pass < 2; ++pass) {
                const unsigned char *src = pass ? s_seq1 : s_sig2;
                const f32x4 bb = pass ? b_seq2 : b_sig3;
                const unsigned char *src_q = src + 16 * q;
                for (int tile = 0; tile < ntiles; tile += 2) {
                    const int col0 = tile * 16 + nn, col1 = col0 + 16;
                    const bool v0 = col0 < ncols, v1 = col1 < ncols;
                    const int2 e0 = s_col3[v0 ? col0 : ncols - 1], e1 = s_col3[v1 ? col1 : ncols - 1];
                    const unsigned char *g0 = src_q + (pass ? e0.y : e0.x), *g1 = src_q + (pass ? e1.y : e1.x);
                    f32x4 a0 = bb, a1 = bb;
                    const bool two = tile + 1 < ntiles;  // wave-uniform
                    if (pass == 0) {
                        if (two) s3_one<F16, 5, true>(Asig3, g0, g1, a0, a1);
                        else s3_one<F16, 5, false>(Asig3, g0, g1, a0, a1);
                    } else {
                        if (two) s3_one<F16, 7, true>(Aseq2, g0, g1, a0, a1);
                        else s3_one<F16, 7, false>(Aseq2, g0, g1, a0, a1);
                    }
                    if (v0) *reinterpret_cast<uint2 *>(cat_w + (size_t)col0 * 80 + 32 * pass) = swish_pack<F16>(a0, ABL(64));
                    if (v1) *reinterpret_cast<uint2 *>(cat_w + (size_t)col1 * 80 + 32 * pass) = swish_pack<F16>(a1, ABL(64));
                }
            }
#else
            const unsigned char *sig2_q = s_sig2 + 16 * q, *seq1_q = s_seq1 + 16 * q;
            auto col_of = [&](int c) { return c < ncols ? c : ncols - 1; };  // columns past the end repeat the last one
            int2 e0 = s_col3[col_of(nn)], e1 = s_col3[col_of(nn + 16)];      // (entries of the next pair are read a pair ahead)
            for (int tile = 0; tile < ntiles; tile += 2) {
                const int col0 = tile * 16 + nn, col1 = col0 + 16;
                const bool v0 = col0 < ncols, v1 = col1 < ncols;
                const unsigned char *g0 = sig2_q + e0.x, *g1 = sig2_q + e1.x;
                const unsigned char *q0 = seq1_q + e0.y, *q1 = seq1_q + e1.y;
                e0 = s_col3[col_of(col0 + 32)];
                e1 = s_col3[col_of(col1 + 32)];
                f32x4 as0 = b_sig3, as1 = b_sig3, aq0 = b_seq2, aq1 = b_seq2;
                if (tile + 1 < ntiles) s3_pair<F16, true>(Asig3, Aseq2, g0, g1, q0, q1, as0, as1, aq0, aq1);  // wave-uniform
                else s3_pair<F16, false>(Asig3, Aseq2, g0, g1, q0, q1, as0, as1, aq0, aq1);
                if (v0) {
                    *reinterpret_cast<uint2 *>(cat_w + (size_t)col0 * 80) = swish_pack<F16>(as0, ABL(64));
                    *reinterpret_cast<uint2 *>(cat_w + (size_t)col0 * 80 + 32) = swish_pack<F16>(aq0, ABL(64));
                }
                if (v1) {
                    *reinterpret_cast<uint2 *>(cat_w + (size_t)col1 * 80) = swish_pack<F16>(as1, ABL(64));
                    *reinterpret_cast<uint2 *>(cat_w + (size_t)col1 * 80 + 32) = swish_pack<F16>(aq1, ABL(64));
                }
            }
#endif
        }
        // the chunk arrays of the block's next iteration leave HBM now and land in LDS after merge_conv1 (their LDS
        // regions were last read in S1)
        // (vmcnt retires in order: the bias of S4 is requested BEFORE the prefetch so that waiting for it does not wait
        //  for the prefetch)
        TS(5);
        const f32x4 b_m1 = *reinterpret_cast<const f32x4 *>(a.b_merge1 + 16 * w + 4 * q);
        if (RMR_FUSED_STREAM_M1) load_m1();
        const InRegs next_in = fetch_inputs(it + gridDim.x);
        RMR_SYNC();
        TS(6);
        // ---- S4: merge_conv1 (K = 5 taps x 128 channels) -> x, bf16 channel-last in HBM ----
        {
            const int ncols = nch * a.T;
            const int ntiles = ABL(32) ? 0 : (ncols + 15) >> 4;
            uint16_t *xo = a.x + (size_t)chunk0 * a.T * 64 + 16 * w + 4 * q;
            const unsigned char *cat_r = s_cat + (size_t)q * a.cat_plane;
            auto col_of = [&](int c) { return c < ncols ? c : ncols - 1; };
            int c0 = s_col4[col_of(nn)], c1 = s_col4[col_of(nn + 16)];
            for (int tile = 0; tile < ntiles; tile += 2) {
                const int col0 = tile * 16 + nn, col1 = col0 + 16;
                const bool v0 = col0 < ncols, v1 = col1 < ncols;
                const unsigned char *r0 = cat_r + c0, *r1 = cat_r + c1;
                c0 = s_col4[col_of(col0 + 32)];
                c1 = s_col4[col_of(col1 + 32)];
                f32x4 acc0 = b_m1, acc1 = b_m1;
#if RMR_FUSED_S4_LEAN
                if (tile + 1 < ntiles) s4_pair_lean<F16, true>(Am1, r0, r1, acc0, acc1);  // wave-uniform
                else s4_pair_lean<F16, false>(Am1, r0, r1, acc0, acc1);
#else
                if (tile + 1 < ntiles) s4_pair<F16, true>(Am1, r0, r1, acc0, acc1);  // wave-uniform
                else s4_pair<F16, false>(Am1, r0, r1, acc0, acc1);
#endif
                if constexpr (WIN) {  // a column's row of x: its window's first row + its position, if the window has that output
                    auto put = [&](int col, const f32x4 &acc) {
                        const int ci = fdiv(col, a.d_T), t = col - ci * a.T;
                        if (t < s_win[4 * ci + 1]) {
                            const long long row = (((long long)s_win[4 * ci + 3] << 32) | (unsigned)s_win[4 * ci + 2]) + t;
                            *reinterpret_cast<uint2 *>(a.x + (size_t)row * 64 + 16 * w + 4 * q) = swish_pack<F16>(acc, ABL(64), 0.6931471805599453f);
                        }
                    };
                    if (v0) put(col0, acc0);
                    if (v1) put(col1, acc1);
                } else {
                    if (v0) *reinterpret_cast<uint2 *>(xo + (size_t)col0 * 64) = swish_pack<F16>(acc0, ABL(64), 0.6931471805599453f);
                    if (v1) *reinterpret_cast<uint2 *>(xo + (size_t)col1 * 64) = swish_pack<F16>(acc1, ABL(64), 0.6931471805599453f);
                }
            }
        }
        TS(7);
        if constexpr (WIN) RMR_SYNC();  // S4 read the window table that the next inputs' store rewrites
        store_inputs(next_in, it + gridDim.x);
        TS(8);
    }
    TS_FLUSH;
}

// Chunks per block iteration and the LDS image for them: the largest count whose image leaves room for two blocks per CU
// (RMR_FUSED_LDS_BUDGET) and whose input runs fit one element per thread (S0); a single chunk may take up to a whole CU's
// LDS (long chunk contexts).  Returns 0 when not even one chunk fits — such shapes run through the unfused bf16 kernels.
static int fused_front_plan(const rmr_model *m, int seq_w, int map_w, FusedArgs &a, int &total, int L_window = 0) {
    const int CG = (4 * m->desc.kmer_len + 7) / 8;
    if (L_window > 0) {  // the geometry of a position window of L_window samples (models/ConvLSTM_w_ref.py:41-50: 5, 5, 9 / 3 or 13 / 3, 5)
        a.L = L_window; a.P1 = a.L - 4; a.P2 = a.P1 - 4; a.P3 = (a.P2 - 9) / 3 + 1; a.T = a.P3 - 4;
    } else {
        a.L = m->L; a.P1 = m->P1; a.P2 = m->P2; a.P3 = m->P3; a.T = m->T;
    }
    auto up16 = [](int b) { return (b + 15) & ~15; };
    // RMR_FUSED_WAVES_EU blocks are resident per CU (one wave of each per SIMD): each gets its share of the 160 KB
    const int budget = ((160 * 1024) / RMR_FUSED_WAVES_EU - 256);
    total = 0;
    for (int cb = 8; cb >= 1; --cb) {
        if (cb * a.L > 1024 || cb * seq_w > 256 || cb * map_w > 256) continue;
        int off = 0;
        a.o_sig = off; off += 256 * 16;  // input regions: one element per thread
        a.o_seq = off; off += 256;
        a.o_map = off; off += 512;
        a.o_len = off; off += 16;
        a.o_tab = off; off += 64 * 16;  // the 16-byte one-hot piece of every pair of base codes
        a.o_win = off; off += cb * 16;  // (windows only: 16 B per virtual chunk)
        a.o_col3 = off; off += up16(cb * a.P3 * 8);  // per output column of S3 / S4: where its operand rows begin
        a.o_col4 = off; off += up16(cb * a.T * 4);
        a.o_sig1 = off; off += up16((cb * a.P1 + 8) * 8);
        a.o_sig2 = off; off += (cb * a.P2 + 4) * 32;
        a.o_seq1 = off; off += (cb * a.P1 + 4) * 32;
        a.o_oh = off;
        a.oh_plane = ((cb * a.L + 8 + 15) & ~15) * 16;
        a.cat_plane = (((cb * a.P3 + 1) * 5 + 15) & ~15) * 16;
        const int oh = CG * a.oh_plane, cat = 4 * a.cat_plane;
        off += oh > cat ? oh : cat;
        total = off;
        if (total <= budget || (cb == 1 && total <= 156 * 1024)) return cb;
    }
    return 0;
}

// Position windows for chunks that do not fit: 96 outputs per window (a multiple of 4, so that a window's first sample 3 * 96
// * win is float4-aligned) need 3 * 96 + 26 = 314 -> 316 samples; consecutive windows overlap by 28 samples (9 %).
static constexpr int FUSED_WINDOW_T = 96, FUSED_WINDOW_L = 316;

bool fused_front_supported(const rmr_model *m, int seq_w, int map_w) {
    if (m->desc.arch != RMR_ARCH_CONV_LSTM || m->desc.size != 64 || m->nparts != 1) return false;
    if ((m->desc.kmer_len != 9 && m->desc.kmer_len != 6) || m->front.kw1 != 5) return false;  // the instantiated k-mer lengths
    // sequence and mapping rows travel one element per thread: at most 256 bases + context / 255 + 1 mapping entries per chunk
    if (m->L % 4 || map_w < 2 || map_w > 256 || seq_w > 256) return false;
    if (seq_w < map_w - 1 + m->desc.kmer_len - 1) return false;
    if (m->fused.a_merge1 == nullptr) return false;
    FusedArgs probe;
    int total;
    // every limit the launcher enforces: whole chunks, or - chunks too long for a CU's LDS - position windows
    return fused_front_plan(m, seq_w, map_w, probe, total) >= 1 || fused_front_plan(m, seq_w, map_w, probe, total, FUSED_WINDOW_L) >= 1;
}

int launch_fused_front(rmr_model *m, const float *signal, const int8_t *seqs, int seq_w, const int16_t *maps, int map_w,
                       const int16_t *lens, int64_t n, uint16_t *x) {
    rmr_engine *e = m->eng;
    if (n <= 0) return 0;
    FusedArgs a;
    a.signal = signal; a.seqs = seqs; a.maps = maps; a.lens = lens;
    a.a_sig2 = reinterpret_cast<const uint4 *>(m->fused.a_sig2); a.a_seq1 = reinterpret_cast<const uint4 *>(m->fused.a_seq1);
    a.a_sig3 = reinterpret_cast<const uint4 *>(m->fused.a_sig3); a.a_seq2 = reinterpret_cast<const uint4 *>(m->fused.a_seq2);
    a.a_merge1 = reinterpret_cast<const uint4 *>(m->fused.a_merge1);
    a.w_sig1 = m->fused.w_sig1; a.b_sig1 = m->fused.b_sig1; a.b_sig2 = m->fused.b_sig2; a.b_seq1 = m->fused.b_seq1;
    a.b_sig3 = m->fused.b_sig3; a.b_seq2 = m->fused.b_seq2; a.b_merge1 = m->fused.b_merge1;
    a.x = x; a.n = n;
    a.seq_w = seq_w; a.map_w = map_w; a.maxlen = map_w - 1;
    int total = 0;
    int cb = fused_front_plan(m, seq_w, map_w, a, total);
    a.nwin = 1; a.Tw = m->T; a.L_total = m->L; a.T_total = m->T;
    const bool win = cb < 1 || tune_int("RMR_FUSED_WINDOWS", 0) != 0;  // (the knob: windows on shapes that would fit, for tests)
    if (win) {
        cb = fused_front_plan(m, seq_w, map_w, a, total, FUSED_WINDOW_L);
        a.Tw = FUSED_WINDOW_T;
        a.nwin = (m->T + a.Tw - 1) / a.Tw;
        if (n * a.nwin >= (1 << 24)) RMR_FAIL(RMR_ERR_INVALID, "fused front: %lld chunks x %d windows per launch (sub-batch too large)", (long long)n, a.nwin);
    }
    if (cb < 1) RMR_FAIL(RMR_ERR_INVALID, "fused front: one chunk of %d samples (sequence width %d) does not fit (%d B of LDS)", a.L, seq_w, total);
    a.cb = cb; a.lds_bytes = total;
    a.d_nwin = make_fastdiv(a.nwin); a.d_L4 = make_fastdiv(a.L >> 2); a.d_seqw = make_fastdiv(seq_w); a.d_mapw = make_fastdiv(map_w);
    a.d_L = make_fastdiv(a.L); a.d_P1 = make_fastdiv(a.P1); a.d_P3 = make_fastdiv(a.P3); a.d_T = make_fastdiv(a.T);
    a.d_maxlen = make_fastdiv(a.maxlen);
    auto magic = [](int d) { return (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); };  // exact for x * d < 2^32
    a.mg_ps2 = magic((((a.P2 + 15) >> 4) + 1) >> 1);
    a.mg_pq1 = magic((((a.P1 + 15) >> 4) + 1) >> 1);
    a.abl = abl_int("RMR_FUSED_ABLATE", 0);  // ignored unless built with -DRMR_TIMING_ABLATIONS
#ifdef RMR_TIMING_ABLATIONS
    const bool stage_clock = abl_int("RMR_FUSED_STAGE_CLOCK", 0) != 0;
    if (stage_clock) {
        a.abl |= 0x1000;
        static const unsigned long long zeros[4][16] = {};
        RMR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stage_clock), zeros, sizeof(zeros)));
    }
#endif
    // (4,4) and (2,3)-style k-mer contexts; bf16 or half operands
    auto pick = [&](auto winc) {
        constexpr bool W = decltype(winc)::value;
        return m->f16 ? (m->desc.kmer_len == 9 ? fused_front_kernel<9, true, W> : fused_front_kernel<6, true, W>)
                      : (m->desc.kmer_len == 9 ? fused_front_kernel<9, false, W> : fused_front_kernel<6, false, W>);
    };
    auto kern = win ? pick(std::true_type{}) : pick(std::false_type{});
    RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(kern)));
    const int64_t iters = (n * a.nwin + cb - 1) / cb;
    int64_t grid = (int64_t)e->num_cus * (2 * RMR_FUSED_WAVES_EU);
    if (grid > iters) grid = iters;
    ProfScope ps(e, K_FUSED_FRONT);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), (size_t)total, e->stream, a);
    RMR_HIP(hipGetLastError());
#ifdef RMR_TIMING_ABLATIONS
    if (stage_clock) {
        unsigned long long h[4][16];
        RMR_HIP(hipStreamSynchronize(e->stream));
        RMR_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stage_clock), sizeof(h)));
        static const char *names[10] = {"barrier top", "S1 (+A fetch)", "barrier", "S2", "A fetch + barrier", "S3", "bias, prefetch, barrier", "S4",
                                        "store inputs", "loop"};
        for (int w = 0; w < 4; ++w) {
            unsigned long long tot = 0;
            for (int i = 0; i < 10; ++i) tot += h[w][i];
            fprintf(stderr, "[stage clock] wave %d:", w);
            for (int i = 0; i < 10; ++i) fprintf(stderr, " %s %.1f%%;", names[i], 100.0 * (double)h[w][i] / (double)(tot ? tot : 1));
            fprintf(stderr, " total %.3g clocks over %lld blocks, cb %d\n", (double)tot, (long long)grid, cb);
        }
    }
#endif
    return 0;
}

}  // namespace rmr
