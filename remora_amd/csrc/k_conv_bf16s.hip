// k_conv_bf16s.hip — the implicit-GEMM convolution of k_conv.hip on the bf16 matrix cores
// (v_mfma_f32_16x16x32_bf16) with split operands (see k_lstm_bf16s.hip for the arithmetic:
// NP = 3 parts / 6 products is fp32-class, NP = 2 / 3 products ~2^-16, NP = 1 plain bf16).
// Same reference lines as k_conv.hip (models/ConvLSTM_w_ref.py:43,46,50).
//
// HBM tensors stay fp32 channel-last; the fp32 -> bf16-part conversion happens once per
// element while the block stages its input tile into LDS (so HBM traffic is unchanged and
// every wave reads ready-made B fragments).
//
// k-step = 32 contraction slots = 4 lane groups q x 8 bf16:
//   IC >= 32 : one tap, channels 32ks + 8q + j            (steps = KW * IC/32)
//   IC == 16 : two taps, tap = 2tp + (q>>1), channels 8(q&1) + j   (steps = ceil(KW/2);
//              the phantom tap of an odd KW has zero weights and reads a finite row)
// LDS image per part: planes of 16-byte slots (8 channels); plane stride a multiple of 16
// slots and an odd row stride => every 16-lane ds_read_b128 group hits 16 distinct slots.
#include "rmr_internal.h"
#include "rmr_math.h"

namespace rmr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// F16 (dtype f16x3, NP = 2): two IEEE half parts instead of bf16 parts (k_lstm_bf16s.hip, split_parts)
template <int NP, bool F16 = false>
__device__ __forceinline__ void split_parts_c(float x, unsigned (&p)[NP]) {
    if constexpr (F16) {
        static_assert(NP == 2, "the half split has two parts");
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        p[0] = (unsigned)__builtin_bit_cast(unsigned short, hi) << 16;
        p[1] = (unsigned)__builtin_bit_cast(unsigned short, lo) << 16;
    } else if (NP == 1) {
        const unsigned b = __float_as_uint(x);
        p[0] = (b + 0x7fffu + ((b >> 16) & 1u)) & 0xffff0000u;
    } else {
        float r = x;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const unsigned b = __float_as_uint(r);
            p[i] = (i + 1 < NP || NP == 3) ? (b & 0xffff0000u) : ((b + 0x7fffu + ((b >> 16) & 1u)) & 0xffff0000u);
            r -= __uint_as_float(p[i]);
        }
    }
}

template <int NP> struct ProdC;
template <> struct ProdC<1> { static constexpr int N = 1; static constexpr int A[1] = {0}; static constexpr int B[1] = {0}; };
template <> struct ProdC<2> { static constexpr int N = 3; static constexpr int A[3] = {0, 0, 1}; static constexpr int B[3] = {0, 1, 0}; };
template <> struct ProdC<3> { static constexpr int N = 6; static constexpr int A[6] = {0, 0, 1, 0, 2, 1}; static constexpr int B[6] = {0, 1, 0, 2, 0, 1}; };

struct ConvSArgs {
    const float *in;
    float *out;
    const uint4 *apack;  // [oc/16][steps][NP][64 lanes]
    const float *bias;
    int64_t n;
    int in_row, pin, pout, out_row, out_coff, cb;
    int plane;     // plane stride in 16-byte slots (multiple of 16)
    int part;      // part stride in slots
    FastDiv div_pout;
};

template <int IC, int KW, int STRIDE, int NP, bool F16>
__global__ __launch_bounds__(256) void conv_bf16s_kernel(ConvSArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 sm4[];
    constexpr bool PAIR = (IC == 16);            // two taps per k-step
    constexpr int KS = PAIR ? 1 : IC / 32;       // slots per row per plane
    constexpr int SLR = (KS % 2 == 0) ? KS + 1 : KS;   // odd row stride in slots
    constexpr int NPL = PAIR ? 2 : 4;            // planes
    constexpr int STEPS = PAIR ? (KW + 1) / 2 : KW * KS;
    using P = ProdC<NP>;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15;

    uint4 A[STEPS][NP];
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
        for (int p = 0; p < NP; ++p) A[s][p] = a.apack[(((size_t)w * STEPS + s) * NP + p) * 64 + lane];
    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(a.bias + 16 * w + 4 * q);

    // lane's plane / row offset inside a k-step
    const int l_plane = PAIR ? (q & 1) : q;
    const int l_rowoff = PAIR ? (q >> 1) : 0;

    // zero the whole image once: the phantom tap of an odd KW (zero weights) reads one row past
    // a chunk, which must be finite (0 * NaN = NaN) even before that row was ever staged
    for (int i = tid; i < NP * a.part; i += blockDim.x) sm4[i] = make_uint4(0, 0, 0, 0);

    const int64_t n_iters = (a.n + a.cb - 1) / a.cb;
    for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
        const int64_t chunk0 = it * a.cb;
        const int nch = (int)((a.n - chunk0) < a.cb ? (a.n - chunk0) : a.cb);
        RMR_SYNC();
        {   // stage: 8 channels (two float4) per item -> NP 16-byte slots
            constexpr int C8 = IC / 8;
            constexpr int UNR = 4;
            const int items = nch * a.pin * C8;
            const float4 *src = reinterpret_cast<const float4 *>(a.in + (size_t)chunk0 * a.pin * a.in_row);
            for (int base = tid; base < items; base += UNR * (int)blockDim.x) {
                float4 v0[UNR], v1[UNR];
                int dsto[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int i = base + u * (int)blockDim.x;
                    const int ii = i < items ? i : items - 1;
                    const int row = ii / C8, c8 = ii - row * C8;
                    v0[u] = src[2 * ii];
                    v1[u] = src[2 * ii + 1];
                    const int pl = PAIR ? c8 : (c8 & 3), sl = PAIR ? 0 : (c8 >> 2);
                    dsto[u] = i < items ? pl * a.plane + row * SLR + sl : NPL * a.plane;  // trash slot
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    unsigned e[8][NP];
                    split_parts_c<NP, F16>(v0[u].x, e[0]); split_parts_c<NP, F16>(v0[u].y, e[1]);
                    split_parts_c<NP, F16>(v0[u].z, e[2]); split_parts_c<NP, F16>(v0[u].w, e[3]);
                    split_parts_c<NP, F16>(v1[u].x, e[4]); split_parts_c<NP, F16>(v1[u].y, e[5]);
                    split_parts_c<NP, F16>(v1[u].z, e[6]); split_parts_c<NP, F16>(v1[u].w, e[7]);
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        sm4[(size_t)p * a.part + dsto[u]] =
                            make_uint4((e[0][p] >> 16) | e[1][p], (e[2][p] >> 16) | e[3][p],
                                       (e[4][p] >> 16) | e[5][p], (e[6][p] >> 16) | e[7][p]);
                }
            }
        }
        RMR_SYNC();
        const int ncols = nch * a.pout;
        const int ntiles = (ncols + 15) >> 4;
        for (int tile = 0; tile < ntiles; tile += 2) {
            int col0 = tile * 16 + nn, col1 = col0 + 16;
            const bool v0 = col0 < ncols, v1 = col1 < ncols;
            col0 = v0 ? col0 : ncols - 1;
            col1 = v1 ? col1 : ncols - 1;
            const int ch0 = (int)(((float)col0 + 0.5f) * a.div_pout.inv);
            const int ch1 = (int)(((float)col1 + 0.5f) * a.div_pout.inv);
            const int p0 = col0 - ch0 * a.pout, p1 = col1 - ch1 * a.pout;
            const uint4 *r0 = sm4 + (size_t)l_plane * a.plane + (size_t)(ch0 * a.pin + p0 * STRIDE + l_rowoff) * SLR;
            const uint4 *r1 = sm4 + (size_t)l_plane * a.plane + (size_t)(ch1 * a.pin + p1 * STRIDE + l_rowoff) * SLR;
            f32x4 acc0 = b4, acc1 = b4;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                // slot offset of step s relative to the lane's base row
                const int off = PAIR ? (2 * s) * SLR : (s / KS) * SLR + (s % KS);
                bf16x8 x0[NP], x1[NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    x0[p] = __builtin_bit_cast(bf16x8, r0[(size_t)p * a.part + off]);
                    x1[p] = __builtin_bit_cast(bf16x8, r1[(size_t)p * a.part + off]);
                }
#pragma unroll
                for (int pr = 0; pr < P::N; ++pr) {
                    if constexpr (F16) {
                        const f16x8 af = __builtin_bit_cast(f16x8, A[s][P::A[pr]]);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(f16x8, x0[P::B[pr]]), acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, __builtin_bit_cast(f16x8, x1[P::B[pr]]), acc1, 0, 0, 0);
                    } else {
                        const bf16x8 af = __builtin_bit_cast(bf16x8, A[s][P::A[pr]]);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, x0[P::B[pr]], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, x1[P::B[pr]], acc1, 0, 0, 0);
                    }
                }
            }
            if (v0) {
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = swish_f(acc0[r]);
                *reinterpret_cast<f32x4 *>(a.out + ((size_t)(chunk0 + ch0) * a.pout + p0) * a.out_row + a.out_coff + 16 * w + 4 * q) = y;
            }
            if (v1) {
                f32x4 y;
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = swish_f(acc1[r]);
                *reinterpret_cast<f32x4 *>(a.out + ((size_t)(chunk0 + ch1) * a.pout + p1) * a.out_row + a.out_coff + 16 * w + 4 * q) = y;
            }
        }
    }
}

template <int IC, int KW, int STRIDE, int NP, bool F16 = false>
static int launch_conv_s_t(rmr_engine *e, const ConvLayer &c, const float *in, int in_row, int pin, float *out,
                           int out_row, int out_coff, int pout, int64_t n) {
    constexpr bool PAIR = (IC == 16);
    constexpr int KS = PAIR ? 1 : IC / 32;
    constexpr int SLR = (KS % 2 == 0) ? KS + 1 : KS;
    constexpr int NPL = PAIR ? 2 : 4;
    const size_t chunk_bytes = (size_t)pin * SLR * 16 * NPL * NP;
    int cb = (int)((size_t)(112 * 1024) / chunk_bytes);
    if (cb < 1) cb = 1;
    if (cb > 8) cb = 8;
    if (cb >= 4) cb &= ~3;
    const int plane = ((cb * pin + 2) * SLR + 15) & ~15;  // +2 guard rows
    const int part = NPL * plane + 16;                    // + trash slot
    const size_t lds = (size_t)part * NP * 16;
    if (lds > 160 * 1024) RMR_FAIL(RMR_ERR_INVALID, "split conv layer needs %zu B of LDS", lds);
    ConvSArgs a;
    a.in = in; a.out = out; a.apack = reinterpret_cast<const uint4 *>(c.spack); a.bias = c.bias; a.n = n;
    a.in_row = in_row; a.pin = pin; a.pout = pout; a.out_row = out_row; a.out_coff = out_coff;
    a.cb = cb; a.plane = plane; a.part = part; a.div_pout = make_fastdiv(pout);
    const int64_t iters = (n + cb - 1) / cb;
    int64_t grid = (int64_t)e->num_cus * 2;
    if (grid > iters) grid = iters;
    if (grid < 1) return 0;
    auto kern = conv_bf16s_kernel<IC, KW, STRIDE, NP, F16>;
    RMR_TRY(e->allow_big_lds(reinterpret_cast<const void *>(kern)));
    ProfScope ps(e, c.kid);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * (c.oc / 16)), lds, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

int launch_conv_split(rmr_engine *e, const ConvLayer &c, int np, const float *in, int in_row, int pin, float *out,
                      int out_row, int out_coff, int pout, int64_t n) {
    if (in_row != c.ic) RMR_FAIL(RMR_ERR_INVALID, "conv input row %d != ic %d", in_row, c.ic);
#define RMR_CONVS_CASE(IC_, KW_, ST_)                                                                  \
    if (c.ic == IC_ && c.kw == KW_ && c.stride == ST_) {                                               \
        if (c.split_f16) return launch_conv_s_t<IC_, KW_, ST_, 2, true>(e, c, in, in_row, pin, out, out_row, out_coff, pout, n); \
        if (np == 1) return launch_conv_s_t<IC_, KW_, ST_, 1>(e, c, in, in_row, pin, out, out_row, out_coff, pout, n); \
        if (np == 2) return launch_conv_s_t<IC_, KW_, ST_, 2>(e, c, in, in_row, pin, out, out_row, out_coff, pout, n); \
        if (np == 3) return launch_conv_s_t<IC_, KW_, ST_, 3>(e, c, in, in_row, pin, out, out_row, out_coff, pout, n); \
    }
    RMR_CONVS_CASE(16, 9, 3)    // sig_conv3
    RMR_CONVS_CASE(16, 13, 3)   // seq_conv2
    RMR_CONVS_CASE(128, 5, 1)   // merge_conv1, size 64
    RMR_CONVS_CASE(64, 5, 1)    // merge_conv1, size 32
#undef RMR_CONVS_CASE
    RMR_FAIL(RMR_ERR_INVALID, "no split-bf16 conv kernel for ic=%d kw=%d stride=%d parts=%d", c.ic, c.kw, c.stride, np);
}

}  // namespace rmr
