// k_lstm.hip — lstm1 (T steps) + lstm2 (ONE step) + fc head of ConvLSTM_w_ref on the fp32
// matrix cores, and the fc head of Conv_w_ref.
//
// Replaces models/ConvLSTM_w_ref.py:51-56:
//     z = swish(lstm1(z)[0]); z = flip(swish(lstm2(flip(z))[0])); z = z[-1]; z = fc(z)
// After the second flip, z[-1] is the FIRST output of lstm2 run on the reversed sequence:
// it depends only on swish(h1[T-1]) with zero initial state, so lstm2 is evaluated as a
// single step (gates = W_ih x + b_ih + b_hh; c = sig(i) tanh(g); h = sig(o) tanh(c)); the
// other T-1 steps of lstm2 never reach the output.  torch gate order i,f,g,o.
//
// One block = H/16 waves handles 16 chunks at a time (one 16-column MFMA tile).
// Wave w owns hidden units 16w..16w+15 for ALL four gates: its four 16x16 D tiles
// (i,f,g,o) put the four gates of one (hidden unit, chunk) in the same lane and register
// index, so the cell update is lane-local and c stays in registers for all T steps.
// W_ih and W_hh slices (2 * 4 * H/4 VGPRs) are register-resident for the block lifetime.
// h_t goes through LDS once per step (every wave needs all H values as its B operand):
// D fragment = 4 consecutive hidden units of one chunk = one ds_write_b128; B fragment =
// one ds_read_b128 per 16-wide k group, each feeding 4 MFMAs x 4 gates.
#include "rmr_internal.h"

namespace rmr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoid_f(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) {
    // 1 - 2/(1+e^{2x}); saturates correctly at +-inf, abs error ~1e-7
    return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * x));
}
__device__ __forceinline__ float swish_l(float x) { return x * sigmoid_f(x); }

struct LstmArgs {
    const float *x;  // [n][T][H] channel-last merge_conv1 output
    float *logits;   // [n][num_out]
    const float *a_ih1, *a_hh1, *b1, *a_ih2, *b2, *w_fc, *b_fc;
    int64_t n;
    int T, num_out;
};

template <int H>
__global__ __launch_bounds__(4 * H) void lstm_head_kernel(LstmArgs a) {
    constexpr int NW = H / 16;   // waves
    constexpr int KS = H / 4;    // MFMA k-steps per operand
    constexpr int G = H / 16;    // 16-wide k groups
    constexpr int RS = H + 4;    // padded LDS row
    __shared__ __attribute__((aligned(16))) float xbuf[2][16][RS];
    __shared__ __attribute__((aligned(16))) float hbuf[2][16][RS];
    __shared__ float part[NW][16][16];

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, q = lane >> 4, nn = lane & 15;

    float Aih[4][KS], Ahh[4][KS];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) {
        const float *pi = a.a_ih1 + ((size_t)(w * 4 + gt) * KS) * 64 + lane;
        const float *ph = a.a_hh1 + ((size_t)(w * 4 + gt) * KS) * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Aih[gt][s] = pi[(size_t)s * 64];
            Ahh[gt][s] = ph[(size_t)s * 64];
        }
    }
    f32x4 bias[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
        bias[gt] = *reinterpret_cast<const f32x4 *>(a.b1 + gt * H + 16 * w + 4 * q);

    // staging role of this thread: chunk row = tid / (H/4), 16-byte piece = tid % (H/4)
    const int st_row = tid / (H / 4), st_c4 = tid - st_row * (H / 4);

    const int64_t n_groups = (a.n + 15) / 16;
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t chunk0 = grp * 16;
        int64_t st_chunk = chunk0 + st_row;
        if (st_chunk >= a.n) st_chunk = a.n - 1;  // clamp ragged tail (results masked)
        const float4 *xsrc = reinterpret_cast<const float4 *>(a.x + (size_t)st_chunk * a.T * H) + st_c4;
        __syncthreads();  // previous group's LDS traffic is done
        *reinterpret_cast<float4 *>(&xbuf[0][st_row][4 * st_c4]) = xsrc[0];
        __syncthreads();

        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < a.T; ++t) {
            float4 xnext;
            const bool more = (t + 1 < a.T);
            if (more) xnext = xsrc[(size_t)(t + 1) * (H / 4)];
            f32x4 acc[4] = {bias[0], bias[1], bias[2], bias[3]};
            const float *xb = &xbuf[t & 1][nn][4 * q];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32x4 bx = *reinterpret_cast<const f32x4 *>(xb + 16 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int gt = 0; gt < 4; ++gt)
                        acc[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Aih[gt][g * 4 + j], bx[j], acc[gt], 0, 0, 0);
            }
            if (t > 0) {
                const float *hb = &hbuf[(t - 1) & 1][nn][4 * q];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const f32x4 bh = *reinterpret_cast<const f32x4 *>(hb + 16 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int gt = 0; gt < 4; ++gt)
                            acc[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ahh[gt][g * 4 + j], bh[j], acc[gt], 0, 0, 0);
                }
            }
            f32x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ig = sigmoid_f(acc[0][r]), fg = sigmoid_f(acc[1][r]);
                const float gg = tanh_f(acc[2][r]), og = sigmoid_f(acc[3][r]);
                c[r] = fg * c[r] + ig * gg;
                h[r] = og * tanh_f(c[r]);
            }
            *reinterpret_cast<f32x4 *>(&hbuf[t & 1][nn][16 * w + 4 * q]) = h;
            if (more) *reinterpret_cast<float4 *>(&xbuf[(t + 1) & 1][st_row][4 * st_c4]) = xnext;
            __syncthreads();
        }

        // ---- lstm2: one step on swish(h1[T-1]), gates i, g, o only (c0 = 0 kills f) ----
        f32x4 acc2[3];
#pragma unroll
        for (int gt = 0; gt < 3; ++gt)
            acc2[gt] = *reinterpret_cast<const f32x4 *>(a.b2 + gt * H + 16 * w + 4 * q);
        {
            const float *hb = &hbuf[(a.T - 1) & 1][nn][4 * q];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                f32x4 z = *reinterpret_cast<const f32x4 *>(hb + 16 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = swish_l(z[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int gt = 0; gt < 3; ++gt) {
                        const float aw = a.a_ih2[((size_t)(w * 3 + gt) * KS + g * 4 + j) * 64 + lane];
                        acc2[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, z[j], acc2[gt], 0, 0, 0);
                    }
            }
        }
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float c2 = sigmoid_f(acc2[0][r]) * tanh_f(acc2[1][r]);
            const float h2 = sigmoid_f(acc2[2][r]) * tanh_f(c2);
            y[r] = swish_l(h2);
        }
        // ---- fc: partial dot over this lane's 4 hidden units, reduce over q then waves ----
        for (int o = 0; o < a.num_out; ++o) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(a.w_fc + (size_t)o * H + 16 * w + 4 * q);
            float p = wv[0] * y[0] + wv[1] * y[1] + wv[2] * y[2] + wv[3] * y[3];
            p += __shfl_xor(p, 16);
            p += __shfl_xor(p, 32);
            if (q == 0) part[w][nn][o] = p;
        }
        __syncthreads();
        if (tid < 16 * a.num_out) {
            const int ch = tid / a.num_out, o = tid - ch * a.num_out;
            if (chunk0 + ch < a.n) {
                float s = a.b_fc[o];
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) s += part[ww][ch][o];
                a.logits[(size_t)(chunk0 + ch) * a.num_out + o] = s;
            }
        }
    }
}

template <int H>
static int launch_lstm_t(rmr_model *m, const float *x, int64_t n, float *logits) {
    rmr_engine *e = m->eng;
    LstmArgs a;
    a.x = x; a.logits = logits; a.n = n; a.T = m->T; a.num_out = m->desc.num_out;
    a.a_ih1 = m->lstm.a_ih1; a.a_hh1 = m->lstm.a_hh1; a.b1 = m->lstm.b1;
    a.a_ih2 = m->lstm.a_ih2; a.b2 = m->lstm.b2; a.w_fc = m->lstm.w_fc; a.b_fc = m->lstm.b_fc;
    const int64_t groups = (n + 15) / 16;
    int64_t grid = (int64_t)e->num_cus * 2;
    if (grid > groups) grid = groups;
    if (grid < 1) return 0;
    ProfScope ps(e, K_LSTM_HEAD);
    hipLaunchKernelGGL(lstm_head_kernel<H>, dim3((unsigned)grid), dim3(4 * H), 0, e->stream, a);
    RMR_HIP(hipGetLastError());
    return 0;
}

int launch_lstm_head(rmr_model *m, const float *x, int64_t n, float *logits) {
    switch (m->desc.size) {
        case 64: return launch_lstm_t<64>(m, x, n, logits);
        case 32: return launch_lstm_t<32>(m, x, n, logits);
        case 16: return launch_lstm_t<16>(m, x, n, logits);
    }
    RMR_FAIL(RMR_ERR_INVALID, "unsupported LSTM size %d", m->desc.size);
}

// ---- Conv_w_ref head: flatten (channel-major, models/Conv_w_ref.py:59) + fc (:60) --------
__global__ void fc_head_kernel(const float *m4, const float *w, const float *b, float *logits,
                               int64_t n, int size, int t4, int num_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * num_out) return;
    const int64_t ch = i / num_out;
    const int o = (int)(i - ch * num_out);
    const float *src = m4 + (size_t)ch * t4 * size;  // channel-last [t4][size]
    const float *wo = w + (size_t)o * size * t4;     // torch flatten index = c * t4 + t
    float s = b[o];
    for (int c = 0; c < size; ++c)
        for (int t = 0; t < t4; ++t) s += wo[c * t4 + t] * src[t * size + c];
    logits[i] = s;
}

int launch_fc_head(rmr_model *m, const float *m4, int64_t n, float *logits) {
    rmr_engine *e = m->eng;
    const int64_t total = n * m->desc.num_out;
    if (total == 0) return 0;
    ProfScope ps(e, K_FC_HEAD);
    hipLaunchKernelGGL(fc_head_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream,
                       m4, m->w_fc, m->b_fc, logits, n, m->desc.size, m->T4, m->desc.num_out);
    RMR_HIP(hipGetLastError());
    return 0;
}

}  // namespace rmr
