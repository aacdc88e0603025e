// Internal declarations shared by the translation units of libremora_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/remora_hip.h"

namespace rmr {

// ---- error plumbing -------------------------------------------------------------------
void set_error(const char *fmt, ...);
#define RMR_FAIL(code, ...)            \
    do {                               \
        ::rmr::set_error(__VA_ARGS__); \
        return (code);                 \
    } while (0)
#define RMR_HIP(expr)                                                                     \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            ::rmr::set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, \
                             __LINE__, #expr);                                            \
            return RMR_ERR_HIP;                                                           \
        }                                                                                 \
    } while (0)
#define RMR_TRY(expr)          \
    do {                       \
        int _rc = (expr);      \
        if (_rc != 0) return _rc; \
    } while (0)

// ---- kernel ids for the profiling table ------------------------------------------------
enum KernelId {
    K_ENCODE = 0,
    K_TRIM,
    K_MOVES,
    K_NORMALISE,
    K_GEOMETRY,
    K_FILL,
    K_FRONT_SIG,
    K_FRONT_SEQ,
    K_SEQ1_DENSE,
    K_CONV_SIG3,
    K_CONV_SEQ2,
    K_CONV_SEQ3,
    K_CONV_MERGE1,
    K_CONV_MERGE2,
    K_CONV_MERGE3,
    K_CONV_MERGE4,
    K_LSTM_HEAD,
    K_FC_HEAD,
    K_COUNT,
    K_MOTIF,
    K_VBZ,
    K_REFINE_BAND,
    K_REFINE_DP,
    K_REFINE_ROWWISE,
    K_FUSED_FRONT,
    K_RESCALE_Q,
    K_SIG3_FRONT,
    K_SEQ2_FRONT,
    K_NUM
};
const char *kernel_name(int id);

}  // namespace rmr

// ---- engine ------------------------------------------------------------------------------
struct rmr_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    // second stream + events for the two-stage sub-batch pipeline (front kernels of sub-batch
    // i+1 run under the matrix kernels of sub-batch i)
    hipStream_t aux = nullptr;
    hipEvent_t ev_in = nullptr, ev_front[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    hipEvent_t ev_handoff = nullptr;  // rmr_engine_wait_for: recorded on this engine's stream, waited for by another's
    int num_cus = 256;
    std::mutex mu;
    int64_t subbatch = 0;
    // RCCL communicator of rmr_comm_init (opaque ncclComm_t), this rank and the job size
    void *comm = nullptr;
    int comm_rank = 0, comm_world = 1;

    // grow-only device scratch arenas
    struct Arena {
        void *ptr = nullptr;
        size_t cap = 0;
    };
    Arena act;      // activations of the fused pipeline
    Arena staging;  // host<->device staging for RMR_MEM_HOST calls
    int ensure(Arena &a, size_t bytes);
    // pinned host bounce buffers (two slots) + copy-done events: large RMR_MEM_HOST batches are uploaded
    // sub-batch by sub-batch on the aux stream under the kernels of the previous sub-batch
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    hipEvent_t ev_h2d[2] = {nullptr, nullptr};
    int ensure_pinned(size_t bytes);
    // pinned staging of rmr_call_read (one read in, its logits out): its own small buffer, so that a single-read call
    // never re-allocates the two big slots above under an upload
    void *pin_call = nullptr;
    size_t pin_call_cap = 0;
    int ensure_pin_call(size_t bytes);

    // kernels whose dynamic-LDS limit was raised ON THIS DEVICE (hipFuncSetAttribute is per device: a flag per
    // process would skip the second engine of a multi-GPU process); guarded by `mu` like every launch
    std::vector<const void *> lds_attr_set;
    int allow_big_lds(const void *kernel, size_t bytes = 160 * 1024);  // `bytes`: the dynamic share (a kernel with static LDS asks for less)

    // profiling
    bool profiling = false;
    struct Rec {
        int id;
        hipEvent_t t0, t1;
    };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    double acc_ms[rmr::K_NUM] = {};
    int64_t acc_n[rmr::K_NUM] = {};
    int prof_begin(int id, hipEvent_t *t1, hipStream_t s);
    int prof_collect();
};

namespace rmr {
// Diagnostics (RMR_POISON=1): in front of EVERY kernel launch of the library a kernel fills the LDS and most of the vector
// registers of every CU with 0xFFFFFFFF (NaN as fp32 / bf16 / half, -1 as an integer).  LDS and registers keep what the
// previous workgroup left; a kernel that reads a word it never wrote normally finds the leftovers of its own kind (often
// the very values it would have written) and only fails next to OTHER kernels - e.g. another process's on the same GPU.
// With the poison such a read shows up in a single process: tests/test_gpu_poison.py.
void poison_before_launch(rmr_engine *e, hipStream_t s);
}  // namespace rmr

// RAII-less helper: brackets one launch with events when profiling is on
struct ProfScope {
    rmr_engine *e;
    hipEvent_t t1 = nullptr;
    bool on = false;
    hipStream_t s;
    ProfScope(rmr_engine *eng, int id, hipStream_t st = nullptr, bool use_st = false)
        : e(eng), s(use_st ? st : eng->stream) {
        rmr::poison_before_launch(e, s);
        if (e->profiling) on = (e->prof_begin(id, &t1, s) == 0);
    }
    ~ProfScope() {
        if (on) (void)hipEventRecord(t1, s);
    }
};

// ---- model -------------------------------------------------------------------------------
namespace rmr {

// One convolution executed on the MFMA path: out[oc][n] = sum_{tap,ic} W[oc][ic][tap] * in
struct ConvLayer {
    int ic = 0, oc = 0, kw = 0, stride = 1;
    float *apack = nullptr;  // device, fragment order [oc/16][kw*ic/4][64]
    float *wpack = nullptr;  // device, Winograd F(4,5) fragments U = G W: [oc/16][8 * ic/4][64] (k_wino.hip; fp32 5-tap stride-1 layers of 64 output channels)
    float *apack4 = nullptr; // device, streamed-kernel order [oc/16][kw*ic/16][64][4] (k_stream.hip; layers of networks with > 64 channels)
    float *apack16 = nullptr; // device, 16-bit A fragments [oc/16][ceil(kw*ic/32)][64 lanes] x 16 B, k = tap * ic + channel (k_stream16.hip)
    float *spack = nullptr;  // device, split-bf16 fragments [oc/16][steps][nparts][64] x 16 B (dtype != 0)
    float *bias = nullptr;   // device, folded bias [oc]
    int kid = 0;             // profiling id
    bool split_f16 = false;  // spack holds two IEEE half parts (dtype f16x3) instead of bf16 parts
};

struct FrontWeights {
    int kw1 = 5;               // kernel width of sig_conv1, sig_conv2, seq_conv1
    float *w_sig1 = nullptr;   // [kw1][4]
    float *b_sig1 = nullptr;   // [4]
    float *w_sig2 = nullptr;   // [kw1][4 ic][16 oc]
    float *b_sig2 = nullptr;   // [16]
    float *wt_seq1 = nullptr;  // [kw1][K][4 base][16 oc] == [kw1][EC][16] (dense seq_conv1)
    float *wt5_seq1 = nullptr; // [kw1][K][5][16]: gather table, row 4 = zeros (missing base)
    float *b_seq1 = nullptr;   // [16]
};

// bf16 A fragments of the fused front kernel (k_fused.hip): [oc/16][k-steps][64 lanes] x 16 B, k = tap * C + channel
// Activations travel scaled by log2(e) inside that kernel: sig_conv1 / seq_conv1 (raw inputs) have weights AND bias
// scaled, the other layers only the bias.
struct FusedWeights {
    float *a_sig2 = nullptr, *a_seq1 = nullptr, *a_sig3 = nullptr, *a_seq2 = nullptr, *a_merge1 = nullptr;
    float *w_sig1 = nullptr, *b_sig1 = nullptr, *b_sig2 = nullptr, *b_seq1 = nullptr, *b_sig3 = nullptr, *b_seq2 = nullptr,
          *b_merge1 = nullptr;
};

struct LstmWeights {
    float *a_ih1 = nullptr, *a_hh1 = nullptr;  // [H/16 waves][4 gates][H/4][64]
    float *b1 = nullptr;                       // [4H]  b_ih + b_hh
    float *a_ih2 = nullptr;                    // [H/16][3 gates i,g,o][H/4][64]
    float *b2 = nullptr;                       // [3H]  (i,g,o) b_ih + b_hh
    float *w_fc = nullptr, *b_fc = nullptr;    // [num_out][H], [num_out]
    // split-bf16 fragments (dtype != 0): [H/16][4 gates][H/32][nparts][64 lanes] x 16 B
    float *s_ih1 = nullptr, *s_hh1 = nullptr;
    // k_lstm_x16.hip (plain bf16, size 64): unit-major tiles [8 waves][2 tiles][2 k-steps][64 lanes] x 16 B, biases
    // [8][2][4 q][4 gates]; lstm2 with a zero f row
    float *x_ih = nullptr, *x_hh = nullptr, *x_ih2 = nullptr, *x_b1 = nullptr, *x_b2 = nullptr;
    float *xs_ih = nullptr, *xs_hh = nullptr, *xs_ih2 = nullptr;  // the same fragments as NP split parts (k_lstm_x16s.hip)
    // k_stream.hip (more than 64 hidden units, fp32): [H/16 waves][H/16 k groups][4 gates (lstm2: 3)][64 lanes][4]
    float *t_ih1 = nullptr, *t_hh1 = nullptr, *t_ih2 = nullptr;
    // lstm_small_kernel (k_lstm.hip; 64 hidden units, fp32, batches of a few hundred chunks): [4 waves][64 k in issue order][64 lanes],
    // lane l = gate l & 3 (lstm2: i, g, o, zero) of unit 16 w + (l >> 2)
    float *q_ih1 = nullptr, *q_hh1 = nullptr, *q_ih2 = nullptr;
    // k_stream16.hip (more than 64 hidden units, bf16 / f16): [H/16 waves][4 tiles][H/32 k-steps][64 lanes] x 16 B, row m of tile t of
    // wave w = (unit 16 w + 4 (m >> 2) + t, gate m & 3), pre-scaled; biases [H/16][4][4 q][4 gates]
    float *s16_ih = nullptr, *s16_hh = nullptr, *s16_ih2 = nullptr, *s16_b1 = nullptr, *s16_b2 = nullptr;
};

}  // namespace rmr

struct rmr_model {
    rmr_engine *eng = nullptr;
    rmr_model_desc desc{};  // desc.size: the channel count the kernels run at (engine.hip padded_size)
    int true_size = 0;      // the network's own `size` (model_params["size"]); channels beyond it carry zero weights
    int nparts = 0;  // 0: fp32 MFMA path; 1..3: bf16 MFMA with 1 / 2 / 3-part split operands
    bool split_f16 = false;  // dtype f16x3: nparts == 2 and the parts are IEEE half (hi, lo)
    bool f16 = false;  // dtype 4: nparts == 1 with IEEE-half operands in the fused kernels (k_fused.hip, k_lstm_x16.hip)
    std::vector<void *> dev_allocs;
    rmr::FrontWeights front;
    // conv_lstm: sig3, seq2, merge1;  conv_only: sig3, seq2, seq3, merge1..4
    rmr::ConvLayer sig3, seq2, seq3, merge1, merge2, merge3, merge4;
    rmr::LstmWeights lstm;
    rmr::FusedWeights fused;  // plain-bf16 ConvLSTM only
    float *w_fc = nullptr, *b_fc = nullptr;  // conv_only head: [num_out][size*3]
    // derived geometry
    int L = 0, P1 = 0, P2 = 0, P3 = 0, PQ2 = 0, T = 0, T2 = 0, T3 = 0, T4 = 0;
};

// ---- kernel launchers (defined in k_*.hip) ------------------------------------------------
namespace rmr {

int launch_encode(rmr_engine *e, int kb, int ka, const int8_t *seqs, int seq_w,
                  const int16_t *maps, int map_w, const int16_t *lens, int64_t n, int sig_len,
                  float *out);
int launch_trim(rmr_engine *e, int sb, int sa, int cb, int ca, int tsc, int8_t *seqs, int seq_w,
                int16_t *maps, int map_w, int16_t *lens, int64_t n);
int launch_moves(rmr_engine *e, const int8_t *mv_tag, int64_t mv_tag_len, int64_t sig_len,
                 int reverse, int64_t *q2s, int64_t *d_count);
int launch_moves_batch(rmr_engine *e, const int8_t *mv_tags, const int64_t *mv_off, const int64_t *sig_len,
                       const int64_t *seq_len, int64_t n, int check, int reverse, int64_t *q2s, int64_t *counts,
                       int32_t *status);
int launch_signal_range(rmr_engine *e, const int16_t *signal, const int64_t *start, const int64_t *len, int64_t n, int32_t *lo, int32_t *hi);
int launch_signal_hist(rmr_engine *e, const int16_t *signal, const int64_t *start, const int64_t *len, const int32_t *lo,
                       const int64_t *hist_off, int64_t n, unsigned int *hist);
int launch_assemble_lengths(rmr_engine *e, const int64_t *q2s, const int64_t *q2s_off, const int64_t *seq_len, int64_t n, int64_t *len_out);
int launch_assemble_reads(rmr_engine *e, const int16_t *signal, const int64_t *src_start, const int64_t *q2s, const int64_t *q2s_off,
                          const int64_t *sig_off, const int64_t *seq_off, int64_t n, int16_t *dacs, int64_t *s2s);
int launch_geometry(rmr_engine *e, const rmr_reads &d, int64_t n_chunks, const int32_t *chunk_read,
                    float *sig_out, int64_t total_sig, const int32_t *sig_read, int64_t *geo,
                    int *d_max_seq_len);
int launch_fill(rmr_engine *e, const rmr_reads &d, int64_t n_chunks, const int32_t *chunk_read,
                const float *sig, const int64_t *geo, float *signal, int8_t *seqs, int seq_w,
                int16_t *maps, int map_w, int16_t *lens, int64_t *rfb);
int launch_chunk_read(rmr_engine *e, const int64_t *focus_off, int64_t n_reads, int64_t n_chunks, int32_t *out);
int launch_count(rmr_engine *e, const float *logits, int64_t n, int num_out, int64_t *counts);
int launch_validation_tally(rmr_engine *e, const float *logits, const int64_t *labels, int64_t n, int km, int kf, const int *colmap,
                            int64_t *conf, float *win, uint8_t *pred, double *loss_sum);
int launch_vbz(rmr_engine *e, const uint8_t *svb, const int64_t *row_off, const int32_t *row_n, const int64_t *out_off,
               int64_t n_rows, int16_t *out, int32_t *status);
int launch_motif_focus(rmr_engine *e, const int8_t *seq, const int64_t *seq_off, int n_reads, const rmr_motif_set &ms,
                       int64_t *counts, const int64_t *foc_off, int64_t *focus);
int launch_motif(rmr_engine *e, const int8_t *seq, const int64_t *seq_off, int n_reads, int64_t total,
                 const rmr_motif_set &ms, uint8_t *flags);

// fused pipeline stages; all tensors channel-last in device scratch
int launch_front(rmr_model *m, hipStream_t st, const float *signal, const int8_t *seqs, int seq_w,
                 const int16_t *maps, int map_w, const int16_t *lens, int kb, int ka, int64_t n,
                 float *sig2, float *seq1 /* nullptr: skip seq path */);
int launch_seq1_dense(rmr_model *m, const float *enc, int64_t n, float *seq1);
// k_conv_front.hip: fp32 sig_conv3 / seq_conv2 with their producers (sig_conv1/2, seq_conv1) folded into the staging
bool conv_front_supported(const rmr_model *m, int kb, int ka, int seq_w, int map_w);
// the signal half alone with sig_conv2 on the matrix cores (both architectures, 5 or 11 taps): signal -> cat channels [0, 64)
bool sig3_front_mfma_supported(const rmr_model *m);
int launch_sig3_front_mfma(rmr_model *m, const float *signal, int64_t n, float *cat);
int launch_conv_front(rmr_model *m, const float *signal, const int8_t *seqs, int seq_w, const int16_t *maps, int map_w,
                      const int16_t *lens, int64_t n, float *cat);
int launch_conv(rmr_engine *e, const ConvLayer &c, const float *in, int in_row, int pin,
                float *out, int out_row, int out_coff, int pout, int64_t n);
int launch_lstm_head(rmr_model *m, const float *x, int64_t n, float *logits);
// k_wino.hip: the 5-tap stride-1 layers of 64 output channels as a Winograd F(4, 5) convolution (0.4 of the direct form's MFMAs)
bool conv_wino_supported(const ConvLayer &c, int pin, int pout);
int launch_conv_wino(rmr_engine *e, const ConvLayer &c, const float *in, int in_row, int pin, float *out, int out_row, int out_coff,
                     int pout, int64_t n);
// the stride-3 layers from 16 to 64 channels (sig_conv3, seq_conv2) as polyphase Winograd convolutions
bool conv_wino_s3_supported(const ConvLayer &c, int pin, int pout);
int launch_conv_wino_s3(rmr_engine *e, const ConvLayer &c, const float *in, int in_row, int pin, float *out, int out_row, int out_coff,
                        int pout, int64_t n);
// k_stream.hip: the same layers with the weights streamed from L2 (channel counts above 64, any multiple of 16 up to 256)
int launch_conv_stream(rmr_engine *e, const ConvLayer &c, const float *in, int in_row, int pin, float *out, int out_row, int out_coff,
                       int pout, int64_t n);
int launch_lstm_stream(rmr_model *m, const float *x, int64_t n, float *logits);
// k_stream16.hip: the same network in the 16-bit dtypes (activations 16-bit in HBM between the launches)
int launch_conv_stream16(rmr_model *m, const ConvLayer &c, const void *in, bool in16, int pin, uint16_t *out, int out_row, int out_coff, int pout,
                         int64_t n);
int launch_lstm_stream16(rmr_model *m, const uint16_t *x, int64_t n, float *logits);
int launch_lstm_head_split(rmr_model *m, const float *x, int64_t n, float *logits);
int launch_conv_split(rmr_engine *e, const ConvLayer &c, int np, const float *in, int in_row, int pin,
                      float *out, int out_row, int out_coff, int pout, int64_t n);
int launch_fc_head(rmr_model *m, const float *m4, int64_t n, float *logits);
// fused bf16 front (k_fused.hip): chunk arrays -> x bf16[n][T][64];  lstm on that tensor (k_lstm_bf16s.hip)
bool fused_front_supported(const rmr_model *m, int seq_w, int map_w);
int launch_fused_front(rmr_model *m, const float *signal, const int8_t *seqs, int seq_w, const int16_t *maps, int map_w,
                       const int16_t *lens, int64_t n, uint16_t *x);
bool lstm_x16s_supported(const rmr_model *m);
int launch_lstm_head_x16s(rmr_model *m, const float *x, int64_t n, float *logits);
int launch_lstm_head_x16(rmr_model *m, const uint16_t *x, int64_t n, float *logits);

// integer switch from the environment (DESIGN.md has the table of every name the product reads)
inline int tune_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
// the same for the knobs of the experiment build (make abl: -DRMR_TIMING_ABLATIONS); the shipped library returns the default
// without looking at the environment, so none of them is a configuration of the product
inline int abl_int(const char *name, int dflt) {
#ifdef RMR_TIMING_ABLATIONS
    return tune_int(name, dflt);
#else
    (void)name;
    return dflt;
#endif
}

// fast integer division by a small runtime constant (exact for 0 <= x < 2^24, d < 2^12)
struct FastDiv {
    int d;
    float inv;
};
inline FastDiv make_fastdiv(int d) { return FastDiv{d, 1.0f / (float)d}; }

}  // namespace rmr
