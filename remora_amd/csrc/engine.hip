// engine.hip — host side of libremora_hip.so: engine/model lifetime, BatchNorm folding and
// MFMA-fragment packing of the weights, the C ABI entry points and the per-kernel HIP-event
// profiler.  See include/remora_hip.h for the contract of every entry point and the
// reference interface (file:line) each one replaces.
#include <dlfcn.h>

#include <cmath>
#include <cstring>
#include <memory>
#include <thread>
#include <algorithm>

#include "rmr_internal.h"
#include "rmr_geometry.h"

namespace rmr {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

static const char *k_names[K_NUM] = {
    "encode_kmers", "trim_chunk_context", "parse_moves", "normalise_signal", "chunk_geometry",
    "chunk_fill", "front_sig", "front_seq", "seq_conv1_dense", "conv_sig3", "conv_seq2", "conv_seq3",
    "conv_merge1", "conv_merge2", "conv_merge3", "conv_merge4", "lstm_head", "fc_head",
    "count_labels", "motif_scan", "vbz_decode", "refine_band", "refine_dp", "refine_dp_rowwise",
    "fused_front", "rescale_quantiles", "sig3_front", "seq2_front"};
const char *kernel_name(int id) { return (id >= 0 && id < K_NUM) ? k_names[id] : "?"; }

}  // namespace rmr

namespace rmr { void rccl_comm_free(void *comm); }  // defined with the collective at the end of this file

using namespace rmr;

// =========================================================================================
// engine
// =========================================================================================
int rmr_engine::ensure_pinned(size_t bytes) {
    if (bytes <= pinned_cap) return 0;
    if (pinned) {
        RMR_HIP(hipStreamSynchronize(aux));
        RMR_HIP(hipHostFree(pinned));
        pinned = nullptr;
        pinned_cap = 0;
    }
    hipError_t err = hipHostMalloc(&pinned, bytes, hipHostMallocDefault);
    if (err != hipSuccess) {
        pinned = nullptr;
        set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
        return RMR_ERR_HIP;
    }
    pinned_cap = bytes;
    return 0;
}

int rmr_engine::ensure_pin_call(size_t bytes) {
    if (bytes <= pin_call_cap) return 0;
    if (pin_call) {
        RMR_HIP(hipStreamSynchronize(stream));
        RMR_HIP(hipHostFree(pin_call));
        pin_call = nullptr;
        pin_call_cap = 0;
    }
    bytes = bytes + bytes / 2 + (1 << 16);
    hipError_t err = hipHostMalloc(&pin_call, bytes, hipHostMallocDefault);
    if (err != hipSuccess) {
        pin_call = nullptr;
        set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
        return RMR_ERR_HIP;
    }
    pin_call_cap = bytes;
    return 0;
}

int rmr_engine::ensure(Arena &a, size_t bytes) {
    if (bytes <= a.cap) return 0;
    if (a.ptr) {
        RMR_HIP(hipStreamSynchronize(stream));
        RMR_HIP(hipFree(a.ptr));
        a.ptr = nullptr;
        a.cap = 0;
    }
    size_t want = bytes + bytes / 8;
    hipError_t err = hipMalloc(&a.ptr, want);
    if (err != hipSuccess) {
        a.ptr = nullptr;
        set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(err));
        return RMR_ERR_NOMEM;
    }
    a.cap = want;
    return 0;
}

// ---- RMR_POISON (rmr_internal.h) -----------------------------------------------------------------------------------
namespace rmr {
__global__ __launch_bounds__(512) void poison_kernel(int lds_words) {
    extern __shared__ unsigned poison_lds[];
    for (int i = threadIdx.x; i < lds_words; i += 512) poison_lds[i] = 0xFFFFFFFFu;
    // v16 .. v255 of the block's 8 waves (2 per SIMD x 256 registers = the whole file of a SIMD) and s40 .. s95
    asm volatile("v_mov_b32 v16, -1\n\t"
                 "v_mov_b32 v17, -1\n\t"
                 "v_mov_b32 v18, -1\n\t"
                 "v_mov_b32 v19, -1\n\t"
                 "v_mov_b32 v20, -1\n\t"
                 "v_mov_b32 v21, -1\n\t"
                 "v_mov_b32 v22, -1\n\t"
                 "v_mov_b32 v23, -1\n\t"
                 "v_mov_b32 v24, -1\n\t"
                 "v_mov_b32 v25, -1\n\t"
                 "v_mov_b32 v26, -1\n\t"
                 "v_mov_b32 v27, -1\n\t"
                 "v_mov_b32 v28, -1\n\t"
                 "v_mov_b32 v29, -1\n\t"
                 "v_mov_b32 v30, -1\n\t"
                 "v_mov_b32 v31, -1\n\t"
                 "v_mov_b32 v32, -1\n\t"
                 "v_mov_b32 v33, -1\n\t"
                 "v_mov_b32 v34, -1\n\t"
                 "v_mov_b32 v35, -1\n\t"
                 "v_mov_b32 v36, -1\n\t"
                 "v_mov_b32 v37, -1\n\t"
                 "v_mov_b32 v38, -1\n\t"
                 "v_mov_b32 v39, -1\n\t"
                 "v_mov_b32 v40, -1\n\t"
                 "v_mov_b32 v41, -1\n\t"
                 "v_mov_b32 v42, -1\n\t"
                 "v_mov_b32 v43, -1\n\t"
                 "v_mov_b32 v44, -1\n\t"
                 "v_mov_b32 v45, -1\n\t"
                 "v_mov_b32 v46, -1\n\t"
                 "v_mov_b32 v47, -1\n\t"
                 "v_mov_b32 v48, -1\n\t"
                 "v_mov_b32 v49, -1\n\t"
                 "v_mov_b32 v50, -1\n\t"
                 "v_mov_b32 v51, -1\n\t"
                 "v_mov_b32 v52, -1\n\t"
                 "v_mov_b32 v53, -1\n\t"
                 "v_mov_b32 v54, -1\n\t"
                 "v_mov_b32 v55, -1\n\t"
                 "v_mov_b32 v56, -1\n\t"
                 "v_mov_b32 v57, -1\n\t"
                 "v_mov_b32 v58, -1\n\t"
                 "v_mov_b32 v59, -1\n\t"
                 "v_mov_b32 v60, -1\n\t"
                 "v_mov_b32 v61, -1\n\t"
                 "v_mov_b32 v62, -1\n\t"
                 "v_mov_b32 v63, -1\n\t"
                 "v_mov_b32 v64, -1\n\t"
                 "v_mov_b32 v65, -1\n\t"
                 "v_mov_b32 v66, -1\n\t"
                 "v_mov_b32 v67, -1\n\t"
                 "v_mov_b32 v68, -1\n\t"
                 "v_mov_b32 v69, -1\n\t"
                 "v_mov_b32 v70, -1\n\t"
                 "v_mov_b32 v71, -1\n\t"
                 "v_mov_b32 v72, -1\n\t"
                 "v_mov_b32 v73, -1\n\t"
                 "v_mov_b32 v74, -1\n\t"
                 "v_mov_b32 v75, -1\n\t"
                 "v_mov_b32 v76, -1\n\t"
                 "v_mov_b32 v77, -1\n\t"
                 "v_mov_b32 v78, -1\n\t"
                 "v_mov_b32 v79, -1\n\t"
                 "v_mov_b32 v80, -1\n\t"
                 "v_mov_b32 v81, -1\n\t"
                 "v_mov_b32 v82, -1\n\t"
                 "v_mov_b32 v83, -1\n\t"
                 "v_mov_b32 v84, -1\n\t"
                 "v_mov_b32 v85, -1\n\t"
                 "v_mov_b32 v86, -1\n\t"
                 "v_mov_b32 v87, -1\n\t"
                 "v_mov_b32 v88, -1\n\t"
                 "v_mov_b32 v89, -1\n\t"
                 "v_mov_b32 v90, -1\n\t"
                 "v_mov_b32 v91, -1\n\t"
                 "v_mov_b32 v92, -1\n\t"
                 "v_mov_b32 v93, -1\n\t"
                 "v_mov_b32 v94, -1\n\t"
                 "v_mov_b32 v95, -1\n\t"
                 "v_mov_b32 v96, -1\n\t"
                 "v_mov_b32 v97, -1\n\t"
                 "v_mov_b32 v98, -1\n\t"
                 "v_mov_b32 v99, -1\n\t"
                 "v_mov_b32 v100, -1\n\t"
                 "v_mov_b32 v101, -1\n\t"
                 "v_mov_b32 v102, -1\n\t"
                 "v_mov_b32 v103, -1\n\t"
                 "v_mov_b32 v104, -1\n\t"
                 "v_mov_b32 v105, -1\n\t"
                 "v_mov_b32 v106, -1\n\t"
                 "v_mov_b32 v107, -1\n\t"
                 "v_mov_b32 v108, -1\n\t"
                 "v_mov_b32 v109, -1\n\t"
                 "v_mov_b32 v110, -1\n\t"
                 "v_mov_b32 v111, -1\n\t"
                 "v_mov_b32 v112, -1\n\t"
                 "v_mov_b32 v113, -1\n\t"
                 "v_mov_b32 v114, -1\n\t"
                 "v_mov_b32 v115, -1\n\t"
                 "v_mov_b32 v116, -1\n\t"
                 "v_mov_b32 v117, -1\n\t"
                 "v_mov_b32 v118, -1\n\t"
                 "v_mov_b32 v119, -1\n\t"
                 "v_mov_b32 v120, -1\n\t"
                 "v_mov_b32 v121, -1\n\t"
                 "v_mov_b32 v122, -1\n\t"
                 "v_mov_b32 v123, -1\n\t"
                 "v_mov_b32 v124, -1\n\t"
                 "v_mov_b32 v125, -1\n\t"
                 "v_mov_b32 v126, -1\n\t"
                 "v_mov_b32 v127, -1\n\t"
                 "v_mov_b32 v128, -1\n\t"
                 "v_mov_b32 v129, -1\n\t"
                 "v_mov_b32 v130, -1\n\t"
                 "v_mov_b32 v131, -1\n\t"
                 "v_mov_b32 v132, -1\n\t"
                 "v_mov_b32 v133, -1\n\t"
                 "v_mov_b32 v134, -1\n\t"
                 "v_mov_b32 v135, -1\n\t"
                 "v_mov_b32 v136, -1\n\t"
                 "v_mov_b32 v137, -1\n\t"
                 "v_mov_b32 v138, -1\n\t"
                 "v_mov_b32 v139, -1\n\t"
                 "v_mov_b32 v140, -1\n\t"
                 "v_mov_b32 v141, -1\n\t"
                 "v_mov_b32 v142, -1\n\t"
                 "v_mov_b32 v143, -1\n\t"
                 "v_mov_b32 v144, -1\n\t"
                 "v_mov_b32 v145, -1\n\t"
                 "v_mov_b32 v146, -1\n\t"
                 "v_mov_b32 v147, -1\n\t"
                 "v_mov_b32 v148, -1\n\t"
                 "v_mov_b32 v149, -1\n\t"
                 "v_mov_b32 v150, -1\n\t"
                 "v_mov_b32 v151, -1\n\t"
                 "v_mov_b32 v152, -1\n\t"
                 "v_mov_b32 v153, -1\n\t"
                 "v_mov_b32 v154, -1\n\t"
                 "v_mov_b32 v155, -1\n\t"
                 "v_mov_b32 v156, -1\n\t"
                 "v_mov_b32 v157, -1\n\t"
                 "v_mov_b32 v158, -1\n\t"
                 "v_mov_b32 v159, -1\n\t"
                 "v_mov_b32 v160, -1\n\t"
                 "v_mov_b32 v161, -1\n\t"
                 "v_mov_b32 v162, -1\n\t"
                 "v_mov_b32 v163, -1\n\t"
                 "v_mov_b32 v164, -1\n\t"
                 "v_mov_b32 v165, -1\n\t"
                 "v_mov_b32 v166, -1\n\t"
                 "v_mov_b32 v167, -1\n\t"
                 "v_mov_b32 v168, -1\n\t"
                 "v_mov_b32 v169, -1\n\t"
                 "v_mov_b32 v170, -1\n\t"
                 "v_mov_b32 v171, -1\n\t"
                 "v_mov_b32 v172, -1\n\t"
                 "v_mov_b32 v173, -1\n\t"
                 "v_mov_b32 v174, -1\n\t"
                 "v_mov_b32 v175, -1\n\t"
                 "v_mov_b32 v176, -1\n\t"
                 "v_mov_b32 v177, -1\n\t"
                 "v_mov_b32 v178, -1\n\t"
                 "v_mov_b32 v179, -1\n\t"
                 "v_mov_b32 v180, -1\n\t"
                 "v_mov_b32 v181, -1\n\t"
                 "v_mov_b32 v182, -1\n\t"
                 "v_mov_b32 v183, -1\n\t"
                 "v_mov_b32 v184, -1\n\t"
                 "v_mov_b32 v185, -1\n\t"
                 "v_mov_b32 v186, -1\n\t"
                 "v_mov_b32 v187, -1\n\t"
                 "v_mov_b32 v188, -1\n\t"
                 "v_mov_b32 v189, -1\n\t"
                 "v_mov_b32 v190, -1\n\t"
                 "v_mov_b32 v191, -1\n\t"
                 "v_mov_b32 v192, -1\n\t"
                 "v_mov_b32 v193, -1\n\t"
                 "v_mov_b32 v194, -1\n\t"
                 "v_mov_b32 v195, -1\n\t"
                 "v_mov_b32 v196, -1\n\t"
                 "v_mov_b32 v197, -1\n\t"
                 "v_mov_b32 v198, -1\n\t"
                 "v_mov_b32 v199, -1\n\t"
                 "v_mov_b32 v200, -1\n\t"
                 "v_mov_b32 v201, -1\n\t"
                 "v_mov_b32 v202, -1\n\t"
                 "v_mov_b32 v203, -1\n\t"
                 "v_mov_b32 v204, -1\n\t"
                 "v_mov_b32 v205, -1\n\t"
                 "v_mov_b32 v206, -1\n\t"
                 "v_mov_b32 v207, -1\n\t"
                 "v_mov_b32 v208, -1\n\t"
                 "v_mov_b32 v209, -1\n\t"
                 "v_mov_b32 v210, -1\n\t"
                 "v_mov_b32 v211, -1\n\t"
                 "v_mov_b32 v212, -1\n\t"
                 "v_mov_b32 v213, -1\n\t"
                 "v_mov_b32 v214, -1\n\t"
                 "v_mov_b32 v215, -1\n\t"
                 "v_mov_b32 v216, -1\n\t"
                 "v_mov_b32 v217, -1\n\t"
                 "v_mov_b32 v218, -1\n\t"
                 "v_mov_b32 v219, -1\n\t"
                 "v_mov_b32 v220, -1\n\t"
                 "v_mov_b32 v221, -1\n\t"
                 "v_mov_b32 v222, -1\n\t"
                 "v_mov_b32 v223, -1\n\t"
                 "v_mov_b32 v224, -1\n\t"
                 "v_mov_b32 v225, -1\n\t"
                 "v_mov_b32 v226, -1\n\t"
                 "v_mov_b32 v227, -1\n\t"
                 "v_mov_b32 v228, -1\n\t"
                 "v_mov_b32 v229, -1\n\t"
                 "v_mov_b32 v230, -1\n\t"
                 "v_mov_b32 v231, -1\n\t"
                 "v_mov_b32 v232, -1\n\t"
                 "v_mov_b32 v233, -1\n\t"
                 "v_mov_b32 v234, -1\n\t"
                 "v_mov_b32 v235, -1\n\t"
                 "v_mov_b32 v236, -1\n\t"
                 "v_mov_b32 v237, -1\n\t"
                 "v_mov_b32 v238, -1\n\t"
                 "v_mov_b32 v239, -1\n\t"
                 "v_mov_b32 v240, -1\n\t"
                 "v_mov_b32 v241, -1\n\t"
                 "v_mov_b32 v242, -1\n\t"
                 "v_mov_b32 v243, -1\n\t"
                 "v_mov_b32 v244, -1\n\t"
                 "v_mov_b32 v245, -1\n\t"
                 "v_mov_b32 v246, -1\n\t"
                 "v_mov_b32 v247, -1\n\t"
                 "v_mov_b32 v248, -1\n\t"
                 "v_mov_b32 v249, -1\n\t"
                 "v_mov_b32 v250, -1\n\t"
                 "v_mov_b32 v251, -1\n\t"
                 "v_mov_b32 v252, -1\n\t"
                 "v_mov_b32 v253, -1\n\t"
                 "v_mov_b32 v254, -1\n\t"
                 "v_mov_b32 v255, -1\n\t"
                 "s_mov_b32 s40, -1\n\t"
                 "s_mov_b32 s41, -1\n\t"
                 "s_mov_b32 s42, -1\n\t"
                 "s_mov_b32 s43, -1\n\t"
                 "s_mov_b32 s44, -1\n\t"
                 "s_mov_b32 s45, -1\n\t"
                 "s_mov_b32 s46, -1\n\t"
                 "s_mov_b32 s47, -1\n\t"
                 "s_mov_b32 s48, -1\n\t"
                 "s_mov_b32 s49, -1\n\t"
                 "s_mov_b32 s50, -1\n\t"
                 "s_mov_b32 s51, -1\n\t"
                 "s_mov_b32 s52, -1\n\t"
                 "s_mov_b32 s53, -1\n\t"
                 "s_mov_b32 s54, -1\n\t"
                 "s_mov_b32 s55, -1\n\t"
                 "s_mov_b32 s56, -1\n\t"
                 "s_mov_b32 s57, -1\n\t"
                 "s_mov_b32 s58, -1\n\t"
                 "s_mov_b32 s59, -1\n\t"
                 "s_mov_b32 s60, -1\n\t"
                 "s_mov_b32 s61, -1\n\t"
                 "s_mov_b32 s62, -1\n\t"
                 "s_mov_b32 s63, -1\n\t"
                 "s_mov_b32 s64, -1\n\t"
                 "s_mov_b32 s65, -1\n\t"
                 "s_mov_b32 s66, -1\n\t"
                 "s_mov_b32 s67, -1\n\t"
                 "s_mov_b32 s68, -1\n\t"
                 "s_mov_b32 s69, -1\n\t"
                 "s_mov_b32 s70, -1\n\t"
                 "s_mov_b32 s71, -1\n\t"
                 "s_mov_b32 s72, -1\n\t"
                 "s_mov_b32 s73, -1\n\t"
                 "s_mov_b32 s74, -1\n\t"
                 "s_mov_b32 s75, -1\n\t"
                 "s_mov_b32 s76, -1\n\t"
                 "s_mov_b32 s77, -1\n\t"
                 "s_mov_b32 s78, -1\n\t"
                 "s_mov_b32 s79, -1\n\t"
                 "s_mov_b32 s80, -1\n\t"
                 "s_mov_b32 s81, -1\n\t"
                 "s_mov_b32 s82, -1\n\t"
                 "s_mov_b32 s83, -1\n\t"
                 "s_mov_b32 s84, -1\n\t"
                 "s_mov_b32 s85, -1\n\t"
                 "s_mov_b32 s86, -1\n\t"
                 "s_mov_b32 s87, -1\n\t"
                 "s_mov_b32 s88, -1\n\t"
                 "s_mov_b32 s89, -1\n\t"
                 "s_mov_b32 s90, -1\n\t"
                 "s_mov_b32 s91, -1\n\t"
                 "s_mov_b32 s92, -1\n\t"
                 "s_mov_b32 s93, -1\n\t"
                 "s_mov_b32 s94, -1\n\t"
                 "s_mov_b32 s95, -1\n\t"
                 
                 :
                 :
                 : "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95");
    __syncthreads();
    if (poison_lds[threadIdx.x] != 0xFFFFFFFFu) __builtin_trap();  // (keeps the stores alive)
}

void poison_before_launch(rmr_engine *e, hipStream_t s) {
    static const int on = tune_int("RMR_POISON", 0);
    if (!on) return;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL(poison_kernel, dim3((unsigned)e->num_cus * 2), dim3(512), (size_t)160 * 1024, s, 160 * 256);
}
}  // namespace rmr

int rmr_engine::allow_big_lds(const void *kernel, size_t bytes) {
    for (const void *k : lds_attr_set)
        if (k == kernel) return 0;
    RMR_HIP(hipSetDevice(device));
    RMR_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    lds_attr_set.push_back(kernel);
    return 0;
}

int rmr_engine::prof_begin(int id, hipEvent_t *t1, hipStream_t s) {
    hipEvent_t ev[2];
    for (int k = 0; k < 2; ++k) {
        if (!pool.empty()) {
            ev[k] = pool.back();
            pool.pop_back();
        } else if (hipEventCreate(&ev[k]) != hipSuccess) {
            return -1;
        }
    }
    if (hipEventRecord(ev[0], s) != hipSuccess) return -1;
    recs.push_back(Rec{id, ev[0], ev[1]});
    *t1 = ev[1];
    return 0;
}

int rmr_engine::prof_collect() {
    if (recs.empty()) return 0;
    RMR_HIP(hipStreamSynchronize(stream));
    if (aux) RMR_HIP(hipStreamSynchronize(aux));
    for (auto &r : recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.t0, r.t1) == hipSuccess) {
            acc_ms[r.id] += ms;
            acc_n[r.id] += 1;
        }
        pool.push_back(r.t0);
        pool.push_back(r.t1);
    }
    recs.clear();
    return 0;
}

extern "C" {

const char *rmr_last_error(void) { return g_err.c_str(); }
const char *rmr_version(void) { return "remora_hip 0.5 (gfx950)"; }  // 0.5: rmr_call_read, rmr_engine_wait_for

int rmr_engine_create(int device, void *stream, int flags, rmr_engine **out) {
    if (!out) RMR_FAIL(RMR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t err = hipGetDeviceCount(&ndev);
    if (err != hipSuccess || ndev == 0) {
        set_error("no HIP device available (%s)", hipGetErrorString(err));
        return RMR_ERR_HIP;
    }
    if (device < 0 || device >= ndev) RMR_FAIL(RMR_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    RMR_HIP(hipSetDevice(device));
    std::unique_ptr<rmr_engine> e(new rmr_engine());
    e->device = device;
    hipDeviceProp_t prop;
    RMR_HIP(hipGetDeviceProperties(&prop, device));
    e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (flags & RMR_ENGINE_USE_STREAM) {
        e->stream = reinterpret_cast<hipStream_t>(stream);
    } else {
        // an engine with a stream of its own is a helper beside the model's engine (chunk extraction, ingest decodes): short
        // kernels whose results somebody waits for - the highest priority the device offers, so that they are dispatched
        // ahead of queued work of the long model kernels wherever the hardware has a choice
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        if (hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, prio_hi) != hipSuccess)
            RMR_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        e->owns_stream = true;
    }
    RMR_HIP(hipStreamCreateWithFlags(&e->aux, hipStreamNonBlocking));
    RMR_HIP(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
    RMR_HIP(hipEventCreateWithFlags(&e->ev_handoff, hipEventDisableTiming));
    for (int k = 0; k < 2; ++k) {
        RMR_HIP(hipEventCreateWithFlags(&e->ev_front[k], hipEventDisableTiming));
        RMR_HIP(hipEventCreateWithFlags(&e->ev_done[k], hipEventDisableTiming));
        RMR_HIP(hipEventCreateWithFlags(&e->ev_h2d[k], hipEventDisableTiming));
    }
    *out = e.release();
    return 0;
}

void rmr_engine_destroy(rmr_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    if (e->aux) { (void)hipStreamSynchronize(e->aux); (void)hipStreamDestroy(e->aux); }
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    if (e->ev_handoff) (void)hipEventDestroy(e->ev_handoff);
    for (int k = 0; k < 2; ++k) {
        if (e->ev_front[k]) (void)hipEventDestroy(e->ev_front[k]);
        if (e->ev_done[k]) (void)hipEventDestroy(e->ev_done[k]);
        if (e->ev_h2d[k]) (void)hipEventDestroy(e->ev_h2d[k]);
    }
    if (e->comm) {
        rmr::rccl_comm_free(e->comm);
        e->comm = nullptr;
    }
    if (e->pinned) (void)hipHostFree(e->pinned);
    if (e->pin_call) (void)hipHostFree(e->pin_call);
    for (auto &r : e->recs) { (void)hipEventDestroy(r.t0); (void)hipEventDestroy(r.t1); }
    for (auto ev : e->pool) (void)hipEventDestroy(ev);
    if (e->act.ptr) (void)hipFree(e->act.ptr);
    if (e->staging.ptr) (void)hipFree(e->staging.ptr);
    if (e->owns_stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int rmr_engine_synchronize(rmr_engine *e) {
    if (!e) RMR_FAIL(RMR_ERR_INVALID, "engine is NULL");
    RMR_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

int rmr_engine_wait_for(rmr_engine *waiter, rmr_engine *producer) {
    if (!waiter || !producer) RMR_FAIL(RMR_ERR_INVALID, "engine is NULL");
    if (waiter == producer || waiter->stream == producer->stream) return 0;
    if (waiter->device != producer->device) RMR_FAIL(RMR_ERR_INVALID, "engines on different devices");
    {
        std::lock_guard<std::mutex> lk(producer->mu);
        RMR_HIP(hipSetDevice(producer->device));
        RMR_HIP(hipEventRecord(producer->ev_handoff, producer->stream));
    }
    std::lock_guard<std::mutex> lk(waiter->mu);
    RMR_HIP(hipStreamWaitEvent(waiter->stream, producer->ev_handoff, 0));
    return 0;
}

int rmr_engine_set_subbatch(rmr_engine *e, int64_t chunks) {
    if (!e || chunks < 0) RMR_FAIL(RMR_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    e->subbatch = chunks;
    return 0;
}

int rmr_profile_enable(rmr_engine *e, int on) {
    if (!e) RMR_FAIL(RMR_ERR_INVALID, "engine is NULL");
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_TRY(e->prof_collect());
    e->profiling = on != 0;
    return 0;
}
int rmr_profile_reset(rmr_engine *e) {
    if (!e) RMR_FAIL(RMR_ERR_INVALID, "engine is NULL");
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_TRY(e->prof_collect());
    for (int i = 0; i < K_NUM; ++i) { e->acc_ms[i] = 0; e->acc_n[i] = 0; }
    return 0;
}
int rmr_profile_num_kernels(void) { return K_NUM; }
const char *rmr_profile_kernel_name(int id) { return kernel_name(id); }
int rmr_profile_get(rmr_engine *e, int id, double *total_ms, int64_t *launches) {
    if (!e || id < 0 || id >= K_NUM) RMR_FAIL(RMR_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_TRY(e->prof_collect());
    if (total_ms) *total_ms = e->acc_ms[id];
    if (launches) *launches = e->acc_n[id];
    return 0;
}

}  // extern "C"

// =========================================================================================
// model: parse the canonical blob, fold BN, pack
// =========================================================================================
namespace {

struct ConvSpec { int ic, oc, kw, stride; };

struct Folded {
    ConvSpec s;
    std::vector<float> w;  // [oc][ic][kw] folded
    std::vector<float> b;  // [oc]
};

size_t conv_count(const ConvSpec &s) { return (size_t)s.oc * s.ic * s.kw + 5 * (size_t)s.oc; }

std::vector<ConvSpec> conv_specs(const rmr_model_desc &d) {
    const int sz = d.size, ec = 4 * d.kmer_len;
    if (d.arch == RMR_ARCH_CONV_LSTM)
        return {{1, 4, 5, 1}, {4, 16, 5, 1}, {16, sz, 9, 3}, {ec, 16, 5, 1}, {16, sz, 13, 3}, {2 * sz, sz, 5, 1}};
    return {{1, 4, 11, 1}, {4, 16, 11, 1}, {16, sz, 9, 3}, {ec, 16, 11, 1}, {16, 32, 11, 1},
            {32, sz, 9, 3}, {2 * sz, sz, 5, 1}, {sz, sz, 5, 1}, {sz, sz, 3, 2}, {sz, sz, 3, 2}};
}

// ---- split-bf16 helpers (host twins of split_parts/pack2 in k_lstm_bf16s.hip) -------------
static inline uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
static inline uint32_t rne_bf16(uint32_t b) { return (b + 0x7fffu + ((b >> 16) & 1u)) & 0xffff0000u; }
// one fp32 value as the 16-bit operand of the fused kernels, in the HIGH half of the returned word (as rne_bf16 returns
// it): bf16, or IEEE half (round to nearest even; the compiler's conversion)
static inline uint32_t to_op16(float v, bool f16) {
    if (!f16) {
        uint32_t b;
        memcpy(&b, &v, 4);
        return rne_bf16(b);
    }
    const _Float16 h = (_Float16)v;
    uint16_t hb;
    memcpy(&hb, &h, 2);
    return (uint32_t)hb << 16;
}
static void split_parts_host(float x, int np, uint32_t *p, bool f16 = false) {
    if (f16) {  // dtype f16x3: hi = half(x), lo = half(x - hi) (split_parts<2, true> in k_lstm_bf16s.hip)
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        uint16_t hb, lb;
        memcpy(&hb, &hi, 2);
        memcpy(&lb, &lo, 2);
        p[0] = (uint32_t)hb << 16;
        p[1] = (uint32_t)lb << 16;
        return;
    }
    if (np == 1) { p[0] = rne_bf16(f2u(x)); return; }
    float r = x;
    for (int i = 0; i < np; ++i) {
        const uint32_t b = f2u(r);
        p[i] = (i + 1 < np || np == 3) ? (b & 0xffff0000u) : rne_bf16(b);
        r -= u2f(p[i]);
    }
}

// [rows][K] row-major fp32 (row = gates[gi]*H + 16*wv + m) -> bf16x8 A fragments
// [H/16 waves][ngates][K/32][np][64 lanes][4 dwords]; lane (q, m) holds k = 32ks + 8q + j
std::vector<float> pack_split_a(const std::vector<float> &w, int K, int nw, const int *rowbase, int ngates, int np, bool f16 = false) {
    const int KS32 = K / 32;
    std::vector<uint32_t> out((size_t)nw * ngates * KS32 * np * 64 * 4);
    for (int wv = 0; wv < nw; ++wv)
        for (int gi = 0; gi < ngates; ++gi)
            for (int ks = 0; ks < KS32; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int q = lane >> 4, mm = lane & 15;
                    const int row = rowbase[gi] + 16 * wv + mm;
                    uint32_t parts[8][3];
                    for (int j = 0; j < 8; ++j) split_parts_host(w[(size_t)row * K + 32 * ks + 8 * q + j], np, parts[j], f16);
                    for (int p = 0; p < np; ++p)
                        for (int i = 0; i < 4; ++i)
                            out[(((((size_t)wv * ngates + gi) * KS32 + ks) * np + p) * 64 + lane) * 4 + i] =
                                (parts[2 * i][p] >> 16) | parts[2 * i + 1][p];
                }
    std::vector<float> f(out.size());
    memcpy(f.data(), out.data(), out.size() * 4);
    return f;
}

// The channel count the kernels run a network of `size` channels at (models/ConvLSTM_w_ref.py:11-37 and Conv_w_ref.py:11-42
// are parametric in `size`, the CLI takes any int: src/remora/parsers.py:858-862).  Up to 64 the kernels with register-resident
// weight slices exist for 16 / 32 / 64; above, the streamed-weight kernels (k_stream.hip) take any multiple of 16 up to 256.
// Channels between `size` and the padded count carry zero weights and zero bias: swish(0) = 0 and an LSTM unit with zero
// weights stays at c = h = 0 exactly, and a zero product added to an fp32 sum leaves it unchanged - the logits are those of
// the unpadded network (pad_model_blob below; rmr_model_pad_weights exposes the transform).
int padded_size(int size, int dtype = 0) {
    if (size <= 16) return 16;
    if (size <= 32) return 32;
    if (size <= 64) return 64;
    return dtype == 0 ? (size + 15) & ~15 : (size + 31) & ~31;  // the 16-bit MFMA takes K in steps of 32 (k_stream16.hip)
}
constexpr int kMaxPaddedSize = 256;

bool desc_ok(const rmr_model_desc &d) {
    if (d.arch != RMR_ARCH_CONV_LSTM && d.arch != RMR_ARCH_CONV_ONLY) return false;
    if (d.size < 1 || d.dtype < 0 || d.dtype > 5 || padded_size(d.size, d.dtype) > kMaxPaddedSize) return false;
    const int sp = padded_size(d.size, d.dtype);
    if (d.kmer_len < 1 || d.kmer_len > 64) return false;
    if (d.num_out < 1 || d.num_out > 16) return false;
    if (d.dtype < 0 || d.dtype > 5) return false;  // 5 = f16x3: two-part IEEE half split on the unfused kernels
    if (d.dtype == 4 && sp <= 64 && (sp != 64 || (d.kmer_len != 9 && d.kmer_len != 6))) return false;  // half up to 64 channels: the fused kernels only
    if (d.dtype != 0 && (d.arch != RMR_ARCH_CONV_LSTM || sp % 32)) return false;
    if (d.dtype != 0 && sp > 64 && d.dtype != 1 && d.dtype != 4) return false;  // above 64 channels: fp32, bf16 or f16 (the split dtypes stop at 64)
    return true;
}

// The canonical blob (include/remora_hip.h, rmr_model_create) of the same network with `sp` channels where `d` has d.size:
// zero weights / bias for the added output channels (BatchNorm of an added channel: gamma 1, beta 0, mean 0, var 1 - it folds
// to weight 0, bias 0), zero columns for the added input channels; merge_conv1 reads cat = [signal branch | sequence branch],
// so its input channel sz + c moves to sp + c.
std::vector<float> pad_model_blob(const rmr_model_desc &d, const float *w, int sp) {
    const int sz = d.size;
    rmr_model_desc pd = d;
    pd.size = sp;
    const std::vector<ConvSpec> ts = conv_specs(d), ps = conv_specs(pd);
    size_t total = 0;
    for (auto &s : ps) total += conv_count(s);
    const size_t H = sz, HP = sp;
    if (d.arch == RMR_ARCH_CONV_LSTM) total += 2 * (2 * 4 * HP * HP + 2 * 4 * HP) + (size_t)d.num_out * HP + d.num_out;
    else total += (size_t)d.num_out * HP * 3 + d.num_out;
    std::vector<float> o(total, 0.0f);
    const float *p = w;
    float *q = o.data();
    const size_t merge1 = d.arch == RMR_ARCH_CONV_LSTM ? 5 : 6;
    for (size_t li = 0; li < ts.size(); ++li) {
        const ConvSpec &t = ts[li], &u = ps[li];
        for (int oc = 0; oc < t.oc; ++oc)
            for (int ic = 0; ic < t.ic; ++ic) {
                const int icp = (li == merge1 && ic >= sz) ? sp + (ic - sz) : ic;
                memcpy(q + ((size_t)oc * u.ic + icp) * u.kw, p + ((size_t)oc * t.ic + ic) * t.kw, (size_t)t.kw * sizeof(float));
            }
        p += (size_t)t.oc * t.ic * t.kw;
        q += (size_t)u.oc * u.ic * u.kw;
        for (int part = 0; part < 5; ++part) {  // bias, gamma, beta, mean, var
            memcpy(q, p, (size_t)t.oc * sizeof(float));
            if (part == 1 || part == 4)
                for (int oc = t.oc; oc < u.oc; ++oc) q[oc] = 1.0f;
            p += t.oc;
            q += u.oc;
        }
    }
    if (d.arch == RMR_ARCH_CONV_LSTM) {
        for (int l = 0; l < 2; ++l) {
            for (int m = 0; m < 2; ++m) {  // weight_ih, weight_hh: [4H][H], row = gate * H + unit
                for (int g = 0; g < 4; ++g)
                    for (size_t r = 0; r < H; ++r) memcpy(q + ((size_t)g * HP + r) * HP, p + ((size_t)g * H + r) * H, H * sizeof(float));
                p += 4 * H * H;
                q += 4 * HP * HP;
            }
            for (int m = 0; m < 2; ++m) {  // bias_ih, bias_hh: [4H]
                for (int g = 0; g < 4; ++g) memcpy(q + (size_t)g * HP, p + (size_t)g * H, H * sizeof(float));
                p += 4 * H;
                q += 4 * HP;
            }
        }
        for (int oo = 0; oo < d.num_out; ++oo) memcpy(q + (size_t)oo * HP, p + (size_t)oo * H, H * sizeof(float));
        p += (size_t)d.num_out * H;
        q += (size_t)d.num_out * HP;
    } else {  // fc over flatten([size][3]): index c * 3 + t, channels first - the added channels sit behind the real ones
        for (int oo = 0; oo < d.num_out; ++oo) memcpy(q + (size_t)oo * HP * 3, p + (size_t)oo * H * 3, H * 3 * sizeof(float));
        p += (size_t)d.num_out * H * 3;
        q += (size_t)d.num_out * HP * 3;
    }
    memcpy(q, p, (size_t)d.num_out * sizeof(float));
    return o;
}

Folded fold(const ConvSpec &s, const float *&p) {
    Folded f;
    f.s = s;
    const size_t nw = (size_t)s.oc * s.ic * s.kw;
    const float *w = p; p += nw;
    const float *b = p; p += s.oc;
    const float *g = p; p += s.oc;
    const float *beta = p; p += s.oc;
    const float *mean = p; p += s.oc;
    const float *var = p; p += s.oc;
    f.w.resize(nw);
    f.b.resize(s.oc);
    for (int o = 0; o < s.oc; ++o) {
        // eval-mode BatchNorm1d, eps = 1e-5 (torch default; the reference folds the same
        // way for its Dorado export, src/remora/model_util.py:199-221)
        const double sc = (double)g[o] / std::sqrt((double)var[o] + 1e-5);
        for (size_t i = 0; i < (size_t)s.ic * s.kw; ++i)
            f.w[(size_t)o * s.ic * s.kw + i] = (float)((double)w[(size_t)o * s.ic * s.kw + i] * sc);
        f.b[o] = (float)(((double)b[o] - (double)mean[o]) * sc + (double)beta[o]);
    }
    return f;
}

int upload(rmr_model *m, const std::vector<float> &h, float **dev) {
    void *p = nullptr;
    RMR_HIP(hipMalloc(&p, h.size() * sizeof(float) + 16));
    m->dev_allocs.push_back(p);
    RMR_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    *dev = reinterpret_cast<float *>(p);
    return 0;
}

int pack_conv(rmr_model *m, const Folded &f, int kid, ConvLayer *out) {
    const ConvSpec &s = f.s;
    if (s.ic % 16 || s.oc % 16) RMR_FAIL(RMR_ERR_INVALID, "conv %dx%d not MFMA-tileable", s.ic, s.oc);
    const int G = s.ic / 16, S = s.kw * s.ic / 4, W = s.oc / 16;
    if (s.oc > 64 || s.ic > 128) {  // a layer of a network with more than 64 channels: the streamed kernel's order (k_stream.hip)
        std::vector<float> ap((size_t)W * s.kw * G * 64 * 4);
        for (int w = 0; w < W; ++w)
            for (int tap = 0; tap < s.kw; ++tap)
                for (int g = 0; g < G; ++g)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j) {
                            const int q = lane >> 4, mm = lane & 15;
                            const int oc = 16 * w + mm, ic = 16 * g + 4 * q + j;
                            ap[((((size_t)w * s.kw + tap) * G + g) * 64 + lane) * 4 + j] = f.w[((size_t)oc * s.ic + ic) * s.kw + tap];
                        }
        out->ic = s.ic; out->oc = s.oc; out->kw = s.kw; out->stride = s.stride; out->kid = kid;
        RMR_TRY(upload(m, ap, &out->apack4));
        RMR_TRY(upload(m, f.b, &out->bias));
        return 0;
    }
    std::vector<float> ap((size_t)W * S * 64);
    for (int w = 0; w < W; ++w)
        for (int tap = 0; tap < s.kw; ++tap)
            for (int g = 0; g < G; ++g)
                for (int j = 0; j < 4; ++j) {
                    const int st = (tap * G + g) * 4 + j;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int q = lane >> 4, mm = lane & 15;
                        const int oc = 16 * w + mm, ic = 16 * g + 4 * q + j;
                        ap[((size_t)w * S + st) * 64 + lane] = f.w[((size_t)oc * s.ic + ic) * s.kw + tap];
                    }
                }
    out->ic = s.ic; out->oc = s.oc; out->kw = s.kw; out->stride = s.stride; out->kid = kid;
    RMR_TRY(upload(m, ap, &out->apack));
    RMR_TRY(upload(m, f.b, &out->bias));
    if (s.kw == 5 && s.stride == 1 && s.oc == 64 && (s.ic == 128 || s.ic == 64)) {
        // Winograd F(4, 5) filter transform U = G W at the Toom-Cook points 0, 1, -1, 2, -2, 1/2, -1/2, inf (k_wino.hip has BT and AT;
        // oracle/winograd.py derives all three), in float64 from the folded fp32 weights, ONE rounding to fp32.  Rows in the kernel's
        // x order - the points 1, -1, 2, -2 (wave half 0), then 0, 1/2, -1/2, inf (half 1); fragment order
        // [oc/16][(x * G + g) * 4 + j][64 lanes].
        static const double GM[8][5] = {{1.0 / 18, 1.0 / 18, 1.0 / 18, 1.0 / 18, 1.0 / 18},
                                        {1.0 / 18, -1.0 / 18, 1.0 / 18, -1.0 / 18, 1.0 / 18},
                                        {1.0 / 360, 1.0 / 180, 1.0 / 90, 1.0 / 45, 2.0 / 45},
                                        {1.0 / 360, -1.0 / 180, 1.0 / 90, -1.0 / 45, 2.0 / 45},
                                        {1.0 / 4, 0, 0, 0, 0},
                                        {16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45, 1.0 / 45},
                                        {16.0 / 45, -8.0 / 45, 4.0 / 45, -2.0 / 45, 1.0 / 45},
                                        {0, 0, 0, 0, 1.0 / 4}};
        const int SW = 8 * s.ic / 4;
        std::vector<float> wp((size_t)W * SW * 64);
        for (int w = 0; w < W; ++w)
            for (int x = 0; x < 8; ++x)
                for (int g = 0; g < G; ++g)
                    for (int j = 0; j < 4; ++j)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int q = lane >> 4, mm = lane & 15;
                            const int oc = 16 * w + mm, ic = 16 * g + 4 * q + j;
                            double u = 0.0;
                            for (int tap = 0; tap < 5; ++tap) u += GM[x][tap] * (double)f.w[((size_t)oc * s.ic + ic) * 5 + tap];
                            wp[((size_t)w * SW + (x * G + g) * 4 + j) * 64 + lane] = (float)u;
                        }
        RMR_TRY(upload(m, wp, &out->wpack));
    }
    if (s.stride == 3 && s.ic == 16 && s.oc == 64 && s.kw == 9) {
        // sig_conv3 (models/ConvLSTM_w_ref.py:22-23,43; models/Conv_w_ref.py:22-23,47) for k_conv_front.hip's sig3_front_wino_kernel: the three
        // phase filters w_p[m] = w[3 m + p] in F(4, 3) form, natural point order (0, 1, -1, 2, -2, inf); [oc/16][(x * 3 + p) * 4 + j][64 lanes]
        static const double G3n[6][3] = {{1.0 / 4, 0, 0}, {1.0 / 6, 1.0 / 6, 1.0 / 6}, {1.0 / 6, -1.0 / 6, 1.0 / 6},
                                         {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1.0}};
        const int SW = 6 * 12;
        std::vector<float> wp((size_t)W * SW * 64);
        for (int w = 0; w < W; ++w)
            for (int x = 0; x < 6; ++x)
                for (int ph = 0; ph < 3; ++ph)
                    for (int j = 0; j < 4; ++j)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int q = lane >> 4, mm = lane & 15;
                            const int oc = 16 * w + mm, ic = 4 * q + j;
                            double u = 0.0;
                            for (int t = 0; t < 3; ++t) u += G3n[x][t] * (double)f.w[((size_t)oc * s.ic + ic) * s.kw + 3 * t + ph];
                            wp[((size_t)w * SW + (x * 3 + ph) * 4 + j) * 64 + lane] = (float)u;
                        }
        RMR_TRY(upload(m, wp, &out->wpack));
    }
    if (s.stride == 3 && s.ic == 16 && s.oc == 64 && s.kw == 13) {
        // seq_conv2 (models/ConvLSTM_w_ref.py:30-31,46) for k_conv_front.hip's seq2_front_wino_kernel: phase filters w_p[m] = w[3 m + p] of 5, 4 and
        // 4 taps, all as F(4, 5) (a zero fifth tap where 3 m + p > 12), natural point order (0, 1, -1, 2, -2, 1/2, -1/2, inf);
        // [oc/16][(x * 3 + p) * 4 + j][64 lanes]
        static const double G5n[8][5] = {{1.0 / 4, 0, 0, 0, 0},
                                         {1.0 / 18, 1.0 / 18, 1.0 / 18, 1.0 / 18, 1.0 / 18},
                                         {1.0 / 18, -1.0 / 18, 1.0 / 18, -1.0 / 18, 1.0 / 18},
                                         {1.0 / 360, 1.0 / 180, 1.0 / 90, 1.0 / 45, 2.0 / 45},
                                         {1.0 / 360, -1.0 / 180, 1.0 / 90, -1.0 / 45, 2.0 / 45},
                                         {16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45, 1.0 / 45},
                                         {16.0 / 45, -8.0 / 45, 4.0 / 45, -2.0 / 45, 1.0 / 45},
                                         {0, 0, 0, 0, 1.0 / 4}};
        const int SW = 8 * 12;
        std::vector<float> wp((size_t)W * SW * 64);
        for (int w = 0; w < W; ++w)
            for (int x = 0; x < 8; ++x)
                for (int ph = 0; ph < 3; ++ph)
                    for (int j = 0; j < 4; ++j)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int q = lane >> 4, mm = lane & 15;
                            const int oc = 16 * w + mm, ic = 4 * q + j;
                            double u = 0.0;
                            for (int t = 0; t < 5; ++t)
                                if (3 * t + ph < s.kw) u += G5n[x][t] * (double)f.w[((size_t)oc * s.ic + ic) * s.kw + 3 * t + ph];
                            wp[((size_t)w * SW + (x * 3 + ph) * 4 + j) * 64 + lane] = (float)u;
                        }
        RMR_TRY(upload(m, wp, &out->wpack));
    }
    if (s.stride == 3 && s.ic == 32 && s.oc == 64 && s.kw == 9) {
        // Conv_w_ref's seq_conv3 (models/Conv_w_ref.py:31-32,51): stride 3 as three phase filters w_p[m] = w[3 m + p] of three taps
        // each, F(4, 3) at 0, +-1, +-2, inf (k_wino.hip wino_s3_kernel; oracle/winograd.py); natural point order; fragment order
        // [oc/16][((x * 3 + p) * G + g) * 4 + j][64 lanes], K of a point's GEMM = (phase, channel)
        static const double G3[6][3] = {{1.0 / 4, 0, 0}, {1.0 / 6, 1.0 / 6, 1.0 / 6}, {1.0 / 6, -1.0 / 6, 1.0 / 6},
                                        {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1.0}};
        static const int XO[6] = {1, 2, 0, 3, 4, 5};  // the kernel's x order: wave half 0 (+1, -1, 0), half 1 (+2, -2, inf)
        const int NX = 6, SW = NX * 3 * G * 4;
        std::vector<float> wp((size_t)W * SW * 64);
        for (int w = 0; w < W; ++w)
            for (int x = 0; x < NX; ++x)
                for (int ph = 0; ph < 3; ++ph)
                    for (int g = 0; g < G; ++g)
                        for (int j = 0; j < 4; ++j)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int q = lane >> 4, mm = lane & 15;
                                const int oc = 16 * w + mm, ic = 16 * g + 4 * q + j;
                                double u = 0.0;
                                for (int t = 0; t < 3; ++t) u += G3[XO[x]][t] * (double)f.w[((size_t)oc * s.ic + ic) * s.kw + 3 * t + ph];
                                wp[((size_t)w * SW + ((x * 3 + ph) * G + g) * 4 + j) * 64 + lane] = (float)u;
                            }
        RMR_TRY(upload(m, wp, &out->wpack));
    }
    return 0;
}

// gate pre-scale used by lstm_step (k_lstm.hip): sigmoid(x) = 1/(1+2^(-x log2 e)) for i,f,o;
// tanh(x) = 1 - 2/(1+2^(2x log2 e)) for g.  torch gate order i,f,g,o.
static inline double lstm1_gate_scale(int gate) {
    const double log2e = 1.4426950408889634;
    return gate == 2 ? 2.0 * log2e : -log2e;
}

// conv weights -> split-bf16 A fragments [oc/16][steps][np][64 lanes][4 dwords]
// (k-slot mapping documented at the top of k_conv_bf16s.hip)
int pack_conv_split(rmr_model *m, const Folded &f, int np, ConvLayer *out) {
    out->split_f16 = m->split_f16;
    const ConvSpec &s = f.s;
    const bool pair = (s.ic == 16);
    if (!pair && s.ic % 32) RMR_FAIL(RMR_ERR_INVALID, "split conv needs ic 16 or a multiple of 32 (got %d)", s.ic);
    const int KS = pair ? 1 : s.ic / 32;
    const int steps = pair ? (s.kw + 1) / 2 : s.kw * KS;
    const int W = s.oc / 16;
    std::vector<uint32_t> o((size_t)W * steps * np * 64 * 4);
    for (int w = 0; w < W; ++w)
        for (int st = 0; st < steps; ++st)
            for (int lane = 0; lane < 64; ++lane) {
                const int q = lane >> 4, oc = 16 * w + (lane & 15);
                uint32_t parts[8][3];
                for (int j = 0; j < 8; ++j) {
                    int tap, ch;
                    if (pair) { tap = 2 * st + (q >> 1); ch = 8 * (q & 1) + j; }
                    else { tap = st / KS; ch = 32 * (st % KS) + 8 * q + j; }
                    const float v = tap < s.kw ? f.w[((size_t)oc * s.ic + ch) * s.kw + tap] : 0.0f;
                    split_parts_host(v, np, parts[j], m->split_f16);
                }
                for (int p = 0; p < np; ++p)
                    for (int i = 0; i < 4; ++i)
                        o[((((size_t)w * steps + st) * np + p) * 64 + lane) * 4 + i] = (parts[2 * i][p] >> 16) | parts[2 * i + 1][p];
            }
    std::vector<float> fl(o.size());
    memcpy(fl.data(), o.data(), o.size() * 4);
    return upload(m, fl, &out->spack);
}

// conv weights -> bf16 A fragments of the fused front kernel: [oc/16][ksteps][64 lanes][4 dwords]; lane (q, m) of
// k-step s holds k = 32 s + 8 q + j, k = tap * C + channel (C = row width of the operand in LDS, k_fused.hip);
// taps >= kw and channels >= ic are zero
int pack_flat_a(rmr_model *m, const Folded &f, int C, int ksteps, double scale, float **dev, bool f16 = false) {
    const ConvSpec &s = f.s;
    const int W = s.oc / 16;
    std::vector<uint32_t> o((size_t)W * ksteps * 64 * 4);
    for (int w = 0; w < W; ++w)
        for (int st = 0; st < ksteps; ++st)
            for (int lane = 0; lane < 64; ++lane) {
                const int q = lane >> 4, oc = 16 * w + (lane & 15);
                uint32_t b[8];
                for (int j = 0; j < 8; ++j) {
                    const int k = 32 * st + 8 * q + j, tap = k / C, ch = k % C;
                    const float v = (tap < s.kw && ch < s.ic) ? (float)(scale * (double)f.w[((size_t)oc * s.ic + ch) * s.kw + tap]) : 0.0f;
                    b[j] = to_op16(v, f16);
                }
                for (int i = 0; i < 4; ++i) o[(((size_t)w * ksteps + st) * 64 + lane) * 4 + i] = (b[2 * i] >> 16) | b[2 * i + 1];
            }
    std::vector<float> fl(o.size());
    memcpy(fl.data(), o.data(), o.size() * 4);
    return upload(m, fl, dev);
}

// LSTM weights for k_lstm_x16.hip (H = 64): 16-row MFMA tiles with UNIT-MAJOR rows — row r of tile (wave wv, t) is
// (unit 8 wv + 2 (r >> 2) + t, gate r & 3) — as bf16 A fragments [8][2][2 k-steps][64 lanes][4 dwords]; gate rows
// pre-scaled (lstm1_gate_scale); `skip_f` zeroes the f rows (lstm2: c0 = 0)
std::vector<float> pack_lstm_x16(const float *w, bool skip_f, bool f16 = false) {
    const int H = 64;
    std::vector<uint32_t> o((size_t)8 * 2 * 2 * 64 * 4);
    for (int wv = 0; wv < 8; ++wv)
        for (int t = 0; t < 2; ++t)
            for (int ks = 0; ks < 2; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int ql = lane >> 4, mm = lane & 15, gate = mm & 3, unit = 8 * wv + 2 * (mm >> 2) + t;
                    uint32_t b[8];
                    for (int j = 0; j < 8; ++j) {
                        const int k = 32 * ks + 8 * ql + j;
                        const double v = (skip_f && gate == 1) ? 0.0 : (double)w[(size_t)(gate * H + unit) * H + k] * lstm1_gate_scale(gate);
                        b[j] = to_op16((float)v, f16);
                    }
                    for (int i = 0; i < 4; ++i)
                        o[((((size_t)wv * 2 + t) * 2 + ks) * 64 + lane) * 4 + i] = (b[2 * i] >> 16) | b[2 * i + 1];
                }
    std::vector<float> f(o.size());
    memcpy(f.data(), o.data(), o.size() * 4);
    return f;
}
// the same fragments as NP split parts (k_lstm_x16s.hip): [8][2][2 k-steps][np][64 lanes][4 dwords]
std::vector<float> pack_lstm_x16_split(const float *w, bool skip_f, int np, bool f16) {
    const int H = 64;
    std::vector<uint32_t> o((size_t)8 * 2 * 2 * np * 64 * 4);
    for (int wv = 0; wv < 8; ++wv)
        for (int t = 0; t < 2; ++t)
            for (int ks = 0; ks < 2; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int ql = lane >> 4, mm = lane & 15, gate = mm & 3, unit = 8 * wv + 2 * (mm >> 2) + t;
                    uint32_t parts[8][3];
                    for (int j = 0; j < 8; ++j) {
                        const int k = 32 * ks + 8 * ql + j;
                        const double v = (skip_f && gate == 1) ? 0.0 : (double)w[(size_t)(gate * H + unit) * H + k] * lstm1_gate_scale(gate);
                        split_parts_host((float)v, np, parts[j], f16);
                    }
                    for (int p = 0; p < np; ++p)
                        for (int i = 0; i < 4; ++i)
                            o[(((((size_t)wv * 2 + t) * 2 + ks) * np + p) * 64 + lane) * 4 + i] = (parts[2 * i][p] >> 16) | parts[2 * i + 1][p];
                }
    std::vector<float> f(o.size());
    memcpy(f.data(), o.data(), o.size() * 4);
    return f;
}
// matching biases [8][2][4 q][4 gates]: (b_ih + b_hh) of unit 8 wv + 2 q + t, pre-scaled
std::vector<float> pack_bias_x16(const float *bih, const float *bhh, bool skip_f) {
    const int H = 64;
    std::vector<float> o((size_t)8 * 2 * 4 * 4);
    for (int wv = 0; wv < 8; ++wv)
        for (int t = 0; t < 2; ++t)
            for (int q = 0; q < 4; ++q)
                for (int gate = 0; gate < 4; ++gate) {
                    const int unit = 8 * wv + 2 * q + t;
                    o[(((size_t)wv * 2 + t) * 4 + q) * 4 + gate] =
                        (skip_f && gate == 1) ? 0.0f : (float)(((double)bih[gate * H + unit] + (double)bhh[gate * H + unit]) * lstm1_gate_scale(gate));
                }
    return o;
}

// [4H][H] row-major -> [H/16 waves][H/16 k groups][ngates][64 lanes][4] (k_stream.hip: one 16-byte fragment per gate and group)
std::vector<float> pack_lstm_stream(const float *w, int H, const int *gates, int ngates, bool prescale) {
    const int G = H / 16, W = H / 16;
    std::vector<float> ap((size_t)W * G * ngates * 64 * 4);
    for (int wv = 0; wv < W; ++wv)
        for (int g = 0; g < G; ++g)
            for (int gi = 0; gi < ngates; ++gi)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int q = lane >> 4, mm = lane & 15;
                        const int row = gates[gi] * H + 16 * wv + mm, k = 16 * g + 4 * q + j;
                        const double sc = prescale ? lstm1_gate_scale(gates[gi]) : 1.0;
                        ap[((((size_t)wv * G + g) * ngates + gi) * 64 + lane) * 4 + j] = (float)((double)w[(size_t)row * H + k] * sc);
                    }
    return ap;
}

// LSTM weights for k_stream16.hip (H a multiple of 32 above 64): [H/16 waves][4 tiles][H/32 k-steps][64 lanes][4 dwords]; row m of tile t of
// wave wv = (unit 16 wv + 4 (m >> 2) + t, gate m & 3); gate rows pre-scaled (lstm1_gate_scale); `skip_f` zeroes the f rows (lstm2: c0 = 0)
std::vector<float> pack_lstm_s16(const float *w, int H, bool skip_f, bool f16) {
    const int W = H / 16, KSH = H / 32;
    std::vector<uint32_t> o((size_t)W * 4 * KSH * 64 * 4);
    for (int wv = 0; wv < W; ++wv)
        for (int t = 0; t < 4; ++t)
            for (int ks = 0; ks < KSH; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int ql = lane >> 4, mm = lane & 15, gate = mm & 3, unit = 16 * wv + 4 * (mm >> 2) + t;
                    uint32_t b[8];
                    for (int j = 0; j < 8; ++j) {
                        const int k = 32 * ks + 8 * ql + j;
                        const double v = (skip_f && gate == 1) ? 0.0 : (double)w[(size_t)(gate * H + unit) * H + k] * lstm1_gate_scale(gate);
                        b[j] = to_op16((float)v, f16);
                    }
                    for (int i = 0; i < 4; ++i) o[((((size_t)wv * 4 + t) * KSH + ks) * 64 + lane) * 4 + i] = (b[2 * i] >> 16) | b[2 * i + 1];
                }
    std::vector<float> f(o.size());
    memcpy(f.data(), o.data(), o.size() * 4);
    return f;
}
// matching biases [H/16][4 tiles][4 q][4 gates]: (b_ih + b_hh) of unit 16 wv + 4 q + t, pre-scaled
std::vector<float> pack_bias_s16(const float *bih, const float *bhh, int H, bool skip_f) {
    const int W = H / 16;
    std::vector<float> o((size_t)W * 4 * 4 * 4);
    for (int wv = 0; wv < W; ++wv)
        for (int t = 0; t < 4; ++t)
            for (int q = 0; q < 4; ++q)
                for (int gate = 0; gate < 4; ++gate) {
                    const int unit = 16 * wv + 4 * q + t;
                    o[(((size_t)wv * 4 + t) * 4 + q) * 4 + gate] =
                        (skip_f && gate == 1) ? 0.0f : (float)(((double)bih[gate * H + unit] + (double)bhh[gate * H + unit]) * lstm1_gate_scale(gate));
                }
    return o;
}

// [4H][H] row-major (H = 64) -> [4 waves][64 k in lstm_head_kernel's order: position (g * 4 + j) * 4 + q = k 16 g + 4 q + j][64 lanes],
// lane l = gate gates[l & 3] (a negative entry: zeros) of unit 16 w + (l >> 2)   (lstm_small_kernel, k_lstm.hip)
std::vector<float> pack_lstm_small(const float *w, const int *gates, bool prescale) {
    const int H = 64;
    std::vector<float> ap((size_t)4 * H * 64);
    for (int wv = 0; wv < 4; ++wv)
        for (int g = 0; g < 4; ++g)
            for (int j = 0; j < 4; ++j)
                for (int q = 0; q < 4; ++q)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int gate = gates[lane & 3], unit = 16 * wv + (lane >> 2), k = 16 * g + 4 * q + j;
                        const double sc = prescale && gate >= 0 ? lstm1_gate_scale(gate) : 1.0;
                        ap[((size_t)wv * H + (g * 4 + j) * 4 + q) * 64 + lane] =
                            gate < 0 ? 0.0f : (float)((double)w[(size_t)(gate * H + unit) * H + k] * sc);
                    }
    return ap;
}

// [4H][H] row-major -> [H/16 waves][ngates][H/4][64]
std::vector<float> pack_lstm(const float *w, int H, const int *gates, int ngates, bool prescale = false) {
    const int KS = H / 4, G = H / 16, W = H / 16;
    std::vector<float> ap((size_t)W * ngates * KS * 64);
    for (int wv = 0; wv < W; ++wv)
        for (int gi = 0; gi < ngates; ++gi)
            for (int g = 0; g < G; ++g)
                for (int j = 0; j < 4; ++j)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int q = lane >> 4, mm = lane & 15;
                        const int row = gates[gi] * H + 16 * wv + mm, k = 16 * g + 4 * q + j;
                        const double sc = prescale ? lstm1_gate_scale(gates[gi]) : 1.0;
                        ap[(((size_t)wv * ngates + gi) * KS + g * 4 + j) * 64 + lane] = (float)((double)w[(size_t)row * H + k] * sc);
                    }
    return ap;
}

}  // namespace

extern "C" {

size_t rmr_model_weight_count(const rmr_model_desc *d) {
    if (!d || !desc_ok(*d)) return 0;
    size_t n = 0;
    for (auto &s : conv_specs(*d)) n += conv_count(s);
    const size_t H = d->size;
    if (d->arch == RMR_ARCH_CONV_LSTM) {
        n += 2 * (2 * 4 * H * H + 2 * 4 * H);
        n += (size_t)d->num_out * H + d->num_out;
    } else {
        n += (size_t)d->num_out * H * 3 + d->num_out;
    }
    return n;
}

void rmr_model_destroy(rmr_model *m) {
    if (!m) return;
    if (m->eng) {
        (void)hipSetDevice(m->eng->device);
        (void)hipStreamSynchronize(m->eng->stream);
    }
    for (void *p : m->dev_allocs) (void)hipFree(p);
    delete m;
}

int rmr_model_padded_size(const rmr_model_desc *d) { return (d && desc_ok(*d)) ? padded_size(d->size, d->dtype) : 0; }

int rmr_model_pad_weights(const rmr_model_desc *desc, const float *weights, size_t n_floats, rmr_model_desc *padded_desc,
                          float *out, size_t out_cap, size_t *out_n) {
    if (!desc || !weights || !padded_desc || !out_n) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (!desc_ok(*desc)) RMR_FAIL(RMR_ERR_INVALID, "unsupported model description");
    if (rmr_model_weight_count(desc) != n_floats)
        RMR_FAIL(RMR_ERR_INVALID, "weight blob has %zu floats, expected %zu", n_floats, rmr_model_weight_count(desc));
    *padded_desc = *desc;
    padded_desc->size = padded_size(desc->size, desc->dtype);
    *out_n = rmr_model_weight_count(padded_desc);
    if (!out) return 0;  // size query
    if (out_cap < *out_n) RMR_FAIL(RMR_ERR_INVALID, "output holds %zu floats, %zu needed", out_cap, *out_n);
    if (padded_desc->size == desc->size) {
        memcpy(out, weights, n_floats * sizeof(float));
        return 0;
    }
    const std::vector<float> o = pad_model_blob(*desc, weights, padded_desc->size);
    if (o.size() != *out_n) RMR_FAIL(RMR_ERR_INVALID, "internal: padded blob has %zu floats, expected %zu", o.size(), *out_n);
    memcpy(out, o.data(), o.size() * sizeof(float));
    return 0;
}

static int model_create_at_kernel_size(rmr_engine *e, const rmr_model_desc *desc, const float *weights, size_t n_floats, rmr_model **out);

int rmr_model_create(rmr_engine *e, const rmr_model_desc *desc, const float *weights,
                     size_t n_floats, rmr_model **out) {
    if (!e || !desc || !weights || !out) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (!desc_ok(*desc))
        RMR_FAIL(RMR_ERR_INVALID,
                 "unsupported model: arch=%d size=%d kmer_len=%d num_out=%d dtype=%d "
                 "(size 1..%d; num_out <= 16; the 16-bit dtypes need conv_lstm; up to 64 channels f16 takes 33..64 channels and a k-mer "
                 "length of 9 or 6; above 64 channels the dtypes are fp32, bf16 and f16)",
                 desc->arch, desc->size, desc->kmer_len, desc->num_out, desc->dtype, kMaxPaddedSize);
    const size_t want = rmr_model_weight_count(desc);
    if (want != n_floats) RMR_FAIL(RMR_ERR_INVALID, "weight blob has %zu floats, expected %zu", n_floats, want);
    const int sp = padded_size(desc->size, desc->dtype);
    if (sp == desc->size) return model_create_at_kernel_size(e, desc, weights, n_floats, out);
    rmr_model_desc pd = *desc;
    pd.size = sp;
    const std::vector<float> blob = pad_model_blob(*desc, weights, sp);
    RMR_TRY(model_create_at_kernel_size(e, &pd, blob.data(), blob.size(), out));
    (*out)->true_size = desc->size;
    return 0;
}

// `desc->size` is a size the kernels run at (padded_size is the identity on it)
static int model_create_at_kernel_size(rmr_engine *e, const rmr_model_desc *desc, const float *weights, size_t n_floats, rmr_model **out) {
    const size_t want = rmr_model_weight_count(desc);
    if (want != n_floats) RMR_FAIL(RMR_ERR_INVALID, "internal: padded weight blob has %zu floats, expected %zu", n_floats, want);
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    std::unique_ptr<rmr_model, void (*)(rmr_model *)> m(new rmr_model(), rmr_model_destroy);
    m->eng = e;
    m->desc = *desc;
    m->true_size = desc->size;
    m->nparts = desc->dtype == 4 ? 1 : (desc->dtype == 5 ? 2 : desc->dtype);  // 0 fp32 MFMA; 1 bf16; 2 bf16x3 (2-part split); 3 bf16x6 (3-part split)
    m->split_f16 = desc->dtype == 5;  // f16x3: the two parts are IEEE half
    m->f16 = desc->dtype == 4;                       // 4: one-part operands as IEEE half (fused kernels)
    const int sz = desc->size, K = desc->kmer_len, L = desc->chunk_len;

    const float *p = weights;
    std::vector<Folded> convs;
    for (auto &s : conv_specs(*desc)) convs.push_back(fold(s, p));

    // ---- geometry ----
    const int kw1 = convs[0].s.kw;
    m->L = L;
    m->P1 = L - kw1 + 1;
    m->P2 = m->P1 - kw1 + 1;
    if (m->P2 < 9) RMR_FAIL(RMR_ERR_INVALID, "chunk_len %d too short for this architecture", L);
    m->P3 = (m->P2 - 9) / 3 + 1;
    if (desc->arch == RMR_ARCH_CONV_LSTM) {
        if ((m->P1 - 13) / 3 + 1 != m->P3) RMR_FAIL(RMR_ERR_INVALID, "branch lengths differ");
        m->T = m->P3 - 4;
        if (m->T < 1) RMR_FAIL(RMR_ERR_INVALID, "chunk_len %d too short", L);
    } else {
        m->PQ2 = m->P1 - 10;
        if (m->PQ2 < 9 || (m->PQ2 - 9) / 3 + 1 != m->P3) RMR_FAIL(RMR_ERR_INVALID, "branch lengths differ");
        m->T = m->P3 - 4;
        m->T2 = m->T - 4;
        m->T3 = (m->T2 - 3) / 2 + 1;
        m->T4 = (m->T3 - 3) / 2 + 1;
        if (m->T2 < 3 || m->T3 < 3 || m->T4 != 3)
            RMR_FAIL(RMR_ERR_INVALID, "Conv_w_ref needs 3 final positions (fc in = size*3), chunk_len %d gives %d", L, m->T4);
    }

    // ---- front weights ----
    {
        const Folded &s1 = convs[0], &s2 = convs[1], &q1 = convs[3];
        std::vector<float> w1((size_t)kw1 * 4), w2((size_t)kw1 * 64), wt((size_t)kw1 * K * 64);
        for (int t = 0; t < kw1; ++t)
            for (int o = 0; o < 4; ++o) w1[t * 4 + o] = s1.w[(size_t)o * kw1 + t];
        for (int t = 0; t < kw1; ++t)
            for (int ic = 0; ic < 4; ++ic)
                for (int o = 0; o < 16; ++o) w2[(t * 4 + ic) * 16 + o] = s2.w[((size_t)o * 4 + ic) * kw1 + t];
        const int ec = 4 * K;
        for (int t = 0; t < kw1; ++t)
            for (int c = 0; c < ec; ++c)
                for (int o = 0; o < 16; ++o) wt[((size_t)t * ec + c) * 16 + o] = q1.w[((size_t)o * ec + c) * kw1 + t];
        m->front.kw1 = kw1;
        RMR_TRY(upload(m.get(), w1, &m->front.w_sig1));
        RMR_TRY(upload(m.get(), s1.b, &m->front.b_sig1));
        RMR_TRY(upload(m.get(), w2, &m->front.w_sig2));
        RMR_TRY(upload(m.get(), s2.b, &m->front.b_sig2));
        RMR_TRY(upload(m.get(), wt, &m->front.wt_seq1));
        std::vector<float> wt5((size_t)kw1 * K * 80, 0.0f);
        for (int t = 0; t < kw1; ++t)
            for (int kp = 0; kp < K; ++kp)
                for (int b = 0; b < 4; ++b)
                    for (int o = 0; o < 16; ++o)
                        wt5[(((size_t)t * K + kp) * 5 + b) * 16 + o] = q1.w[((size_t)o * ec + 4 * kp + b) * kw1 + t];
        RMR_TRY(upload(m.get(), wt5, &m->front.wt5_seq1));
        RMR_TRY(upload(m.get(), q1.b, &m->front.b_seq1));
    }
    RMR_TRY(pack_conv(m.get(), convs[2], K_CONV_SIG3, &m->sig3));
    RMR_TRY(pack_conv(m.get(), convs[4], K_CONV_SEQ2, &m->seq2));
    if (desc->arch == RMR_ARCH_CONV_LSTM) {
        RMR_TRY(pack_conv(m.get(), convs[5], K_CONV_MERGE1, &m->merge1));
        if (m->nparts > 0) {
            RMR_TRY(pack_conv_split(m.get(), convs[2], m->nparts, &m->sig3));
            RMR_TRY(pack_conv_split(m.get(), convs[4], m->nparts, &m->seq2));
            RMR_TRY(pack_conv_split(m.get(), convs[5], m->nparts, &m->merge1));
        }
        if (m->nparts == 1 && sz > 64) {  // k_stream16.hip: 16-bit A fragments of the three size-wide layers, k = tap * ic + channel
            RMR_TRY(pack_flat_a(m.get(), convs[2], convs[2].s.ic, (convs[2].s.kw * convs[2].s.ic + 31) / 32, 1.0, &m->sig3.apack16, m->f16));
            RMR_TRY(pack_flat_a(m.get(), convs[4], convs[4].s.ic, (convs[4].s.kw * convs[4].s.ic + 31) / 32, 1.0, &m->seq2.apack16, m->f16));
            RMR_TRY(pack_flat_a(m.get(), convs[5], convs[5].s.ic, (convs[5].s.kw * convs[5].s.ic + 31) / 32, 1.0, &m->merge1.apack16, m->f16));
        }
        if (m->nparts == 1 && sz == 64 && (K == 9 || K == 6) && kw1 == 5) {  // operands of the fused front kernel
            const int cg = (4 * K + 7) / 8;
            const double log2e = 1.4426950408889634;
            RMR_TRY(pack_flat_a(m.get(), convs[1], 4, 1, 1.0, &m->fused.a_sig2, m->f16));
            RMR_TRY(pack_flat_a(m.get(), convs[3], 8 * cg, (5 * cg * 8 + 31) / 32, log2e, &m->fused.a_seq1, m->f16));
            RMR_TRY(pack_flat_a(m.get(), convs[2], 16, 5, 1.0, &m->fused.a_sig3, m->f16));
            RMR_TRY(pack_flat_a(m.get(), convs[4], 16, 7, 1.0, &m->fused.a_seq2, m->f16));
            RMR_TRY(pack_flat_a(m.get(), convs[5], 2 * sz, 20, 1.0, &m->fused.a_merge1, m->f16));
            auto scaled = [&](const std::vector<float> &v) {
                std::vector<float> o(v.size());
                for (size_t i = 0; i < v.size(); ++i) o[i] = (float)((double)v[i] * log2e);
                return o;
            };
            std::vector<float> w1s((size_t)kw1 * 4);
            for (int t = 0; t < kw1; ++t)
                for (int o = 0; o < 4; ++o) w1s[t * 4 + o] = (float)((double)convs[0].w[(size_t)o * kw1 + t] * log2e);
            RMR_TRY(upload(m.get(), w1s, &m->fused.w_sig1));
            RMR_TRY(upload(m.get(), scaled(convs[0].b), &m->fused.b_sig1));
            RMR_TRY(upload(m.get(), scaled(convs[1].b), &m->fused.b_sig2));
            RMR_TRY(upload(m.get(), scaled(convs[3].b), &m->fused.b_seq1));
            RMR_TRY(upload(m.get(), scaled(convs[2].b), &m->fused.b_sig3));
            RMR_TRY(upload(m.get(), scaled(convs[4].b), &m->fused.b_seq2));
            RMR_TRY(upload(m.get(), scaled(convs[5].b), &m->fused.b_merge1));
        }
        const int H = sz;
        const float *wih1 = p; p += (size_t)4 * H * H;
        const float *whh1 = p; p += (size_t)4 * H * H;
        const float *bih1 = p; p += 4 * H;
        const float *bhh1 = p; p += 4 * H;
        const float *wih2 = p; p += (size_t)4 * H * H;
        p += (size_t)4 * H * H;  // lstm2.weight_hh_l0 multiplies h0 == 0: never reaches the output
        const float *bih2 = p; p += 4 * H;
        const float *bhh2 = p; p += 4 * H;
        const float *wfc = p; p += (size_t)desc->num_out * H;
        const float *bfc = p; p += desc->num_out;
        const int g4[4] = {0, 1, 2, 3}, g3[3] = {0, 2, 3};
        if (H > 64) {  // k_stream.hip
            RMR_TRY(upload(m.get(), pack_lstm_stream(wih1, H, g4, 4, true), &m->lstm.t_ih1));
            RMR_TRY(upload(m.get(), pack_lstm_stream(whh1, H, g4, 4, true), &m->lstm.t_hh1));
            RMR_TRY(upload(m.get(), pack_lstm_stream(wih2, H, g3, 3, false), &m->lstm.t_ih2));
        } else {
            RMR_TRY(upload(m.get(), pack_lstm(wih1, H, g4, 4, true), &m->lstm.a_ih1));
            RMR_TRY(upload(m.get(), pack_lstm(whh1, H, g4, 4, true), &m->lstm.a_hh1));
            RMR_TRY(upload(m.get(), pack_lstm(wih2, H, g3, 3), &m->lstm.a_ih2));
            if (H == 64 && m->nparts == 0) {  // the four-chunk kernel of small batches (one read per call)
                const int g3z[4] = {0, 2, 3, -1};
                RMR_TRY(upload(m.get(), pack_lstm_small(wih1, g4, true), &m->lstm.q_ih1));
                RMR_TRY(upload(m.get(), pack_lstm_small(whh1, g4, true), &m->lstm.q_hh1));
                RMR_TRY(upload(m.get(), pack_lstm_small(wih2, g3z, false), &m->lstm.q_ih2));
            }
        }
        if (m->nparts > 0) {
            std::vector<float> si((size_t)4 * H * H), sh((size_t)4 * H * H);
            for (int r = 0; r < 4 * H; ++r)
                for (int k = 0; k < H; ++k) {
                    si[(size_t)r * H + k] = (float)((double)wih1[(size_t)r * H + k] * lstm1_gate_scale(r / H));
                    sh[(size_t)r * H + k] = (float)((double)whh1[(size_t)r * H + k] * lstm1_gate_scale(r / H));
                }
            const int rb[4] = {0, H, 2 * H, 3 * H};
            RMR_TRY(upload(m.get(), pack_split_a(si, H, H / 16, rb, 4, m->nparts, m->split_f16), &m->lstm.s_ih1));
            RMR_TRY(upload(m.get(), pack_split_a(sh, H, H / 16, rb, 4, m->nparts, m->split_f16), &m->lstm.s_hh1));
        }
        if (m->nparts >= 2 && H == 64) {  // split operands in the x16 layout (k_lstm_x16s.hip)
            RMR_TRY(upload(m.get(), pack_lstm_x16_split(wih1, false, m->nparts, m->split_f16), &m->lstm.xs_ih));
            RMR_TRY(upload(m.get(), pack_lstm_x16_split(whh1, false, m->nparts, m->split_f16), &m->lstm.xs_hh));
            RMR_TRY(upload(m.get(), pack_lstm_x16_split(wih2, true, m->nparts, m->split_f16), &m->lstm.xs_ih2));
            RMR_TRY(upload(m.get(), pack_bias_x16(bih1, bhh1, false), &m->lstm.x_b1));
            RMR_TRY(upload(m.get(), pack_bias_x16(bih2, bhh2, true), &m->lstm.x_b2));
        }
        if (m->nparts == 1 && H > 64) {  // k_stream16.hip
            RMR_TRY(upload(m.get(), pack_lstm_s16(wih1, H, false, m->f16), &m->lstm.s16_ih));
            RMR_TRY(upload(m.get(), pack_lstm_s16(whh1, H, false, m->f16), &m->lstm.s16_hh));
            RMR_TRY(upload(m.get(), pack_lstm_s16(wih2, H, true, m->f16), &m->lstm.s16_ih2));
            RMR_TRY(upload(m.get(), pack_bias_s16(bih1, bhh1, H, false), &m->lstm.s16_b1));
            RMR_TRY(upload(m.get(), pack_bias_s16(bih2, bhh2, H, true), &m->lstm.s16_b2));
        }
        if (m->nparts == 1 && H == 64) {
            RMR_TRY(upload(m.get(), pack_lstm_x16(wih1, false, m->f16), &m->lstm.x_ih));
            RMR_TRY(upload(m.get(), pack_lstm_x16(whh1, false, m->f16), &m->lstm.x_hh));
            RMR_TRY(upload(m.get(), pack_lstm_x16(wih2, true, m->f16), &m->lstm.x_ih2));
            RMR_TRY(upload(m.get(), pack_bias_x16(bih1, bhh1, false), &m->lstm.x_b1));
            RMR_TRY(upload(m.get(), pack_bias_x16(bih2, bhh2, true), &m->lstm.x_b2));
        }
        std::vector<float> b1(4 * H), b2(3 * H);
        for (int i = 0; i < 4 * H; ++i)
            b1[i] = (float)(((double)bih1[i] + (double)bhh1[i]) * lstm1_gate_scale(i / H));
        for (int gi = 0; gi < 3; ++gi)
            for (int u = 0; u < H; ++u) b2[gi * H + u] = bih2[g3[gi] * H + u] + bhh2[g3[gi] * H + u];
        RMR_TRY(upload(m.get(), b1, &m->lstm.b1));
        RMR_TRY(upload(m.get(), b2, &m->lstm.b2));
        RMR_TRY(upload(m.get(), std::vector<float>(wfc, wfc + (size_t)desc->num_out * H), &m->lstm.w_fc));
        RMR_TRY(upload(m.get(), std::vector<float>(bfc, bfc + desc->num_out), &m->lstm.b_fc));
    } else {
        RMR_TRY(pack_conv(m.get(), convs[5], K_CONV_SEQ3, &m->seq3));
        RMR_TRY(pack_conv(m.get(), convs[6], K_CONV_MERGE1, &m->merge1));
        RMR_TRY(pack_conv(m.get(), convs[7], K_CONV_MERGE2, &m->merge2));
        RMR_TRY(pack_conv(m.get(), convs[8], K_CONV_MERGE3, &m->merge3));
        RMR_TRY(pack_conv(m.get(), convs[9], K_CONV_MERGE4, &m->merge4));
        const float *wfc = p; p += (size_t)desc->num_out * sz * 3;
        const float *bfc = p; p += desc->num_out;
        RMR_TRY(upload(m.get(), std::vector<float>(wfc, wfc + (size_t)desc->num_out * sz * 3), &m->w_fc));
        RMR_TRY(upload(m.get(), std::vector<float>(bfc, bfc + desc->num_out), &m->b_fc));
    }
    if ((size_t)(p - weights) != n_floats) RMR_FAIL(RMR_ERR_INVALID, "internal: blob walk mismatch");
    *out = m.release();
    return 0;
}

}  // extern "C"

// =========================================================================================
// fused pipeline
// =========================================================================================
namespace {

size_t act_floats_per_chunk(const rmr_model *m) {
    const size_t sz = m->desc.size;
    size_t n = (size_t)m->P1 * 16 + (size_t)m->P2 * 16 + (size_t)m->P3 * 2 * sz;
    if (m->desc.arch == RMR_ARCH_CONV_LSTM) {
        n += (size_t)m->T * sz;
    } else {
        n += (size_t)m->PQ2 * 32 + (size_t)(m->T + m->T2 + m->T3 + m->T4) * sz;
    }
    return n;
}

// enc != nullptr: dense seqs path; otherwise gather path from (seqs, maps, lens)
int run_pipeline(rmr_model *m, const float *signal, const float *enc, const int8_t *seqs, int seq_w,
                 const int16_t *maps, int map_w, const int16_t *lens, int kb, int ka, int64_t n,
                 float *logits) {
    rmr_engine *e = m->eng;
    if (n <= 0) return 0;
    if (m->nparts == 1 && m->desc.size > 64) {
        // bf16 / f16 above 64 channels (k_stream16.hip): fp32 front kernels (sig_conv1/2, seq_conv1: 16 channels), then the three
        // size-wide convolutions and the LSTM on the 16-bit matrix cores with streamed weights; cat and x are 16-bit in HBM
        const int sz = m->desc.size, L = m->L, EC = 4 * m->desc.kmer_len;
        int64_t sb = e->subbatch > 0 ? e->subbatch : 131072;
        if (sb > n) sb = n;
        const size_t front_fl = (size_t)(m->P1 + m->P2) * 16;
        const size_t cat_el = (size_t)m->P3 * 2 * sz, x_el = (size_t)m->T * sz;
        RMR_TRY(e->ensure(e->act, (front_fl * sizeof(float) + (cat_el + x_el) * sizeof(uint16_t) + 64) * sb));
        float *seq1 = reinterpret_cast<float *>(e->act.ptr);
        for (int64_t c0 = 0; c0 < n; c0 += sb) {
            const int64_t nb = (n - c0) < sb ? (n - c0) : sb;
            float *sig2 = seq1 + (size_t)nb * m->P1 * 16;
            uint16_t *cat = reinterpret_cast<uint16_t *>(seq1 + front_fl * sb);
            uint16_t *x16 = cat + cat_el * sb + 32;
            const float *sig_b = signal + (size_t)c0 * L;
            if (enc) {
                RMR_TRY(launch_front(m, e->stream, sig_b, nullptr, 0, nullptr, 0, nullptr, 0, 0, nb, sig2, nullptr));
                RMR_TRY(launch_seq1_dense(m, enc + (size_t)c0 * EC * L, nb, seq1));
            } else {
                RMR_TRY(launch_front(m, e->stream, sig_b, seqs + (size_t)c0 * seq_w, seq_w, maps + (size_t)c0 * map_w, map_w, lens + c0, kb, ka, nb,
                                     sig2, seq1));
            }
            RMR_TRY(launch_conv_stream16(m, m->sig3, sig2, false, m->P2, cat, 2 * sz, 0, m->P3, nb));
            RMR_TRY(launch_conv_stream16(m, m->seq2, seq1, false, m->P1, cat, 2 * sz, sz, m->P3, nb));
            RMR_TRY(launch_conv_stream16(m, m->merge1, cat, true, m->P3, x16, sz, 0, m->T, nb));
            RMR_TRY(launch_lstm_stream16(m, x16, nb, logits + (size_t)c0 * m->desc.num_out));
        }
        return 0;
    }
    if (m->f16 && (enc || !fused_front_supported(m, seq_w, map_w)))
        RMR_FAIL(RMR_ERR_INVALID, "dtype f16 runs on the fused kernels only: chunk arrays (not a dense one-hot tensor), sequence rows of at "
                                  "most 256 columns, a chunk length that is a multiple of 4");
    if (!enc && fused_front_supported(m, seq_w, map_w) && (m->f16 || tune_int("RMR_FUSED", 1))) {
        // plain-bf16 ConvLSTM: two launches per sub-batch, x (bf16, 3 KB/chunk @C100) is the only intermediate in
        // HBM; sub-batches are sized so that x stays in the 256 MiB Infinity Cache between producer and consumer
        int64_t sb = e->subbatch > 0 ? e->subbatch : 65536;
        if (sb > n) sb = n;
        const size_t x_elems = (size_t)m->T * m->desc.size;
        RMR_TRY(e->ensure(e->act, x_elems * sb * sizeof(uint16_t)));
        uint16_t *x16 = reinterpret_cast<uint16_t *>(e->act.ptr);
        for (int64_t c0 = 0; c0 < n; c0 += sb) {
            const int64_t nb = (n - c0) < sb ? (n - c0) : sb;
            RMR_TRY(launch_fused_front(m, signal + (size_t)c0 * m->L, seqs + (size_t)c0 * seq_w, seq_w,
                                       maps + (size_t)c0 * map_w, map_w, lens + c0, nb, x16));
#ifdef RMR_TIMING_ABLATIONS  // experiment build only (make abl; tools/stress_determinism.py): x of every sub-batch, appended
            if (const char *dump = getenv("RMR_FUSED_DUMP_X")) {
                std::vector<uint16_t> h(x_elems * nb);
                RMR_HIP(hipStreamSynchronize(e->stream));
                RMR_HIP(hipMemcpy(h.data(), x16, h.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
                if (FILE *f = fopen(dump, "ab")) {
                    fwrite(h.data(), sizeof(uint16_t), h.size(), f);
                    fclose(f);
                }
            }
            if (abl_int("RMR_DEBUG_SKIP_LSTM", 0)) continue;  // the front kernel alone (x through RMR_FUSED_DUMP_X)
#endif
            RMR_TRY(launch_lstm_head_x16(m, x16, nb, logits + (size_t)c0 * m->desc.num_out));
        }
        return 0;
    }
    const size_t per = act_floats_per_chunk(m);
    // 262144 chunks per sub-batch: the tail of every persistent-block kernel is paid half as often as with 131072
    // (+1.2 % measured; 524288: +0.2 % more for twice the 5.5 GB arena)
    int64_t sb = e->subbatch > 0 ? e->subbatch : 262144;
    if (sb > n) sb = n;
    const int sz = m->desc.size, L = m->L, EC = 4 * m->desc.kmer_len;
    // fp32 ConvLSTM size 64 straight from the chunk arrays: sig_conv1/2 and seq_conv1 are produced inside the staging of
    // sig_conv3 / seq_conv2 (k_conv_front.hip); sig2 / seq1 never exist in HBM.  (RMR_CONV_FRONT=0: the separate front
    // kernels - the comparand of tests/test_gpu_conv_front.py)
    const bool fold = !enc && tune_int("RMR_CONV_FRONT", 1) && conv_front_supported(m, kb, ka, seq_w, map_w);
    // every other fp32 path (Conv_w_ref; ConvLSTM shapes the two-branch fold does not cover): the signal branch alone is
    // folded - sig_conv1 / sig_conv2 (matrix cores) produced inside the staging of sig_conv3, sig2 never in HBM
    const bool sigfold = !fold && m->nparts == 0 && sig3_front_mfma_supported(m);
    // (Running the front kernels of sub-batch i + 1 on a second stream under the matrix kernels of sub-batch i was measured in
    //  rounds 1-2 in two forms and gained nothing - they share the CUs with conv_sig3, or half a register file under the LSTM -
    //  and is gone; profiles/NOTES_r03.md.)
    const size_t front_fl = fold ? 0 : (size_t)(m->P1 + m->P2) * 16;
    RMR_TRY(e->ensure(e->act, (per + front_fl) * sb * sizeof(float)));
    float *arena = reinterpret_cast<float *>(e->act.ptr);
    float *seq1 = arena, *rest = arena + front_fl * sb;
    const bool split_conv = m->nparts > 0;
    for (int64_t c0 = 0; c0 < n; c0 += sb) {
        const int64_t nb = (n - c0) < sb ? (n - c0) : sb;
        float *sig2 = seq1 + (size_t)nb * m->P1 * 16;
        const float *sig_b = signal + (size_t)c0 * L;
        if (!fold) {  // sig_conv1/2 -> sig2, seq_conv1 -> seq1 (k_front.hip)
            if (enc) {
                if (!sigfold) RMR_TRY(launch_front(m, e->stream, sig_b, nullptr, 0, nullptr, 0, nullptr, 0, 0, nb, sig2, nullptr));
                RMR_TRY(launch_seq1_dense(m, enc + (size_t)c0 * EC * L, nb, seq1));
            } else {
                RMR_TRY(launch_front(m, e->stream, sig_b, seqs + (size_t)c0 * seq_w, seq_w, maps + (size_t)c0 * map_w, map_w, lens + c0, kb, ka,
                                     nb, sigfold ? nullptr : sig2, seq1));
            }
        }
        float *base = rest;
        float *cat = base; base += (size_t)nb * m->P3 * 2 * sz;
        if (fold) RMR_TRY(launch_conv_front(m, sig_b, seqs + (size_t)c0 * seq_w, seq_w, maps + (size_t)c0 * map_w, map_w, lens + c0, nb, cat));
        else if (split_conv) RMR_TRY(launch_conv_split(e, m->sig3, m->nparts, sig2, 16, m->P2, cat, 2 * sz, 0, m->P3, nb));
        else if (sigfold) RMR_TRY(launch_sig3_front_mfma(m, sig_b, nb, cat));
        else RMR_TRY(launch_conv(e, m->sig3, sig2, 16, m->P2, cat, 2 * sz, 0, m->P3, nb));
#ifdef RMR_TIMING_ABLATIONS  // experiment build only (tools/stress_determinism.py): cat [nb][P3][2 sz] of the last sub-batch
        if (const char *dump = fold ? getenv("RMR_DUMP_CAT") : nullptr) {
            std::vector<float> h((size_t)nb * m->P3 * 2 * sz);
            RMR_HIP(hipStreamSynchronize(e->stream));
            RMR_HIP(hipMemcpy(h.data(), cat, h.size() * sizeof(float), hipMemcpyDeviceToHost));
            if (FILE *f = fopen(dump, "wb")) {
                fwrite(h.data(), sizeof(float), h.size(), f);
                fclose(f);
            }
        }
#endif
        if (m->desc.arch == RMR_ARCH_CONV_LSTM) {
            float *x = base; base += (size_t)nb * m->T * sz;
            if (fold) {
                RMR_TRY(launch_conv(e, m->merge1, cat, 2 * sz, m->P3, x, sz, 0, m->T, nb));
            } else if (split_conv) {
                RMR_TRY(launch_conv_split(e, m->seq2, m->nparts, seq1, 16, m->P1, cat, 2 * sz, sz, m->P3, nb));
                RMR_TRY(launch_conv_split(e, m->merge1, m->nparts, cat, 2 * sz, m->P3, x, sz, 0, m->T, nb));
            } else {
                RMR_TRY(launch_conv(e, m->seq2, seq1, 16, m->P1, cat, 2 * sz, sz, m->P3, nb));
                RMR_TRY(launch_conv(e, m->merge1, cat, 2 * sz, m->P3, x, sz, 0, m->T, nb));
            }
            if (m->nparts > 0 && lstm_x16s_supported(m)) RMR_TRY(launch_lstm_head_x16s(m, x, nb, logits + (size_t)c0 * m->desc.num_out));
            else if (m->nparts > 0) RMR_TRY(launch_lstm_head_split(m, x, nb, logits + (size_t)c0 * m->desc.num_out));
            else RMR_TRY(launch_lstm_head(m, x, nb, logits + (size_t)c0 * m->desc.num_out));
        } else {
            float *seq2 = base; base += (size_t)nb * m->PQ2 * 32;
            float *m1 = base; base += (size_t)nb * m->T * sz;
            float *m2 = base; base += (size_t)nb * m->T2 * sz;
            float *m3 = base; base += (size_t)nb * m->T3 * sz;
            float *m4 = base; base += (size_t)nb * m->T4 * sz;
            RMR_TRY(launch_conv(e, m->seq2, seq1, 16, m->P1, seq2, 32, 0, m->PQ2, nb));
            RMR_TRY(launch_conv(e, m->seq3, seq2, 32, m->PQ2, cat, 2 * sz, sz, m->P3, nb));
            RMR_TRY(launch_conv(e, m->merge1, cat, 2 * sz, m->P3, m1, sz, 0, m->T, nb));
            RMR_TRY(launch_conv(e, m->merge2, m1, sz, m->T, m2, sz, 0, m->T2, nb));
            RMR_TRY(launch_conv(e, m->merge3, m2, sz, m->T2, m3, sz, 0, m->T3, nb));
            RMR_TRY(launch_conv(e, m->merge4, m3, sz, m->T3, m4, sz, 0, m->T4, nb));
            RMR_TRY(launch_fc_head(m, m4, nb, logits + (size_t)c0 * m->desc.num_out));
        }
    }
    return 0;
}

// host <-> device staging helper: a bump allocator over the engine's staging arena
struct Stage {
    rmr_engine *e;
    char *base = nullptr;
    size_t off = 0, cap = 0;
    int init(size_t bytes) {
        RMR_TRY(e->ensure(e->staging, bytes));
        base = reinterpret_cast<char *>(e->staging.ptr);
        cap = bytes;
        return 0;
    }
    template <typename T>
    T *take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T *p = reinterpret_cast<T *>(base + off);
        off += count * sizeof(T);
        return p;
    }
    static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }
};

#define H2D(dst, src, bytes) do { if ((bytes) > 0) RMR_HIP(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyHostToDevice, e->stream)); } while (0)
#define D2H(dst, src, bytes) do { if ((bytes) > 0) RMR_HIP(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToHost, e->stream)); } while (0)

}  // namespace

extern "C" {

int rmr_encode_kmers(rmr_engine *e, int kb, int ka, const int8_t *seqs, int seq_w,
                     const int16_t *maps, int map_w, const int16_t *lens, int64_t n, int sig_len,
                     float *out, int mem) {
    if (!e || !seqs || !maps || !lens || !out) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (kb < 0 || ka < 0 || n < 0 || sig_len <= 0 || seq_w <= 0 || map_w <= 0)
        RMR_FAIL(RMR_ERR_INVALID, "bad sizes");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    const size_t out_b = (size_t)n * 4 * (kb + ka + 1) * sig_len * sizeof(float);
    if (mem == RMR_MEM_DEVICE) return launch_encode(e, kb, ka, seqs, seq_w, maps, map_w, lens, n, sig_len, out);
    Stage st{e};
    RMR_TRY(st.init(Stage::pad(n * seq_w) + Stage::pad(n * map_w * 2) + Stage::pad(n * 2) + Stage::pad(out_b) + 1024));
    int8_t *ds = st.take<int8_t>(n * seq_w);
    int16_t *dm = st.take<int16_t>(n * map_w);
    int16_t *dl = st.take<int16_t>(n);
    float *dout = st.take<float>(out_b / 4);
    H2D(ds, seqs, (size_t)n * seq_w);
    H2D(dm, maps, (size_t)n * map_w * 2);
    H2D(dl, lens, (size_t)n * 2);
    RMR_TRY(launch_encode(e, kb, ka, ds, seq_w, dm, map_w, dl, n, sig_len, dout));
    D2H(out, dout, out_b);
    RMR_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

int rmr_trim_chunk_context(rmr_engine *e, int sb, int sa, int cb, int ca, int tsc, int8_t *seqs,
                           int seq_w, int16_t *maps, int map_w, int16_t *lens, int64_t n, int mem) {
    if (!e || !seqs || !maps || !lens) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (n <= 0) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    if (mem == RMR_MEM_DEVICE) return launch_trim(e, sb, sa, cb, ca, tsc, seqs, seq_w, maps, map_w, lens, n);
    Stage st{e};
    RMR_TRY(st.init(Stage::pad(n * seq_w) + Stage::pad(n * map_w * 2) + Stage::pad(n * 2) + 1024));
    int8_t *ds = st.take<int8_t>(n * seq_w);
    int16_t *dm = st.take<int16_t>(n * map_w);
    int16_t *dl = st.take<int16_t>(n);
    H2D(ds, seqs, (size_t)n * seq_w);
    H2D(dm, maps, (size_t)n * map_w * 2);
    H2D(dl, lens, (size_t)n * 2);
    RMR_TRY(launch_trim(e, sb, sa, cb, ca, tsc, ds, seq_w, dm, map_w, dl, n));
    D2H(seqs, ds, (size_t)n * seq_w);
    D2H(maps, dm, (size_t)n * map_w * 2);
    D2H(lens, dl, (size_t)n * 2);
    RMR_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

int rmr_parse_moves(rmr_engine *e, const int8_t *mv_tag, int64_t mv_tag_len, int64_t sig_len,
                    int64_t seq_len, int check, int reverse_signal, int64_t *q2s, int64_t *n_out,
                    int mem) {
    if (!e || !mv_tag || !q2s || !n_out) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (mv_tag_len < 1) RMR_FAIL(RMR_ERR_INVALID, "empty move tag");
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    Stage st{e};
    RMR_TRY(st.init(Stage::pad(mv_tag_len) + Stage::pad((mv_tag_len + 1) * 8) + 2048));
    int64_t *dcount = st.take<int64_t>(1);
    int8_t stride_h = 0;
    const int8_t *dmv = mv_tag;
    int64_t *dq = q2s;
    if (mem == RMR_MEM_HOST) {
        int8_t *t = st.take<int8_t>(mv_tag_len);
        H2D(t, mv_tag, (size_t)mv_tag_len);
        dmv = t;
        dq = st.take<int64_t>(mv_tag_len + 1);
        stride_h = mv_tag[0];
    } else {
        RMR_HIP(hipMemcpyAsync(&stride_h, mv_tag, 1, hipMemcpyDeviceToHost, e->stream));
    }
    RMR_TRY(launch_moves(e, dmv, mv_tag_len, sig_len, reverse_signal, dq, dcount));
    int64_t cnt = 0;
    D2H(&cnt, dcount, 8);
    RMR_HIP(hipStreamSynchronize(e->stream));
    if (mem == RMR_MEM_HOST) {
        RMR_HIP(hipMemcpy(q2s, dq, (size_t)cnt * 8, hipMemcpyDeviceToHost));
    }
    *n_out = cnt;
    if (stride_h <= 0) RMR_FAIL(RMR_ERR_INVALID, "move table stride %d", (int)stride_h);
    if (check && seq_len >= 0 && cnt - 1 != seq_len) {
        set_error("Move table discordant with basecalls");
        return RMR_ERR_DISCORDANT_SEQ;
    }
    if (check && (mv_tag_len - 1) != sig_len / stride_h) {
        set_error("Move table discordant with signal");
        return RMR_ERR_DISCORDANT_SIG;
    }
    return 0;
}

int rmr_parse_moves_batch(rmr_engine *e, const int8_t *mv_tags, const int64_t *mv_off, const int64_t *sig_len,
                          const int64_t *seq_len, int64_t n_reads, int check, int reverse_signal, int64_t *q2s,
                          int64_t *counts, int32_t *status, int mem) {
    if (!e || !mv_tags || !mv_off || !sig_len || !seq_len || !q2s || !counts || !status)
        RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (n_reads <= 0) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    if (mem == RMR_MEM_DEVICE)
        return launch_moves_batch(e, mv_tags, mv_off, sig_len, seq_len, n_reads, check, reverse_signal, q2s, counts, status);
    const int64_t total = mv_off[n_reads];
    if (mv_off[0] != 0 || total < n_reads) RMR_FAIL(RMR_ERR_INVALID, "bad move table offsets");
    Stage st{e};
    RMR_TRY(st.init(Stage::pad(total) + Stage::pad((size_t)total * 8) + 5 * Stage::pad((size_t)(n_reads + 1) * 8) + 4096));
    int8_t *dmv = st.take<int8_t>(total);
    int64_t *doff = st.take<int64_t>(n_reads + 1);
    int64_t *dsl = st.take<int64_t>(n_reads);
    int64_t *dql = st.take<int64_t>(n_reads);
    int64_t *dq = st.take<int64_t>(total);
    int64_t *dcnt = st.take<int64_t>(n_reads);
    int32_t *dst = st.take<int32_t>(n_reads);
    H2D(dmv, mv_tags, (size_t)total);
    H2D(doff, mv_off, (size_t)(n_reads + 1) * 8);
    H2D(dsl, sig_len, (size_t)n_reads * 8);
    H2D(dql, seq_len, (size_t)n_reads * 8);
    RMR_TRY(launch_moves_batch(e, dmv, doff, dsl, dql, n_reads, check, reverse_signal, dq, dcnt, dst));
    D2H(q2s, dq, (size_t)total * 8);
    D2H(counts, dcnt, (size_t)n_reads * 8);
    D2H(status, dst, (size_t)n_reads * 4);
    RMR_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

int rmr_signal_histograms(rmr_engine *e, const int16_t *signal, const int64_t *start, const int64_t *len, int64_t n, int32_t *lo, int32_t *hi,
                          const int64_t *hist_off, uint32_t *hist) {
    if (!e || !signal || !start || !len || !lo || !hi || (hist != nullptr) != (hist_off != nullptr)) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (n < 0 || n > (int64_t)1 << 24) RMR_FAIL(RMR_ERR_INVALID, "bad n");
    if (n == 0) return 0;
    for (int64_t i = 0; i < n; ++i)
        if (start[i] < 0 || len[i] < 0) RMR_FAIL(RMR_ERR_INVALID, "span %lld: negative extent", (long long)i);
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    const size_t nn = (size_t)n;
    Stage st{e};
    if (!hist) {  // pass 1: the range of every span
        RMR_TRY(st.init(2 * Stage::pad(nn * 8) + 2 * Stage::pad(nn * 4) + 4096));
        int64_t *d_start = st.take<int64_t>(nn), *d_len = st.take<int64_t>(nn);
        int32_t *d_lo = st.take<int32_t>(nn), *d_hi = st.take<int32_t>(nn);
        H2D(d_start, start, nn * 8);
        H2D(d_len, len, nn * 8);
        RMR_TRY(launch_signal_range(e, signal, d_start, d_len, n, d_lo, d_hi));
        D2H(lo, d_lo, nn * 4);
        D2H(hi, d_hi, nn * 4);
        RMR_HIP(hipStreamSynchronize(e->stream));
        return 0;
    }
    // pass 2: counts over [lo, hi] of every span, at the offsets the caller summed up
    if (hist_off[0] != 0) RMR_FAIL(RMR_ERR_INVALID, "hist_off[0] != 0");
    for (int64_t i = 0; i < n; ++i) {
        const int64_t width = len[i] > 0 ? (int64_t)hi[i] - lo[i] + 1 : 0;
        if (hist_off[i + 1] - hist_off[i] != (width > 0 ? width : 0)) RMR_FAIL(RMR_ERR_INVALID, "span %lld: hist_off does not match hi - lo + 1", (long long)i);
    }
    const size_t total = (size_t)hist_off[n];
    if (total == 0) return 0;
    RMR_TRY(st.init(2 * Stage::pad(nn * 8) + Stage::pad(nn * 4) + Stage::pad((nn + 1) * 8) + Stage::pad(total * 4) + 4096));
    int64_t *d_start = st.take<int64_t>(nn), *d_len = st.take<int64_t>(nn);
    int32_t *d_lo = st.take<int32_t>(nn);
    int64_t *d_off = st.take<int64_t>(nn + 1);
    uint32_t *d_hist = st.take<uint32_t>(total);
    H2D(d_start, start, nn * 8);
    H2D(d_len, len, nn * 8);
    H2D(d_lo, lo, nn * 4);
    H2D(d_off, hist_off, (nn + 1) * 8);
    RMR_HIP(hipMemsetAsync(d_hist, 0, total * 4, e->stream));
    RMR_TRY(launch_signal_hist(e, signal, d_start, d_len, d_lo, d_off, n, d_hist));
    D2H(hist, d_hist, total * 4);
    RMR_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

int rmr_assemble_reads(rmr_engine *e, int64_t n_reads, const int16_t *signal, const int64_t *src_start, const int64_t *q2s,
                       const int64_t *q2s_off, const int64_t *seq_len, int16_t *dacs, int64_t dacs_cap, int64_t *s2s,
                       int64_t *d_sig_off, int64_t *d_seq_off, int64_t *sig_off) {
    if (!e || !signal || !src_start || !q2s || !q2s_off || !seq_len || !dacs || !s2s || !d_sig_off || !d_seq_off || !sig_off)
        RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (n_reads < 0) RMR_FAIL(RMR_ERR_INVALID, "bad sizes");
    sig_off[0] = 0;
    if (n_reads == 0) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    const size_t n = (size_t)n_reads;
    Stage st{e};
    RMR_TRY(st.init(4 * Stage::pad(n * 8) + 4096));
    int64_t *d_start = st.take<int64_t>(n), *d_qoff = st.take<int64_t>(n), *d_slen = st.take<int64_t>(n), *d_len = st.take<int64_t>(n);
    H2D(d_start, src_start, n * 8);
    H2D(d_qoff, q2s_off, n * 8);
    H2D(d_slen, seq_len, n * 8);
    RMR_TRY(launch_assemble_lengths(e, q2s, d_qoff, d_slen, n_reads, d_len));
    std::vector<int64_t> len(n), seq_off(n + 1, 0);
    D2H(len.data(), d_len, n * 8);
    RMR_HIP(hipStreamSynchronize(e->stream));
    for (size_t i = 0; i < n; ++i) {
        if (len[i] < 0 || seq_len[i] < 0) RMR_FAIL(RMR_ERR_INVALID, "read %zu: a mapping that runs backwards", i);
        sig_off[i + 1] = sig_off[i] + len[i];
        seq_off[i + 1] = seq_off[i] + seq_len[i];
    }
    if (sig_off[n] > dacs_cap) RMR_FAIL(RMR_ERR_INVALID, "dacs capacity %lld < %lld samples", (long long)dacs_cap, (long long)sig_off[n]);
    H2D(d_sig_off, sig_off, (n + 1) * 8);
    H2D(d_seq_off, seq_off.data(), (n + 1) * 8);
    RMR_TRY(launch_assemble_reads(e, signal, d_start, q2s, d_qoff, d_sig_off, d_seq_off, n_reads, dacs, s2s));
    RMR_HIP(hipStreamSynchronize(e->stream));  // (the pageable offset vectors above are read by the copies)
    return 0;
}

}  // extern "C"

// ---- chunk extraction ------------------------------------------------------------------------
namespace {

// device-side copy of an rmr_reads whose arrays live on the host
struct DevReads {
    rmr_reads d{};
    int32_t *chunk_read = nullptr;
    int32_t *sig_read = nullptr;
    int64_t n_chunks = 0, total_sig = 0, total_bases = 0;
};

int read_offsets_host(rmr_engine *e, const rmr_reads *r, int mem, std::vector<int64_t> &sig_off,
                      std::vector<int64_t> &seq_off, std::vector<int64_t> &foc_off) {
    const size_t n1 = (size_t)r->n_reads + 1;
    sig_off.resize(n1); seq_off.resize(n1); foc_off.resize(n1);
    if (mem == RMR_MEM_HOST) {
        memcpy(sig_off.data(), r->sig_off, n1 * 8);
        memcpy(seq_off.data(), r->seq_off, n1 * 8);
        memcpy(foc_off.data(), r->focus_off, n1 * 8);
    } else if (r->host_sig_off && r->host_seq_off && r->host_focus_off) {
        // the caller kept host copies of the offsets: three small device-to-host copies (and their syncs) saved per call
        memcpy(sig_off.data(), r->host_sig_off, n1 * 8);
        memcpy(seq_off.data(), r->host_seq_off, n1 * 8);
        memcpy(foc_off.data(), r->host_focus_off, n1 * 8);
    } else {
        RMR_HIP(hipMemcpy(sig_off.data(), r->sig_off, n1 * 8, hipMemcpyDeviceToHost));
        RMR_HIP(hipMemcpy(seq_off.data(), r->seq_off, n1 * 8, hipMemcpyDeviceToHost));
        RMR_HIP(hipMemcpy(foc_off.data(), r->focus_off, n1 * 8, hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i + 1 < n1; ++i)
        if (sig_off[i + 1] < sig_off[i] || seq_off[i + 1] < seq_off[i] || foc_off[i + 1] < foc_off[i])
            RMR_FAIL(RMR_ERR_INVALID, "offsets of read %zu are not increasing", i);
    if (sig_off[0] != 0 || seq_off[0] != 0 || foc_off[0] != 0) RMR_FAIL(RMR_ERR_INVALID, "offsets must start at 0");
    return 0;
}

int stage_reads(rmr_engine *e, Stage &st, const rmr_reads *r, int mem, bool need_dacs, const std::vector<int64_t> &sig_off,
                const std::vector<int64_t> &seq_off, const std::vector<int64_t> &foc_off, DevReads *o) {
    const int64_t nr = r->n_reads;
    o->total_sig = sig_off[nr];
    o->total_bases = seq_off[nr];
    o->n_chunks = foc_off[nr];
    o->d = *r;
    o->chunk_read = st.take<int32_t>(o->n_chunks + 1);
    if (mem == RMR_MEM_DEVICE) {
        // device-resident batch: the read index of every chunk comes from the offsets where they are - no host loop, no
        // upload, no wait (a batch of the reads pipeline paid two of these round trips per extraction, each behind whatever
        // the GPU was running)
        return launch_chunk_read(e, r->focus_off, nr, o->n_chunks, o->chunk_read);
    }
    // read index per chunk (host-built, tiny next to the data itself)
    std::vector<int32_t> cr((size_t)o->n_chunks);
    for (int64_t k = 0; k < nr; ++k)
        for (int64_t i = foc_off[k]; i < foc_off[k + 1]; ++i) cr[(size_t)i] = (int32_t)k;
    H2D(o->chunk_read, cr.data(), cr.size() * 4);
    if (mem == RMR_MEM_HOST) {
#define STAGE_ARR(field, T, count)                                        \
    {                                                                     \
        T *d_ = st.take<T>((count) + 1);                                  \
        H2D(d_, r->field, (size_t)(count) * sizeof(T));                   \
        o->d.field = d_;                                                  \
    }
        if (need_dacs) STAGE_ARR(dacs, int16_t, o->total_sig)
        STAGE_ARR(sig_off, int64_t, nr + 1)
        STAGE_ARR(seq_to_sig, int64_t, o->total_bases + nr)
        STAGE_ARR(int_seq, int8_t, o->total_bases)
        STAGE_ARR(seq_off, int64_t, nr + 1)
        STAGE_ARR(shift, double, nr)
        STAGE_ARR(scale, double, nr)
        STAGE_ARR(focus_bases, int64_t, o->n_chunks)
        STAGE_ARR(focus_off, int64_t, nr + 1)
#undef STAGE_ARR
    }
    // the H2D copies above read from host vectors that die with this frame
    RMR_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

size_t reads_stage_bytes(const rmr_reads *r, int64_t total_sig, int64_t total_bases, int64_t n_chunks) {
    const int64_t nr = r->n_reads;
    return Stage::pad(total_sig * 2) + Stage::pad(total_sig * 4 + 8) + 4 * Stage::pad((nr + 2) * 8) +
           Stage::pad((total_bases + nr + 1) * 8) + Stage::pad(total_bases + 1) + 2 * Stage::pad((nr + 1) * 8) +
           Stage::pad((n_chunks + 1) * 8) + Stage::pad((n_chunks + 1) * 4) + 8192;
}

}  // namespace

extern "C" {

int rmr_chunk_geometry(rmr_engine *e, const rmr_reads *reads, float *sig_out, int64_t *geo,
                       int64_t *max_seq_len, int mem) {
    if (!e || !reads || !sig_out || !geo || !max_seq_len) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (reads->n_reads < 0) RMR_FAIL(RMR_ERR_INVALID, "n_reads < 0");
    *max_seq_len = 0;
    if (reads->n_reads == 0) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    std::vector<int64_t> so, qo, fo;
    RMR_TRY(read_offsets_host(e, reads, mem, so, qo, fo));
    const int64_t nr = reads->n_reads, ts = so[nr], tb = qo[nr], nc = fo[nr];
    Stage st{e};
    RMR_TRY(st.init(reads_stage_bytes(reads, ts, tb, nc) + Stage::pad(ts * 4) + Stage::pad(nc * 48) + 4096));
    DevReads dr;
    RMR_TRY(stage_reads(e, st, reads, mem, true, so, qo, fo, &dr));
    int *dmax = st.take<int>(4);
    RMR_HIP(hipMemsetAsync(dmax, 0, 16, e->stream));
    float *dsig = sig_out;
    int64_t *dgeo = geo;
    if (mem == RMR_MEM_HOST) {
        dsig = st.take<float>(ts + 1);
        dgeo = st.take<int64_t>(nc * 6 + 1);
    }
    RMR_TRY(launch_geometry(e, dr.d, nc, dr.chunk_read, dsig, ts, dr.sig_read, dgeo, dmax));
    int hmax = 0;
    D2H(&hmax, dmax, 4);
    if (mem == RMR_MEM_HOST) {
        D2H(sig_out, dsig, (size_t)ts * 4);
        D2H(geo, dgeo, (size_t)nc * 48);
    }
    RMR_HIP(hipStreamSynchronize(e->stream));
    *max_seq_len = hmax;
    return 0;
}

int rmr_chunk_fill(rmr_engine *e, const rmr_reads *reads, const float *sig, const int64_t *geo,
                   float *signal, int8_t *seqs, int seq_w, int16_t *maps, int map_w, int16_t *lens,
                   int64_t *read_focus_bases, int mem) {
    if (!e || !reads || !sig || !geo || !signal || !seqs || !maps || !lens || !read_focus_bases)
        RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (reads->n_reads <= 0) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    std::vector<int64_t> so, qo, fo;
    RMR_TRY(read_offsets_host(e, reads, mem, so, qo, fo));
    const int64_t nr = reads->n_reads, ts = so[nr], tb = qo[nr], nc = fo[nr];
    if (nc == 0) return 0;
    const int L = reads->cc_before + reads->cc_after;
    Stage st{e};
    RMR_TRY(st.init(reads_stage_bytes(reads, ts, tb, nc) + Stage::pad(ts * 4) + Stage::pad(nc * 48) +
                    Stage::pad((size_t)nc * L * 4) + Stage::pad((size_t)nc * seq_w) +
                    Stage::pad((size_t)nc * map_w * 2) + Stage::pad(nc * 2) + Stage::pad(nc * 8) + 8192));
    DevReads dr;
    RMR_TRY(stage_reads(e, st, reads, mem, false, so, qo, fo, &dr));
    if (mem == RMR_MEM_DEVICE)
        return launch_fill(e, dr.d, nc, dr.chunk_read, sig, geo, signal, seqs, seq_w, maps, map_w, lens,
                           read_focus_bases);
    float *dsig = st.take<float>(ts + 1);
    int64_t *dgeo = st.take<int64_t>(nc * 6);
    float *dsignal = st.take<float>((size_t)nc * L);
    int8_t *dseqs = st.take<int8_t>((size_t)nc * seq_w);
    int16_t *dmaps = st.take<int16_t>((size_t)nc * map_w);
    int16_t *dlens = st.take<int16_t>(nc);
    int64_t *drfb = st.take<int64_t>(nc);
    H2D(dsig, sig, (size_t)ts * 4);
    H2D(dgeo, geo, (size_t)nc * 48);
    RMR_TRY(launch_fill(e, dr.d, nc, dr.chunk_read, dsig, dgeo, dsignal, dseqs, seq_w, dmaps, map_w, dlens, drfb));
    D2H(signal, dsignal, (size_t)nc * L * 4);
    D2H(seqs, dseqs, (size_t)nc * seq_w);
    D2H(maps, dmaps, (size_t)nc * map_w * 2);
    D2H(lens, dlens, (size_t)nc * 2);
    D2H(read_focus_bases, drfb, (size_t)nc * 8);
    RMR_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

int rmr_count_labels(rmr_engine *e, const float *logits, int64_t n, int num_out, int64_t *counts,
                     int mem) {
    if (!e || !logits || !counts) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (num_out < 1 || num_out > 16) RMR_FAIL(RMR_ERR_INVALID, "num_out %d not in [1,16]", num_out);
    if (n <= 0) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    if (mem == RMR_MEM_DEVICE) return launch_count(e, logits, n, num_out, counts);
    Stage st{e};
    RMR_TRY(st.init(Stage::pad((size_t)n * num_out * 4) + 4096));
    float *dl = st.take<float>((size_t)n * num_out);
    int64_t *dc = st.take<int64_t>(16);
    H2D(dl, logits, (size_t)n * num_out * 4);
    H2D(dc, counts, (size_t)num_out * 8);
    RMR_TRY(launch_count(e, dl, n, num_out, dc));
    D2H(counts, dc, (size_t)num_out * 8);
    RMR_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

int rmr_validation_tally(rmr_engine *e, const float *logits, const int64_t *labels, int64_t n, int num_out, int num_labels,
                         const int32_t *label_of_column, int64_t *confusion, float *win_prob, uint8_t *call, double *loss_sum) {
    if (!e || !logits || !labels || !label_of_column || !confusion || !win_prob || !call || !loss_sum)
        RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (num_out < 1 || num_out > 16 || num_labels < num_out || num_labels > 16)
        RMR_FAIL(RMR_ERR_INVALID, "num_out %d / num_labels %d not in [1,16], num_labels >= num_out", num_out, num_labels);
    for (int c = 0; c < num_labels; ++c)
        if (label_of_column[c] < -1 || label_of_column[c] >= num_out) RMR_FAIL(RMR_ERR_INVALID, "label_of_column[%d] = %d", c, label_of_column[c]);
    if (n <= 0) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    return launch_validation_tally(e, logits, labels, n, num_out, num_labels, label_of_column, confusion, win_prob, call, loss_sum);
}

static int check_motifs(const rmr_motif_set *motifs) {
    if (motifs->n_motifs < 1 || motifs->n_motifs > 8) RMR_FAIL(RMR_ERR_INVALID, "1..8 motifs supported");
    for (int m = 0; m < motifs->n_motifs; ++m)
        if (motifs->len[m] < 1 || motifs->len[m] > 16 || motifs->focus_pos[m] >= motifs->len[m] || motifs->focus_pos[m] < -64)
            RMR_FAIL(RMR_ERR_INVALID, "motif %d: length %d / focus %d unsupported", m, motifs->len[m], motifs->focus_pos[m]);
    return 0;
}

int rmr_motif_focus_counts(rmr_engine *e, const int8_t *int_seq, const int64_t *seq_off, int64_t n_reads, const rmr_motif_set *motifs,
                           int64_t *counts) {
    if (!e || !int_seq || !seq_off || !motifs || !counts) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (n_reads < 0 || n_reads > (int64_t)1 << 30) RMR_FAIL(RMR_ERR_INVALID, "bad n_reads");
    RMR_TRY(check_motifs(motifs));
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    return launch_motif_focus(e, int_seq, seq_off, (int)n_reads, *motifs, counts, nullptr, nullptr);
}

int rmr_motif_focus_fill(rmr_engine *e, const int8_t *int_seq, const int64_t *seq_off, int64_t n_reads, const rmr_motif_set *motifs,
                         const int64_t *foc_off, int64_t *focus) {
    if (!e || !int_seq || !seq_off || !motifs || !foc_off || !focus) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (n_reads < 0 || n_reads > (int64_t)1 << 30) RMR_FAIL(RMR_ERR_INVALID, "bad n_reads");
    RMR_TRY(check_motifs(motifs));
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    return launch_motif_focus(e, int_seq, seq_off, (int)n_reads, *motifs, nullptr, foc_off, focus);
}

// ---- the one collective: RCCL, loaded on first use (a single-GPU process never touches it) ----------------
namespace {
struct NcclId { char internal[RMR_COMM_ID_BYTES]; };  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(void **, int, NcclId, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
constexpr int kNcclInt64 = 4, kNcclSum = 0;  // rccl.h ncclDataType_t / ncclRedOp_t

int rccl_api(Rccl **out) {
    static Rccl api;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!api.lib) {
        // the RCCL already in the process (PyTorch-ROCm ships one bound to the HIP runtime this process uses), else ROCm's
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *h = nullptr;
        for (const char *nm : names)
            if ((h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;
        for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!h) RMR_FAIL(RMR_ERR_INVALID, "cannot load librccl (%s)", dlerror());
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy || !api.GetErrorString)
            RMR_FAIL(RMR_ERR_INVALID, "librccl lacks an expected symbol");
        api.lib = h;
    }
    *out = &api;
    return 0;
}
#define RMR_NCCL(api, expr)                                                                  \
    do {                                                                                     \
        const int _r = (expr);                                                               \
        if (_r != 0) RMR_FAIL(RMR_ERR_HIP, "RCCL error %s (%s)", (api)->GetErrorString(_r), #expr); \
    } while (0)
}  // namespace

}  // extern "C"
namespace rmr {
void rccl_comm_free(void *comm) {
    Rccl *r;
    if (comm && rccl_api(&r) == 0) (void)r->CommDestroy(comm);
}
}  // namespace rmr
extern "C" {

int rmr_comm_unique_id(uint8_t id[RMR_COMM_ID_BYTES]) {
    if (!id) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    Rccl *r;
    RMR_TRY(rccl_api(&r));
    NcclId u;
    RMR_NCCL(r, r->GetUniqueId(&u));
    memcpy(id, u.internal, RMR_COMM_ID_BYTES);
    return 0;
}

int rmr_comm_init(rmr_engine *e, const uint8_t id[RMR_COMM_ID_BYTES], int rank, int world) {
    if (!e || !id) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) RMR_FAIL(RMR_ERR_INVALID, "rank %d / world %d", rank, world);
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->comm) RMR_FAIL(RMR_ERR_INVALID, "engine already has a communicator (rmr_comm_destroy first)");
    Rccl *r;
    RMR_TRY(rccl_api(&r));
    RMR_HIP(hipSetDevice(e->device));
    NcclId u;
    memcpy(u.internal, id, RMR_COMM_ID_BYTES);
    void *comm = nullptr;
    RMR_NCCL(r, r->CommInitRank(&comm, world, u, rank));
    e->comm = comm;
    e->comm_rank = rank;
    e->comm_world = world;
    return 0;
}

int rmr_comm_destroy(rmr_engine *e) {
    if (!e) RMR_FAIL(RMR_ERR_INVALID, "engine is NULL");
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->comm) return 0;
    Rccl *r;
    RMR_TRY(rccl_api(&r));
    RMR_HIP(hipSetDevice(e->device));
    RMR_HIP(hipStreamSynchronize(e->stream));
    void *c = e->comm;
    e->comm = nullptr;
    e->comm_world = 1;
    e->comm_rank = 0;
    RMR_NCCL(r, r->CommDestroy(c));
    return 0;
}

int rmr_allreduce_counts(rmr_engine *e, int64_t *counts, int n, int mem) {
    if (!e || !counts) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (n < 1 || n > 4096) RMR_FAIL(RMR_ERR_INVALID, "n %d not in [1,4096]", n);
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->comm || e->comm_world == 1) return 0;  // one process: the sum over ranks is the input
    Rccl *r;
    RMR_TRY(rccl_api(&r));
    RMR_HIP(hipSetDevice(e->device));
    if (mem == RMR_MEM_DEVICE) {
        RMR_NCCL(r, r->AllReduce(counts, counts, (size_t)n, kNcclInt64, kNcclSum, e->comm, e->stream));
        return 0;
    }
    Stage st{e};
    RMR_TRY(st.init(Stage::pad((size_t)n * 8) + 1024));
    int64_t *dc = st.take<int64_t>(n);
    H2D(dc, counts, (size_t)n * 8);
    RMR_NCCL(r, r->AllReduce(dc, dc, (size_t)n, kNcclInt64, kNcclSum, e->comm, e->stream));
    D2H(counts, dc, (size_t)n * 8);
    RMR_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

int rmr_motif_flags(rmr_engine *e, const int8_t *int_seq, const int64_t *seq_off, int64_t n_reads,
                    const rmr_motif_set *motifs, uint8_t *flags, int mem) {
    if (!e || !int_seq || !seq_off || !motifs || !flags) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (n_reads < 0 || n_reads > (int64_t)1 << 30) RMR_FAIL(RMR_ERR_INVALID, "bad n_reads");
    if (motifs->n_motifs < 1 || motifs->n_motifs > 8) RMR_FAIL(RMR_ERR_INVALID, "1..8 motifs supported");
    for (int m = 0; m < motifs->n_motifs; ++m)
        if (motifs->len[m] < 1 || motifs->len[m] > 16 || motifs->focus_pos[m] >= motifs->len[m] ||
            motifs->focus_pos[m] < -64)
            RMR_FAIL(RMR_ERR_INVALID, "motif %d: length %d / focus %d unsupported", m, motifs->len[m], motifs->focus_pos[m]);
    if (n_reads == 0) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    int64_t total = 0;
    if (mem == RMR_MEM_HOST) total = seq_off[n_reads];
    else RMR_HIP(hipMemcpy(&total, seq_off + n_reads, 8, hipMemcpyDeviceToHost));
    if (total <= 0) return 0;
    if (mem == RMR_MEM_DEVICE) return launch_motif(e, int_seq, seq_off, (int)n_reads, total, *motifs, flags);
    Stage st{e};
    RMR_TRY(st.init(2 * Stage::pad((size_t)total) + Stage::pad((size_t)(n_reads + 1) * 8) + 4096));
    int8_t *ds = st.take<int8_t>(total);
    int64_t *d_off = st.take<int64_t>(n_reads + 1);
    uint8_t *df = st.take<uint8_t>(total);
    H2D(ds, int_seq, (size_t)total);
    H2D(d_off, seq_off, (size_t)(n_reads + 1) * 8);
    RMR_TRY(launch_motif(e, ds, d_off, (int)n_reads, total, *motifs, df));
    D2H(flags, df, (size_t)total);
    RMR_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

int rmr_vbz_decode(rmr_engine *e, const uint8_t *svb, const int64_t *row_off, const int32_t *row_samples,
                   int64_t n_rows, int16_t *out, int mem) {
    if (!e || !svb || !row_off || !row_samples || !out) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (n_rows < 0 || n_rows > (int64_t)1 << 30) RMR_FAIL(RMR_ERR_INVALID, "bad n_rows");
    if (n_rows == 0) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    std::vector<int64_t> ro((size_t)n_rows + 1), oo((size_t)n_rows + 1);
    std::vector<int32_t> rn((size_t)n_rows);
    if (mem == RMR_MEM_HOST) {
        memcpy(ro.data(), row_off, ro.size() * 8);
        memcpy(rn.data(), row_samples, rn.size() * 4);
    } else {
        RMR_HIP(hipMemcpy(ro.data(), row_off, ro.size() * 8, hipMemcpyDeviceToHost));
        RMR_HIP(hipMemcpy(rn.data(), row_samples, rn.size() * 4, hipMemcpyDeviceToHost));
    }
    oo[0] = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        if (rn[r] < 0 || ro[r + 1] < ro[r] || ro[r + 1] - ro[r] < ((int64_t)rn[r] + 7) / 8 + rn[r])
            RMR_FAIL(RMR_ERR_INVALID, "corrupt VBZ signal block (row %lld)", (long long)r);
        oo[r + 1] = oo[r] + rn[r];
    }
    const int64_t nbytes = ro[n_rows], nout = oo[n_rows];
    Stage st{e};
    RMR_TRY(st.init(Stage::pad((size_t)nbytes + 16) + 2 * Stage::pad((size_t)(n_rows + 1) * 8) + 2 * Stage::pad((size_t)n_rows * 4) +
                    Stage::pad((size_t)nout * 2 + 16) + 8192));
    int64_t *d_oo = st.take<int64_t>(n_rows + 1);
    int32_t *d_st = st.take<int32_t>(n_rows);
    H2D(d_oo, oo.data(), oo.size() * 8);
    RMR_HIP(hipMemsetAsync(d_st, 0, (size_t)n_rows * 4, e->stream));
    const uint8_t *d_svb = svb;
    const int64_t *d_ro = row_off;
    const int32_t *d_rn = row_samples;
    int16_t *d_out = out;
    if (mem == RMR_MEM_HOST) {
        uint8_t *b = st.take<uint8_t>(nbytes + 16);
        int64_t *o = st.take<int64_t>(n_rows + 1);
        int32_t *c = st.take<int32_t>(n_rows);
        d_out = st.take<int16_t>(nout + 8);
        H2D(b, svb, (size_t)nbytes);
        H2D(o, row_off, ro.size() * 8);
        H2D(c, row_samples, rn.size() * 4);
        d_svb = b; d_ro = o; d_rn = c;
    }
    RMR_TRY(launch_vbz(e, d_svb, d_ro, d_rn, d_oo, n_rows, d_out, d_st));
    std::vector<int32_t> hst((size_t)n_rows);
    D2H(hst.data(), d_st, (size_t)n_rows * 4);
    if (mem == RMR_MEM_HOST) D2H(out, d_out, (size_t)nout * 2);
    RMR_HIP(hipStreamSynchronize(e->stream));
    for (int64_t r = 0; r < n_rows; ++r)
        if (hst[r]) RMR_FAIL(RMR_ERR_INVALID, "corrupt VBZ signal block (row %lld)", (long long)r);
    return 0;
}

int rmr_forward(rmr_model *m, const float *sigs, const float *seqs, int64_t n, float *logits, int mem) {
    if (!m || !sigs || !seqs || !logits) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (n <= 0) return 0;
    rmr_engine *e = m->eng;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    if (mem == RMR_MEM_DEVICE) return run_pipeline(m, sigs, seqs, nullptr, 0, nullptr, 0, nullptr, 0, 0, n, logits);
    const size_t L = m->L, EC = 4 * (size_t)m->desc.kmer_len, no = m->desc.num_out;
    Stage st{e};
    RMR_TRY(st.init(Stage::pad(n * L * 4) + Stage::pad(n * EC * L * 4) + Stage::pad(n * no * 4) + 4096));
    float *ds = st.take<float>(n * L);
    float *dq = st.take<float>(n * EC * L);
    float *dl = st.take<float>(n * no);
    H2D(ds, sigs, n * L * 4);
    H2D(dq, seqs, n * EC * L * 4);
    RMR_TRY(run_pipeline(m, ds, dq, nullptr, 0, nullptr, 0, nullptr, 0, 0, n, dl));
    D2H(logits, dl, n * no * 4);
    RMR_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

// ---- one read, one call: staging + X1-X3 + the network, ONE stream synchronisation -------------------------------------
int rmr_call_read(rmr_model *m, const rmr_read *r, float *logits, int64_t *read_focus_bases) {
    if (!m || !r || !logits || !read_focus_bases) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (r->n_focus <= 0) return 0;
    if (!r->dacs || !r->seq_to_sig || !r->int_seq || !r->focus_bases) RMR_FAIL(RMR_ERR_INVALID, "NULL array in rmr_read");
    if (r->n_sig <= 0 || r->n_bases <= 0) RMR_FAIL(RMR_ERR_INVALID, "empty read");
    if (r->seq_itemsize != 1 && r->seq_itemsize != 2 && r->seq_itemsize != 4 && r->seq_itemsize != 8)
        RMR_FAIL(RMR_ERR_INVALID, "int_seq itemsize %d not in {1,2,4,8}", r->seq_itemsize);
    if (r->kb < 0 || r->ka < 0 || r->kb + r->ka + 1 != m->desc.kmer_len)
        RMR_FAIL(RMR_ERR_INVALID, "kmer context (%d,%d) does not match model kmer_len %d", r->kb, r->ka, m->desc.kmer_len);
    if (r->cc_before + r->cc_after != m->L)
        RMR_FAIL(RMR_ERR_INVALID, "chunk context (%d,%d) does not match model chunk_len %d", r->cc_before, r->cc_after, m->L);
    rmr_engine *e = m->eng;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    const int64_t ns = r->n_sig, nb = r->n_bases, nc = r->n_focus;
    const int no = m->desc.num_out, L = m->L;
    // one blob, host (pinned) and device images with the same offsets: everything the extraction kernels read; the
    // geometry rows behind it travel in a second, small copy
    size_t off = 0;
    auto seg = [&off](size_t bytes) { const size_t o = off; off += Stage::pad(bytes); return o; };
    const size_t o_dacs = seg(ns * 2 + 16), o_map = seg((nb + 1) * 8), o_seq = seg(nb + 16), o_foc = seg(nc * 8), o_off = seg(6 * 8),
                 o_sc = seg(2 * 8), o_cr = seg((nc + 1) * 4), blob_bytes = off, o_geo = seg(nc * 48), in_bytes = off;
    const size_t out_bytes = Stage::pad((size_t)nc * no * 4) + 256;
    RMR_TRY(e->ensure_pin_call(in_bytes + out_bytes));
    char *hp = reinterpret_cast<char *>(e->pin_call);
    // The staging buffer is pinned host memory the GPU can address: for ONE read the kernels fetch the read's arrays from it
    // across PCIe themselves and write the logits back into it (150 KB in, 2.5 KB out) instead of three queued copies, each
    // of which cost a launch on the host and a blit kernel + a dependency gap on the stream - a sixth of the call
    // (profiles/NOTES_r05.md section 1d).  RMR_CALL_READ_ZERO_COPY: bit 0 the read's arrays, bit 1 the chunk geometry,
    // bit 2 the logits; 0 = the copies.
    static const int zc = 7;
    char *hp_dev = nullptr;
    if (zc) RMR_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&hp_dev), hp, 0));
    memcpy(hp + o_dacs, r->dacs, (size_t)ns * 2);
    memcpy(hp + o_map, r->seq_to_sig, (size_t)(nb + 1) * 8);
    {
        int8_t *q = reinterpret_cast<int8_t *>(hp + o_seq);
        switch (r->seq_itemsize) {
        case 1: memcpy(q, r->int_seq, (size_t)nb); break;
        case 2: { const int16_t *s = reinterpret_cast<const int16_t *>(r->int_seq); for (int64_t i = 0; i < nb; ++i) q[i] = (int8_t)s[i]; } break;
        case 4: { const int32_t *s = reinterpret_cast<const int32_t *>(r->int_seq); for (int64_t i = 0; i < nb; ++i) q[i] = (int8_t)s[i]; } break;
        default: { const int64_t *s = reinterpret_cast<const int64_t *>(r->int_seq); for (int64_t i = 0; i < nb; ++i) q[i] = (int8_t)s[i]; } break;
        }
    }
    memcpy(hp + o_foc, r->focus_bases, (size_t)nc * 8);
    int64_t *ho = reinterpret_cast<int64_t *>(hp + o_off);
    ho[0] = 0; ho[1] = ns; ho[2] = 0; ho[3] = nb; ho[4] = 0; ho[5] = nc;
    double *hs = reinterpret_cast<double *>(hp + o_sc);
    hs[0] = r->shift; hs[1] = r->scale;
    memset(hp + o_cr, 0, (size_t)(nc + 1) * 4);  // every chunk belongs to read 0
    // The arena is sized before the widths of the chunk rows are known, for chunks of up to `cap` bases (a chunk of L samples
    // holds more only where bases have no samples of their own: then it grows to the exact number below, before anything that
    // depends on it is queued).
    Stage st{e};
    const int64_t cap = std::min<int64_t>(nb + 1, 2 * (int64_t)L + 8);
    auto arena_bytes = [&](int64_t msl_) {
        return in_bytes + Stage::pad(ns * 4 + 16) + Stage::pad((size_t)nc * L * 4) + Stage::pad((size_t)nc * (msl_ + r->kb + r->ka + 1)) +
               Stage::pad((size_t)nc * (msl_ + 2) * 2) + Stage::pad(nc * 2) + Stage::pad(nc * 8) + Stage::pad((size_t)nc * no * 4) + 8192;
    };
    // the read's arrays first and on their way ...
    RMR_TRY(st.init(arena_bytes(cap)));
    char *dp = st.take<char>(in_bytes);
    char *arena_in = dp;
    if (zc & 1) dp = hp_dev;
    else RMR_HIP(hipMemcpyAsync(dp, hp, blob_bytes, hipMemcpyHostToDevice, e->stream));
    rmr_reads d{};
    d.n_reads = 1;
    d.dacs = reinterpret_cast<const int16_t *>(dp + o_dacs);
    d.seq_to_sig = reinterpret_cast<const int64_t *>(dp + o_map);
    d.int_seq = reinterpret_cast<const int8_t *>(dp + o_seq);
    d.focus_bases = reinterpret_cast<const int64_t *>(dp + o_foc);
    d.sig_off = reinterpret_cast<const int64_t *>(dp + o_off);
    d.seq_off = d.sig_off + 2;
    d.focus_off = d.sig_off + 4;
    d.shift = reinterpret_cast<const double *>(dp + o_sc);
    d.scale = d.shift + 1;
    d.cc_before = r->cc_before; d.cc_after = r->cc_after; d.kb = r->kb; d.ka = r->ka;
    d.base_start_justify = r->base_start_justify; d.offset = r->offset;
    const int32_t *chunk_read = reinterpret_cast<const int32_t *>(dp + o_cr);
    const int64_t *dgeo = reinterpret_cast<const int64_t *>(((zc & 2) ? hp_dev : arena_in) + o_geo);
    float *dsig = st.take<float>(ns + 4);
    RMR_TRY(launch_geometry(e, d, 0, chunk_read, dsig, ns, nullptr, nullptr, nullptr));  // n_chunks 0: the signal normalisation alone
    // ... then, while they cross PCIe and the signal is normalised, the geometry of the chunks on the host: integer
    // arithmetic on the mapping - the function the geometry kernel runs (rmr_geometry.h), its searches started at the focus
    // base when the mapping is monotone.  The widths of the chunk rows are then known without asking the GPU: the whole call
    // is queued behind one another and waited for once.
    const int64_t *map = reinterpret_cast<const int64_t *>(hp + o_map);
    bool monotone = true;
    for (int64_t i = 0; i < nb; ++i) monotone &= map[i + 1] >= map[i];
    int64_t *hgeo = reinterpret_cast<int64_t *>(hp + o_geo);
    int64_t msl = 0;
    for (int64_t i = 0; i < nc; ++i) {
        const int64_t sl = chunk_geometry_row(map, nb, ns, r->focus_bases[i], r->base_start_justify, r->offset, r->cc_before, r->cc_after,
                                              hgeo + i * 6, monotone);
        msl = sl > msl ? sl : msl;
        read_focus_bases[i] = hgeo[i * 6 + 3];
    }
    if (msl > nb + 1 || msl > 32000) RMR_FAIL(RMR_ERR_INVALID, "chunk of %lld bases", (long long)msl);
    if (msl > cap) {  // zero-dwell bases made a chunk wider than the arena was sized for: start over with the exact size
        RMR_HIP(hipStreamSynchronize(e->stream));
        st = Stage{e};
        RMR_TRY(st.init(arena_bytes(msl)));
        dp = st.take<char>(in_bytes);
        arena_in = dp;
        if (zc & 1) dp = hp_dev;
        else RMR_HIP(hipMemcpyAsync(dp, hp, blob_bytes, hipMemcpyHostToDevice, e->stream));
        d.dacs = reinterpret_cast<const int16_t *>(dp + o_dacs);
        d.seq_to_sig = reinterpret_cast<const int64_t *>(dp + o_map);
        d.int_seq = reinterpret_cast<const int8_t *>(dp + o_seq);
        d.focus_bases = reinterpret_cast<const int64_t *>(dp + o_foc);
        d.sig_off = reinterpret_cast<const int64_t *>(dp + o_off);
        d.seq_off = d.sig_off + 2;
        d.focus_off = d.sig_off + 4;
        d.shift = reinterpret_cast<const double *>(dp + o_sc);
        d.scale = d.shift + 1;
        chunk_read = reinterpret_cast<const int32_t *>(dp + o_cr);
        dgeo = reinterpret_cast<const int64_t *>(((zc & 2) ? hp_dev : arena_in) + o_geo);
        dsig = st.take<float>(ns + 4);
        RMR_TRY(launch_geometry(e, d, 0, chunk_read, dsig, ns, nullptr, nullptr, nullptr));
    }
    if (!(zc & 2)) RMR_HIP(hipMemcpyAsync(arena_in + o_geo, hp + o_geo, (size_t)nc * 48, hipMemcpyHostToDevice, e->stream));
    const int seq_w = (int)std::max<int64_t>(msl + r->kb + r->ka, r->kb + r->ka + 1), map_w = (int)std::max<int64_t>(msl + 1, 2);
    float *dsignal = st.take<float>((size_t)nc * L);
    int8_t *dseqs = st.take<int8_t>((size_t)nc * seq_w);
    int16_t *dmaps = st.take<int16_t>((size_t)nc * map_w);
    int16_t *dlens = st.take<int16_t>(nc);
    int64_t *drfb = st.take<int64_t>(nc);
    float *dlog = st.take<float>((size_t)nc * no);
    float *hlog = reinterpret_cast<float *>(hp + in_bytes);
    if (zc & 4) dlog = reinterpret_cast<float *>(hp_dev + in_bytes);
    RMR_TRY(launch_fill(e, d, nc, chunk_read, dsig, dgeo, dsignal, dseqs, seq_w, dmaps, map_w, dlens, drfb));
    RMR_TRY(run_pipeline(m, dsignal, nullptr, dseqs, seq_w, dmaps, map_w, dlens, r->kb, r->ka, nc, dlog));
    if (!(zc & 4)) RMR_HIP(hipMemcpyAsync(hlog, dlog, (size_t)nc * no * 4, hipMemcpyDeviceToHost, e->stream));
    RMR_HIP(hipStreamSynchronize(e->stream));
    memcpy(logits, hlog, (size_t)nc * no * 4);
    return 0;
}

int rmr_infer_chunks(rmr_model *m, const float *signal, const int8_t *seqs, int seq_w,
                     const int16_t *maps, int map_w, const int16_t *lens, int kb, int ka, int64_t n,
                     float *logits, int64_t *label_counts, int mem) {
    if (!m || !signal || !seqs || !maps || !lens || !logits) RMR_FAIL(RMR_ERR_INVALID, "NULL argument");
    if (kb < 0 || ka < 0 || kb + ka + 1 != m->desc.kmer_len)
        RMR_FAIL(RMR_ERR_INVALID, "kmer context (%d,%d) does not match model kmer_len %d", kb, ka, m->desc.kmer_len);
    if (seq_w < kb + ka + 1 || map_w < 2) RMR_FAIL(RMR_ERR_INVALID, "bad array widths");
    if (n <= 0) return 0;
    rmr_engine *e = m->eng;
    std::lock_guard<std::mutex> lk(e->mu);
    RMR_HIP(hipSetDevice(e->device));
    const int no = m->desc.num_out;
    if (mem == RMR_MEM_DEVICE) {
        RMR_TRY(run_pipeline(m, signal, nullptr, seqs, seq_w, maps, map_w, lens, kb, ka, n, logits));
        if (label_counts) RMR_TRY(launch_count(e, logits, n, no, label_counts));
        return 0;
    }
    const size_t L = m->L;
    Stage st{e};
    RMR_TRY(st.init(Stage::pad(n * L * 4) + Stage::pad((size_t)n * seq_w) + Stage::pad((size_t)n * map_w * 2) +
                    Stage::pad(n * 2) + Stage::pad((size_t)n * no * 4) + 8192));
    float *dsig = st.take<float>(n * L);
    int8_t *ds = st.take<int8_t>((size_t)n * seq_w);
    int16_t *dm = st.take<int16_t>((size_t)n * map_w);
    int16_t *dl = st.take<int16_t>(n);
    float *dlog = st.take<float>((size_t)n * no);
    int64_t *dc = st.take<int64_t>(16);
    const int64_t hsb = tune_int("RMR_HOST_SUBBATCH", 131072);
    if (hsb > 0 && n > hsb) {
        // pipelined upload: the CPU copies sub-batch i+1 into a pinned slot and the aux stream uploads it
        // while the kernels of sub-batch i run on the main stream
        const size_t o_seq = Stage::pad((size_t)hsb * L * 4), o_map = o_seq + Stage::pad((size_t)hsb * seq_w);
        const size_t o_len = o_map + Stage::pad((size_t)hsb * map_w * 2), slot_b = o_len + Stage::pad((size_t)hsb * 2);
        RMR_TRY(e->ensure_pinned(2 * slot_b));
        const int nthr = (int)4;
        int64_t idx = 0;
        for (int64_t c0 = 0; c0 < n; c0 += hsb, ++idx) {
            const int64_t nb = (n - c0) < hsb ? (n - c0) : hsb;
            const int slot = (int)(idx & 1);
            char *pb = reinterpret_cast<char *>(e->pinned) + (size_t)slot * slot_b;
            if (idx >= 2) RMR_HIP(hipEventSynchronize(e->ev_h2d[slot]));  // the upload that used this slot is done
            {
                const char *src = reinterpret_cast<const char *>(signal + (size_t)c0 * L);
                const size_t bytes = (size_t)nb * L * 4, part = (bytes / nthr + 4095) & ~(size_t)4095;
                std::vector<std::thread> pool;
                for (int t = 1; t < nthr; ++t) {
                    const size_t b0 = (size_t)t * part;
                    if (b0 < bytes) pool.emplace_back([=] { memcpy(pb + b0, src + b0, std::min(part, bytes - b0)); });
                }
                memcpy(pb, src, std::min(part, bytes));
                memcpy(pb + o_seq, seqs + (size_t)c0 * seq_w, (size_t)nb * seq_w);
                memcpy(pb + o_map, maps + (size_t)c0 * map_w, (size_t)nb * map_w * 2);
                memcpy(pb + o_len, lens + c0, (size_t)nb * 2);
                for (auto &th : pool) th.join();
            }
            RMR_HIP(hipMemcpyAsync(dsig + (size_t)c0 * L, pb, (size_t)nb * L * 4, hipMemcpyHostToDevice, e->aux));
            RMR_HIP(hipMemcpyAsync(ds + (size_t)c0 * seq_w, pb + o_seq, (size_t)nb * seq_w, hipMemcpyHostToDevice, e->aux));
            RMR_HIP(hipMemcpyAsync(dm + (size_t)c0 * map_w, pb + o_map, (size_t)nb * map_w * 2, hipMemcpyHostToDevice, e->aux));
            RMR_HIP(hipMemcpyAsync(dl + c0, pb + o_len, (size_t)nb * 2, hipMemcpyHostToDevice, e->aux));
            RMR_HIP(hipEventRecord(e->ev_h2d[slot], e->aux));
            RMR_HIP(hipStreamWaitEvent(e->stream, e->ev_h2d[slot], 0));
            RMR_TRY(run_pipeline(m, dsig + (size_t)c0 * L, nullptr, ds + (size_t)c0 * seq_w, seq_w, dm + (size_t)c0 * map_w,
                                 map_w, dl + c0, kb, ka, nb, dlog + (size_t)c0 * no));
        }
    } else {
        H2D(dsig, signal, n * L * 4);
        H2D(ds, seqs, (size_t)n * seq_w);
        H2D(dm, maps, (size_t)n * map_w * 2);
        H2D(dl, lens, (size_t)n * 2);
        RMR_TRY(run_pipeline(m, dsig, nullptr, ds, seq_w, dm, map_w, dl, kb, ka, n, dlog));
    }
    if (label_counts) {
        H2D(dc, label_counts, (size_t)no * 8);
        RMR_TRY(launch_count(e, dlog, n, no, dc));
        D2H(label_counts, dc, (size_t)no * 8);
    }
    D2H(logits, dlog, (size_t)n * no * 4);
    RMR_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

}  // extern "C"
