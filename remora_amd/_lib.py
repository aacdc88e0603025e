"""ctypes binding of libremora_hip.so (include/remora_hip.h).  Loading fails loudly: the
product path has no fallback."""
import ctypes
import os
import threading

from . import RemoraError

_HERE = os.path.dirname(os.path.abspath(__file__))
# REMORA_HIP_LIB: another build of the same library (experiment builds such as `make abl`); never a fallback
LIB_PATH = os.environ.get("REMORA_HIP_LIB") or os.path.join(_HERE, "libremora_hip.so")

MEM_HOST, MEM_DEVICE = 0, 1
ARCH_CONV_LSTM, ARCH_CONV_ONLY = 0, 1
ENGINE_OWN_STREAM, ENGINE_USE_STREAM = 0, 1
ERR_DISCORDANT_SEQ, ERR_DISCORDANT_SIG = -4, -5

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_vp = ctypes.c_void_p


class ModelDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("arch", "size", "kmer_len", "num_out", "chunk_len", "dtype")]


class BamBatch(ctypes.Structure):
    """rmr_bam_batch of include/remora_hip.h."""

    _fields_ = [("n_records", c_i64)] + [(n, c_vp) for n in (
        "flag", "ref_id", "pos", "mapq", "l_seq", "n_cigar", "raw_off", "raw", "name_off", "names", "seq_off", "seq",
        "cigar_off", "cigar", "tags_off", "has", "mv_off", "mv", "ts", "ns", "sp", "sm", "sd", "pi_off", "pi", "md_off",
        "md", "ref_ok", "refseq_off", "refseq", "voffset")]


class MotifSet(ctypes.Structure):
    _fields_ = [("n_motifs", ctypes.c_int32), ("len", ctypes.c_int32 * 8), ("focus_pos", ctypes.c_int32 * 8),
                ("mask", (ctypes.c_uint8 * 16) * 8)]


class Read(ctypes.Structure):
    """rmr_read (include/remora_hip.h): one read for rmr_call_read, host pointers."""
    _fields_ = [("dacs", c_vp), ("n_sig", c_i64), ("seq_to_sig", c_vp), ("int_seq", c_vp), ("seq_itemsize", ctypes.c_int32),
                ("_pad", ctypes.c_int32), ("n_bases", c_i64), ("shift", ctypes.c_double), ("scale", ctypes.c_double),
                ("focus_bases", c_vp), ("n_focus", c_i64), ("cc_before", ctypes.c_int32), ("cc_after", ctypes.c_int32),
                ("kb", ctypes.c_int32), ("ka", ctypes.c_int32), ("base_start_justify", ctypes.c_int32), ("offset", ctypes.c_int32)]


class Reads(ctypes.Structure):
    _fields_ = [
        ("n_reads", c_i64), ("dacs", c_vp), ("sig_off", c_vp), ("seq_to_sig", c_vp),
        ("int_seq", c_vp), ("seq_off", c_vp), ("shift", c_vp), ("scale", c_vp),
        ("focus_bases", c_vp), ("focus_off", c_vp),
        ("cc_before", ctypes.c_int32), ("cc_after", ctypes.c_int32), ("kb", ctypes.c_int32),
        ("ka", ctypes.c_int32), ("base_start_justify", ctypes.c_int32), ("offset", ctypes.c_int32),
        ("host_sig_off", c_vp), ("host_seq_off", c_vp), ("host_focus_off", c_vp),  # optional, see include/remora_hip.h
    ]


# name -> (restype, argtypes); every symbol include/remora_hip.h declares
SIGNATURES = {
    "rmr_last_error": (ctypes.c_char_p, []),
    "rmr_version": (ctypes.c_char_p, []),
    "rmr_engine_create": (c_int, [c_int, c_vp, c_int, ctypes.POINTER(c_vp)]),
    "rmr_engine_destroy": (None, [c_vp]),
    "rmr_engine_synchronize": (c_int, [c_vp]),
    "rmr_engine_wait_for": (c_int, [c_vp, c_vp]),
    "rmr_engine_set_subbatch": (c_int, [c_vp, c_i64]),
    "rmr_model_create": (c_int, [c_vp, ctypes.POINTER(ModelDesc), c_vp, ctypes.c_size_t, ctypes.POINTER(c_vp)]),
    "rmr_model_destroy": (None, [c_vp]),
    "rmr_model_weight_count": (ctypes.c_size_t, [ctypes.POINTER(ModelDesc)]),
    "rmr_model_padded_size": (c_int, [ctypes.POINTER(ModelDesc)]),
    "rmr_model_pad_weights": (c_int, [ctypes.POINTER(ModelDesc), c_vp, ctypes.c_size_t, ctypes.POINTER(ModelDesc), c_vp, ctypes.c_size_t,
                                      ctypes.POINTER(ctypes.c_size_t)]),
    "rmr_encode_kmers": (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_i64, c_int, c_vp, c_int]),
    "rmr_trim_chunk_context": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_i64, c_int]),
    "rmr_parse_moves": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_vp, ctypes.POINTER(c_i64), c_int]),
    "rmr_parse_moves_batch": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int]),
    "rmr_assemble_reads": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "rmr_bam_open": (c_int, [ctypes.c_char_p, ctypes.POINTER(c_vp)]),
    "rmr_bam_open_threads": (c_int, [ctypes.c_char_p, c_int, ctypes.POINTER(c_vp)]),
    "rmr_bam_close": (None, [c_vp]),
    "rmr_bam_header": (c_int, [c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "rmr_bam_ref_name": (ctypes.c_char_p, [c_vp, c_i64]),
    "rmr_bam_read_batch": (c_int, [c_vp, c_i64, c_int, c_vp]),
    "rmr_bam_seek": (c_int, [c_vp, c_i64]),
    "rmr_bam_scan": (c_int, [c_vp, c_i64, c_vp, c_i64, ctypes.POINTER(c_i64)]),
    "rmr_bam_guess_start": (c_int, [c_vp, c_i64, ctypes.POINTER(c_i64)]),
    "rmr_format_mm_ml": (c_int, [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, ctypes.c_char_p, ctypes.c_char, ctypes.c_char, c_vp, c_i64,
                                 c_vp, c_vp, c_i64, c_vp]),
    "rmr_records_with_mod_tags": (c_int, [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.POINTER(c_i64)]),
    "rmr_records_with_mod_tags_ref": (c_int, [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "rmr_inflate_raw": (c_int, [c_vp, c_i64, c_vp, c_i64]),
    "rmr_bgzf_huffman": (c_int, [c_vp, c_i64, c_int, c_vp, c_i64, ctypes.POINTER(c_i64)]),
    "rmr_zstd_frame_sizes": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "rmr_zstd_rows": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_int]),
    "rmr_vbz_decode": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_int]),
    "rmr_motif_flags": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_int]),
    "rmr_motif_focus_counts": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "rmr_motif_focus_fill": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "rmr_ref_to_signal": (c_int, [c_vp, c_i64, c_int, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "rmr_ref_anchor_batch": (c_int, [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int]),
    "rmr_pack_reads": (c_int, [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int]),
    "rmr_signal_histograms": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "rmr_pack_reads_narrow": (c_int, [c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "rmr_orient_bases": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int]),
    "rmr_chunk_geometry": (c_int, [c_vp, ctypes.POINTER(Reads), c_vp, c_vp, ctypes.POINTER(c_i64), c_int]),
    "rmr_chunk_fill": (c_int, [c_vp, ctypes.POINTER(Reads), c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int]),
    "rmr_forward": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, c_int]),
    "rmr_infer_chunks": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, c_i64, c_vp, c_vp, c_int]),
    "rmr_call_read": (c_int, [c_vp, c_vp, c_vp, c_vp]),
    "rmr_count_labels": (c_int, [c_vp, c_vp, c_i64, c_int, c_vp, c_int]),
    "rmr_validation_tally": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "rmr_comm_unique_id": (c_int, [c_vp]),
    "rmr_comm_init": (c_int, [c_vp, c_vp, c_int, c_int]),
    "rmr_comm_destroy": (c_int, [c_vp]),
    "rmr_allreduce_counts": (c_int, [c_vp, c_vp, c_int, c_int]),
    "rmr_refiner_create": (c_int, [c_vp, c_vp, ctypes.POINTER(c_vp)]),
    "rmr_refiner_destroy": (None, [c_vp]),
    "rmr_refine_status_message": (ctypes.c_char_p, [c_int]),
    "rmr_refine_signal_maps": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int]),
    "rmr_rescale_quantiles": (c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp,
                                      c_vp, c_vp]),
    "rmr_profile_enable": (c_int, [c_vp, c_int]),
    "rmr_profile_reset": (c_int, [c_vp]),
    "rmr_profile_num_kernels": (c_int, []),
    "rmr_profile_kernel_name": (ctypes.c_char_p, [c_int]),
    "rmr_profile_get": (c_int, [c_vp, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)]),
}

_lib = None
_lock = threading.Lock()


def lib():
    """The loaded library; raises RemoraError if it was not built (python __graft_entry__.py
    or `make -C remora_amd/csrc`)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RemoraError(
                    f"{LIB_PATH} not found: build the HIP extension first "
                    "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback"
                )
            # PyTorch-ROCm ships its own HIP runtime: it has to be in the process before this library pulls in
            # /opt/rocm's, otherwise the second runtime to initialise finds "no ROCm-capable device"
            import torch  # noqa: F401

            try:
                L = ctypes.CDLL(LIB_PATH)
            except OSError as e:
                raise RemoraError(f"cannot load {LIB_PATH}: {e}")
            for name, (res, args) in SIGNATURES.items():
                try:
                    fn = getattr(L, name)
                except AttributeError:
                    raise RemoraError(f"{LIB_PATH} does not export {name}")
                fn.restype = res
                fn.argtypes = args
            _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().rmr_last_error()
        raise RemoraError((msg or b"unknown error").decode(errors="replace"))
