"""Model loading for the hot path — drop-in for the load side of `remora.model_util`
(src/remora/model_util.py): load_model :566-699, load_torchscript_model :532-563,
_raw_load_torchscript_model :468-481, add_derived_metadata :341-448.

The model file format is unchanged: a TorchScript archive whose extra file `meta.txt` is a
JSON dict (:115-176).  torch.jit.load is used only to read the file; the weights go to the
HIP engine (BatchNorm folding and MFMA packing happen inside rmr_model_create) and what is
returned in place of the ScriptModule is a `HipModel`.
"""
import json
import os
from os.path import isfile

import numpy as np

from . import RemoraError
from .engine import HipModel, _torch
from .refine_signal_map import SigMapRefiner


def add_derived_metadata(md):
    """Expand the raw `meta.txt` dict the way the reference does (:341-448)."""
    md.setdefault("reverse_signal", False)
    md.setdefault("pa_scaling", None)
    if md["mod_bases"] == "None":
        md["mod_bases"] = None
        md["mod_long_names"] = None
    else:
        md["mod_long_names"] = [md[f"mod_long_names_{i}"] for i in range(len(md["mod_bases"]))]
    if "kmer_context_bases" not in md:
        md["kmer_context_bases"] = (int(md["kmer_context_bases_0"]), int(md["kmer_context_bases_1"]))
    md["kmer_len"] = sum(md["kmer_context_bases"]) + 1
    if "chunk_context" not in md:
        md["chunk_context"] = (int(md["chunk_context_0"]), int(md["chunk_context_1"]))
    md["chunk_len"] = sum(md["chunk_context"])
    if "num_motifs" not in md:
        md["motifs"] = [(md["motif"], int(md["motif_offset"]))]
        md["motif_offset"] = int(md["motif_offset"])
    else:
        md["motifs"] = [(md[f"motif_{i}"], int(md[f"motif_offset_{i}"])) for i in range(int(md["num_motifs"]))]
    md["can_base"] = md["motifs"][0][0][md["motifs"][0][1]]
    md["motif"] = md["motifs"][0] if len(md["motifs"]) == 1 else (md["can_base"], 0)
    if md["mod_bases"] is not None:
        mod_str = "; ".join(f"{b}={n}" for b, n in zip(md["mod_bases"], md["mod_long_names"]))
        md["alphabet_str"] = f"loaded modified base model to call (alt to {md['can_base']}): {mod_str}"
    if md.get("refine_kmer_levels") is not None:
        levels = np.frombuffer(md["refine_kmer_levels"].encode("cp437"), dtype=np.float32)
        sd_arr = np.frombuffer(md["refine_sd_arr"].encode("cp437"), dtype=np.float32)
        md["sig_map_refiner"] = SigMapRefiner(
            _levels_array=levels, center_idx=int(md["refine_kmer_center_idx"]),
            do_rough_rescale=md["refine_do_rough_rescale"], scale_iters=int(md["refine_scale_iters"]),
            algo=md["refine_algo"], half_bandwidth=int(md["refine_half_bandwidth"]), sd_arr=sd_arr)
    else:
        md["sig_map_refiner"] = SigMapRefiner()
        md["base_start_justify"] = False
        md["offset"] = 0
    for k in [k for k in md if k.startswith("refine_")]:
        del md[k]


def _raw_load_torchscript_model(model_filename, device=None):
    torch = _torch()
    extra = {"meta.txt": ""}
    script = torch.jit.load(model_filename, _extra_files=extra, map_location="cpu")
    md = json.loads(extra["meta.txt"])
    state = {k: v.detach().cpu().numpy() for k, v in script.state_dict().items()
             if not k.endswith("num_batches_tracked")}
    return state, md


def load_torchscript_model(model_filename, device=None, quiet=False, eval_only=False, dtype=None):
    state, md = _raw_load_torchscript_model(model_filename, device)
    add_derived_metadata(md)
    # arithmetic of the GEMM stages: "fp32" (default, exact fp32 MFMA) or a bf16-MFMA mode
    # ("bf16x6" fp32-class split, "bf16x3", "bf16"); REMORA_HIP_DTYPE overrides the default
    dtype = dtype or os.environ.get("REMORA_HIP_DTYPE", "fp32")
    model = HipModel(state, md["chunk_len"], device=device, dtype=dtype)
    if model.kmer_len != md["kmer_len"]:
        raise RemoraError(f"model weights expect kmer_len {model.kmer_len}, metadata says {md['kmer_len']}")
    return model.eval(), md


def load_model(model_filename=None, *, pore=None, basecall_model_type=None, basecall_model_version=None,
               modified_bases=None, remora_model_type=None, remora_model_version=None, device=None,
               quiet=True, eval_only=False):
    """Same signature as remora.model_util.load_model (:566-578).  Returns
    (HipModel, model_metadata).  Pretrained-model lookup by pore/basecaller needs a network
    download in the reference (:684-695); here it raises RemoraError unless a file is given."""
    if model_filename is not None:
        if not isfile(model_filename):
            raise RemoraError(f"Remora model file ({model_filename}) not found.")
        try:
            return load_torchscript_model(model_filename, device, quiet=quiet, eval_only=eval_only)
        except (AttributeError, RuntimeError, KeyError, json.JSONDecodeError):
            raise RemoraError("Failed loading torchscript model.")
    if pore is None:
        raise RemoraError("Must specify a pore.")
    raise RemoraError(
        f"load_model(pore={pore!r}, basecall_model_type={basecall_model_type!r}, ..., modified_bases={modified_bases!r}): the "
        "reference resolves these through its pretrained-model registry and downloads the file "
        "(remora.model_util.load_model, src/remora/model_util.py:592-699 -> get_pretrained_models / ModelDownload); that "
        "registry is outside this engine's scope - pass model_filename= (the TorchScript .pt with meta.txt that "
        "`remora model download` or the reference's load_model caches under its models directory)")


def model_from_state(state, model_metadata, device=None, engine=None, dtype="fp32"):
    """HipModel straight from a state_dict-like {name: array} (numpy or torch) — the entry point
    for callers that already hold the weights."""
    st = {}
    for k, v in state.items():
        if k.endswith("num_batches_tracked"):
            continue
        st[k] = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
    return HipModel(st, int(sum(model_metadata["chunk_context"])), device=device, engine=engine, dtype=dtype)
