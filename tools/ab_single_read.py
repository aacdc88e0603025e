"""A/B of the single-read call (inference.call_read_mods -> rmr_call_read) for one setting of the environment: median and best
of 5 passes over 256 synthetic 5 kb reads, fp32 and bf16 models, and a checksum of the logits (the settings must agree bit for
bit).  Run once per setting:  RMR_CALL_READ_ZERO_COPY=0 python tools/ab_single_read.py"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from remora_amd import synth
from remora_amd.data_chunks import RemoraRead
from remora_amd.inference import call_read_mods
from remora_amd.model_util import model_from_state

md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"],
          can_base="C", base_start_justify=False, offset=0, sig_map_refiner=None, reverse_signal=False, pa_scaling=None)
rs = []
for i in range(256):
    r = synth.synth_read(5000, idx=i)
    rs.append(RemoraRead(dacs=r["dacs"], shift=r["shift"], scale=r["scale"], seq_to_sig_map=r["seq_to_sig_map"], int_seq=r["int_seq"]))
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("RMR_")) or "(defaults)"
for dtype in ("fp32", "bf16"):
    model = model_from_state(synth.synth_state(), md, device=0, dtype=dtype) if dtype != "fp32" else model_from_state(synth.synth_state(), md, device=0)
    h = hashlib.sha256()
    for r in rs[:16]:
        probs, _, pos = call_read_mods(r, model, md, return_mod_probs=True)
        h.update(np.ascontiguousarray(probs).tobytes())
        h.update(np.ascontiguousarray(pos).tobytes())
    torch.cuda.synchronize()
    per = []
    for _ in range(5):
        t = time.perf_counter()
        for r in rs:
            call_read_mods(r, model, md)
        per.append((time.perf_counter() - t) / len(rs) * 1e6)
    print(f"{tag:40s} {dtype}: median {np.median(per):6.1f} us  best {min(per):6.1f} us per read = {1e6 / np.median(per) / 1e3:.2f} k reads/s   sha {h.hexdigest()[:12]}", flush=True)
