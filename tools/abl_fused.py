"""Stage cost of the fused bf16 front kernel by timing ablations (experiment build: make -C remora_amd/csrc abl;
run with REMORA_HIP_LIB=remora_amd/libremora_hip_abl.so).  Prints the HIP-event time of fused_front per launch for
each RMR_FUSED_ABLATE mask; the logits of an ablated run are garbage by construction."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from remora_amd import synth
    from remora_amd.engine import get_engine
    from remora_amd.model_util import model_from_state

    cfg = sys.argv[1] if len(sys.argv) > 1 else "C100"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
    masks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 4, 8, 16, 32, 64, 6, 56, 63, 62, 57, 55, 47, 31, 95]
    cc, kcb, msl, num_out, _ = synth.CONFIGS[cfg]
    state = synth.synth_state("conv_lstm", 64, 9, num_out, seed=0)
    model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=kcb), device=0, dtype="bf16")
    d = synth.synth_chunks_config(cfg, n)
    dev = [torch.from_numpy(d[k]).cuda() for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
    eng = get_engine(0)
    names = {0: "full", 1: "-S0 loads", 2: "-S1a sig1/pidx/code", 4: "-S1b one-hot", 8: "-S2 sig2/seq1", 16: "-S3 sig3/seq2",
             32: "-S4 merge1", 64: "-swish", 6: "-S1a-S1b", 56: "-S2-S3-S4 (no MFMA)", 63: "nothing (barriers + loop)",
             62: "only S0", 57: "only S1a+S1b", 55: "only S2", 47: "only S3", 31: "only S4", 31 + 64: "only S4, no swish"}
    base = None
    for mask in masks:
        os.environ["RMR_FUSED_ABLATE"] = str(mask)
        for _ in range(2):
            model.infer_chunks(*dev, kcb)
        eng.profile_reset()
        eng.profile_enable(True)
        for _ in range(3):
            model.infer_chunks(*dev, kcb)
        eng.profile_enable(False)
        ms, launches = eng.profile()["fused_front"]
        per_chunk_ns = ms * 1e6 / (3 * n)
        base = base or per_chunk_ns
        print(f"mask {mask:3d}  {names.get(mask, ''):28s} {per_chunk_ns:7.3f} ns/chunk  ({per_chunk_ns - base:+.3f})", flush=True)


if __name__ == "__main__":
    main()
