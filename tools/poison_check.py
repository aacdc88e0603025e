"""Logits hashes of every pipeline with and without RMR_POISON=1 (LDS and vector registers of every CU filled with NaN
patterns in front of every kernel launch: rmr_internal.h).  A pipeline whose hash moves reads a word it never wrote.

    python tools/poison_check.py            # runs itself twice and compares"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("conv_lstm", "C100", "fp32"), ("conv_lstm", "C100", "bf16"), ("conv_lstm", "C100", "f16"), ("conv_lstm", "C100", "bf16x6"),
         ("conv_lstm", "C100", "bf16x3"), ("conv_lstm", "C200", "fp32"), ("conv_lstm", "C200", "bf16"), ("conv_only", "C100", "fp32")]


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch

    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    out = {}
    for arch, cfg, dtype in CASES:
        cc, kcb, _, num_out, _ = synth.CONFIGS[cfg]
        state = synth.synth_state(arch, 64, 9, num_out, seed=0)
        model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=kcb), device=0, dtype=dtype)
        for n in (20000, 37):
            d = synth.synth_chunks_config(cfg, n, shard=7)
            dev = [torch.from_numpy(d[k]).cuda() for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
            hs = set()
            for _ in range(3):
                lg = model.infer_chunks(*dev, kcb).cpu().numpy()
                hs.add(hashlib.sha256(lg.tobytes()).hexdigest()[:12] + ("!nan" if np.isnan(lg).any() else ""))
            # host-buffer path too (staging kernels / copies)
            lg = model.infer_chunks(d["signal"], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"], kcb)
            hs.add(hashlib.sha256(np.asarray(lg).tobytes()).hexdigest()[:12] + ("!nan" if np.isnan(lg).any() else ""))
            out[f"{arch}/{cfg}/{dtype}/n{n}"] = sorted(hs)
    print("RESULT " + json.dumps(out))


def main():
    if "--child" in sys.argv:
        return child()
    res = {}
    for poison in ("0", "1"):
        env = dict(os.environ, RMR_POISON=poison)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=900)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        if not line:
            print(f"RMR_POISON={poison}: FAILED rc={p.returncode}\n{p.stderr[-2000:]}")
            return 1
        res[poison] = json.loads(line[-1][7:])
    bad = 0
    for k in res["0"]:
        same = res["0"][k] == res["1"][k] and len(res["0"][k]) == 1
        bad += not same
        print(f"{'ok  ' if same else 'DIFF'} {k}: clean {res['0'][k]} poisoned {res['1'][k]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
