"""Diagnostic (GPU): error statistics of the 16-bit pipelines against the float64 network and against the float64 network rounded at
the pipeline's 16-bit sites (oracle/lowp_emulation.py) - and the distance between the kernel's result and the emulation's."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import lowp_emulation, torch_ref, oracle as O  # noqa: E402
from remora_amd import synth  # noqa: E402
from remora_amd.model_util import model_from_state  # noqa: E402

SITES = ("wconv.sig3", "wconv.seq2", "wconv.merge1", "aconv.sig2", "aconv.seq1", "aconv.cat", "x", "wlstm", "h")
for cfg, size in (("C100", 128), ("C100", 96), ("C100", 64)):
    cc, kcb, _, num_out, _ = synth.CONFIGS[cfg]
    state = synth.synth_state("conv_lstm", size, 9, num_out, seed=3)
    net = torch_ref.from_state(state)
    d = synth.synth_chunks_config(cfg, 1500, shard=9)
    enc = torch.from_numpy(O.compute_encoded_kmer_batch(kcb[0], kcb[1], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"]))
    sig = torch.from_numpy(d["signal"])
    with torch.no_grad():
        exact = lowp_emulation.forward(net, sig, enc, sites=()).numpy()
        for dtype in ("bf16", "f16"):
            model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=kcb), device=0, dtype=dtype)
            out = model.infer_chunks(d["signal"], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"], kcb).astype(np.float64)
            gpu = np.abs(out - exact)
            print(f"{cfg} {size} {dtype} GPU vs exact: mean {gpu.mean():.3e} q99 {np.quantile(gpu, 0.99):.3e} max {gpu.max():.3e}")
            sets = {"stream-sites": SITES, "fused-sites": lowp_emulation.ALL_SITES, "conv only": SITES[:7], "lstm only": SITES[7:],
                    "stream + sig1": SITES + ("aconv.sig1",), "stream + wconv.sig2/seq1": SITES + ("wconv.sig2", "wconv.seq1")}
            for name, sites in sets.items():
                emu = lowp_emulation.forward(net, sig, enc, sites=sites, fmt=dtype).numpy()
                e, dd = np.abs(emu - exact), np.abs(out - emu)
                print(f"    emu[{name:26s}] vs exact: mean {e.mean():.3e} q99 {np.quantile(e, 0.99):.3e} | GPU vs this emu: mean {dd.mean():.3e} q99 {np.quantile(dd, 0.99):.3e}")

# the reference-generated golden models (torch's default weight scale): max |error| against the reference's logits
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in ("convlstm_s64_l100_o2", "convlstm_s96_l100_o2", "convlstm_s128_l100_o2", "convlstm_s40_l100_o2"):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"model_{name}.npz"))
    state = O.state_from_npz(g)
    size, kb, ka, L, num_out = (int(x) for x in g["params"])
    net = torch_ref.from_state(state)
    enc = torch.from_numpy(O.compute_encoded_kmer_batch(kb, ka, g["seqs"], g["maps"], g["lens"]))
    with torch.no_grad():
        exact = lowp_emulation.forward(net, torch.from_numpy(g["sigs"]), enc, sites=()).numpy()
        for dtype in ("bf16", "f16"):
            model = model_from_state(state, dict(chunk_context=(L // 2, L - L // 2), kmer_context_bases=(kb, ka)), device=0, dtype=dtype)
            out = model.infer_chunks(g["sigs"], g["seqs"], g["maps"], g["lens"], (kb, ka))
            emu = lowp_emulation.forward(net, torch.from_numpy(g["sigs"]), enc, sites=SITES if model.kernel_size > 64 else lowp_emulation.ALL_SITES, fmt=dtype).numpy()
            print(f"golden {name} {dtype}: GPU max |err| vs reference logits {np.abs(out - g['logits']).max():.3e} (mean {np.abs(out - g['logits']).mean():.3e}); "
                  f"emulation max {np.abs(emu - exact).max():.3e} (mean {np.abs(emu - exact).mean():.3e}); kernel size {model.kernel_size}")
