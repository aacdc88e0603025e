"""The numbers the 16-bit parity gates of tests/test_gpu_fused.py are set from (gate = measured level x 1.2 .. 1.3):
max |logit error| of dtype f16 / bf16 on the reference-generated golden ConvLSTM models, and - on the synthetic C100 / C200
networks - against the oracle's fp32 forward on 8192 chunks and against the fp32 HIP path on 100 k chunks.

    python tools/measure_16bit_parity.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    from conftest import golden
    from oracle import oracle as O
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    out = {}
    for name in ("convlstm_s64_l100_o2", "convlstm_s64_l200_o3", "convlstm_s64_l100_k23"):
        g = golden(f"model_{name}.npz")
        state = O.state_from_npz(g)
        size, kb, ka, L, num_out = (int(x) for x in g["params"])
        for dtype in ("f16", "bf16"):
            model = model_from_state(state, dict(chunk_context=(L // 2, L - L // 2), kmer_context_bases=(kb, ka)), device=0, dtype=dtype)
            lg = model.infer_chunks(g["sigs"], g["seqs"], g["maps"], g["lens"], (kb, ka))
            out[f"golden/{name}/{dtype}"] = dict(max=float(np.abs(lg - g["logits"]).max()), n=int(lg.shape[0]), scale=float(np.abs(g["logits"]).max()))
    for cfg in ("C100", "C200"):
        cc, kcb, _, num_out, _ = synth.CONFIGS[cfg]
        state = synth.synth_state("conv_lstm", 64, 9, num_out, seed=0)
        md = dict(chunk_context=cc, kmer_context_bases=kcb)
        net = torch_ref.from_state(state)
        d = synth.synth_chunks_config(cfg, 100_000, shard=3)
        keys = ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")
        sub = {k: d[k][:8192] for k in keys}
        enc = torch.from_numpy(O.compute_encoded_kmer_batch(kcb[0], kcb[1], sub["sequence"], sub["sequence_to_signal_mapping"], sub["sequence_lengths"]))
        with torch.no_grad():
            ref_cpu = net(torch.from_numpy(sub["signal"]), enc).numpy()
        dev = [torch.from_numpy(d[k]).cuda() for k in keys]
        ref_gpu = model_from_state(state, md, device=0, dtype="fp32").infer_chunks(*dev, kcb)
        out[f"synth/{cfg}/fp32_vs_oracle_8k"] = dict(max=float(np.abs(ref_gpu[:8192].cpu().numpy() - ref_cpu).max()))
        for dtype in ("f16", "bf16"):
            lg = model_from_state(state, md, device=0, dtype=dtype).infer_chunks(*dev, kcb)
            e8 = np.abs(lg[:8192].cpu().numpy() - ref_cpu)
            e100 = (lg - ref_gpu).abs()
            per_chunk = e100.max(dim=1).values.float()
            out[f"synth/{cfg}/{dtype}"] = dict(vs_oracle_8k_max=float(e8.max()), vs_oracle_8k_mean=float(e8.mean()),
                                               vs_fp32_100k_max=float(e100.max()), vs_fp32_100k_mean=float(e100.mean()),
                                               vs_fp32_100k_p999=float(torch.quantile(per_chunk, 0.999)),
                                               logit_scale=float(np.abs(ref_cpu).max()))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
