"""fp32-MFMA path and bf16xN paths against a float64 CPU evaluation of the same network (synthetic
bench weights + data): who is closer to the truth, and by how much."""
import sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import oracle as O, torch_ref
from remora_amd import synth
from remora_amd.model_util import model_from_state
torch.set_num_threads(32)
for cfg, no in (("C100", 2), ("C200", 3)):
    cc, kcb, msl, _, _ = synth.CONFIGS[cfg]
    st = synth.synth_state("conv_lstm", 64, 9, no, seed=0)
    d = synth.synth_chunks_config(cfg, 3000)
    enc = O.compute_encoded_kmer_batch(4, 4, d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])
    net64 = torch_ref.from_state(st).double()
    net32 = torch_ref.from_state(st)
    with torch.no_grad():
        ref64 = net64(torch.from_numpy(d["signal"]).double(), torch.from_numpy(enc).double()).numpy()
        ref32 = net32(torch.from_numpy(d["signal"]), torch.from_numpy(enc)).numpy()
    print(cfg, "logit abs max", np.abs(ref64).max(), "torch fp32 CPU vs fp64:", np.abs(ref32 - ref64).max())
    for dt in ("fp32", "bf16x6", "bf16x3", "bf16"):
        m = model_from_state(st, dict(chunk_context=cc, kmer_context_bases=kcb), device=0, dtype=dt)
        out = m.infer_chunks(d["signal"], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"], kcb)
        print("   ", dt, "vs fp64:", float(np.abs(out - ref64).max()), " vs torch fp32:", float(np.abs(out - ref32).max()))
