"""Timing ablations of wino_conv_kernel (GPU; experiment build: `make -C remora_amd/csrc abl`, loaded through REMORA_HIP_LIB).
RMR_WINO_ABLATE bits: 1 no input transform / fetch, 2 no MFMAs, 4 no output stage.  Prints ms per launch of conv_merge1 (250 k chunks)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("REMORA_HIP_LIB", os.path.join(ROOT, "remora_amd", "libremora_hip_abl.so"))
import torch  # noqa: E402

from remora_amd import synth  # noqa: E402
from remora_amd.engine import get_engine  # noqa: E402
from remora_amd.model_util import model_from_state  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "conv_lstm"
n = 250_000
model = model_from_state(synth.synth_state(arch, 64, 9, 2, seed=0), dict(chunk_context=(50, 50), kmer_context_bases=(4, 4)), device=0, dtype="fp32")
d = synth.synth_chunks_config("C100", n)
dev = [torch.from_numpy(d[k]).cuda() for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
eng = get_engine(0)
for abl in (0, 1, 2, 4, 3, 5, 6, 7):
    os.environ["RMR_WINO_ABLATE"] = str(abl)
    for _ in range(2):
        model.infer_chunks(*dev, (4, 4))
    eng.profile_reset()
    eng.profile_enable(True)
    for _ in range(5):
        model.infer_chunks(*dev, (4, 4))
    torch.cuda.synchronize()
    eng.profile_enable(False)
    prof = eng.profile()
    print(f"ablate {abl}: " + "  ".join(f"{k} {v[0] / v[1]:.3f} ms" for k, v in prof.items() if k.startswith("conv_")), flush=True)
