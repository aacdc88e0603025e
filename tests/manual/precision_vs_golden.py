import numpy as np, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import oracle as O
from remora_amd.model_util import model_from_state
for name in ["convlstm_s64_l100_o2", "convlstm_s64_l200_o3", "convlstm_s64_l100_k23"]:
    g = np.load(f'/root/repo/tests/golden/model_{name}.npz')
    state = O.state_from_npz(g); size, kb, ka, L, no = (int(x) for x in g["params"])
    for dt in ["fp32", "bf16x6", "bf16x3", "bf16"]:
        m = model_from_state(state, dict(chunk_context=(L//2, L-L//2), kmer_context_bases=(kb, ka)), device=0, dtype=dt)
        out = m.infer_chunks(g["sigs"], g["seqs"], g["maps"], g["lens"], (kb, ka))
        print(name, dt, "max|dlogit|", float(np.abs(out - g["logits"]).max()))
