import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np
from remora_amd import synth
from remora_amd.data_chunks import RemoraRead
from remora_amd.inference import call_read_mods
from remora_amd.model_util import model_from_state
from remora_amd.engine import get_engine
md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"], can_base="C", base_start_justify=False, offset=0, sig_map_refiner=None)
model = model_from_state(synth.synth_state(seed=0), md, device=0, dtype="fp32")
rs = []
for i in range(64):
    r = synth.synth_read(5000, idx=i)
    rs.append(RemoraRead(dacs=r["dacs"], shift=r["shift"], scale=r["scale"], seq_to_sig_map=r["seq_to_sig_map"], int_seq=r["int_seq"], read_id=f"s{i}"))
for r in rs[:8]: call_read_mods(r, model, md)
eng = model.engine
eng.profile_reset(); eng.profile_enable(True)
for r in rs: call_read_mods(r, model, md)
eng.profile_enable(False)
p = eng.profile()
tot = 0
for k,(ms,n) in sorted(p.items(), key=lambda kv:-kv[1][0]):
    print(f"{k:16s} {ms/n*1e3:8.1f} us x {n}")
    tot += ms/n*1e3 if n==64 else 0
print("sum of per-read kernels", round(tot,1))
