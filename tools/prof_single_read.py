"""cProfile of the single-read API (call_read_mods, the reference's drop-in signature) on 5 kb reads."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from remora_amd import synth
from remora_amd.data_chunks import RemoraRead
from remora_amd.inference import call_read_mods
from remora_amd.model_util import model_from_state

st = synth.synth_state()
md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"],
          can_base="C", base_start_justify=False, offset=0, sig_map_refiner=None, reverse_signal=False, pa_scaling=None)
model = model_from_state(st, md, device=0)
rs = []
for i in range(128):
    r = synth.synth_read(5000, idx=i)
    rs.append(RemoraRead(dacs=r["dacs"], shift=r["shift"], scale=r["scale"], seq_to_sig_map=r["seq_to_sig_map"], int_seq=r["int_seq"]))
for r in rs[:8]:
    call_read_mods(r, model, md)
torch.cuda.synchronize()
t = time.perf_counter()
for r in rs:
    call_read_mods(r, model, md)
print("ms per read", (time.perf_counter() - t) / len(rs) * 1e3)
pr = cProfile.Profile()
pr.enable()
for r in rs:
    call_read_mods(r, model, md)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
from remora_amd.engine import get_engine

eng = get_engine(0)
eng.profile_reset()
eng.profile_enable(True)
for r in rs[:32]:
    call_read_mods(r, model, md)
eng.profile_enable(False)
tot = 0.0
for k, (ms, n) in eng.profile().items():
    print(f"  {k:18s} {ms / 32 * 1e3:8.1f} us per read  x{n / 32:.1f}")
    tot += ms
print("  kernel sum per read (us)", tot / 32 * 1e3)
