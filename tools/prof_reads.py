import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import torch
from remora_amd import synth
from remora_amd.data_chunks import RemoraRead
from remora_amd.inference import call_reads_mods
from remora_amd.model_util import model_from_state
st = synth.synth_state()
md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"],
          can_base="C", base_start_justify=False, offset=0, sig_map_refiner=None)
model = model_from_state(st, md, device=0)
rs = []
for i in range(512):
    r = synth.synth_read(5000, idx=i)
    rs.append(RemoraRead(dacs=r["dacs"], shift=r["shift"], scale=r["scale"], seq_to_sig_map=r["seq_to_sig_map"], int_seq=r["int_seq"]))
call_reads_mods(rs, model, md)
torch.cuda.synchronize()
t = time.perf_counter(); call_reads_mods(rs, model, md); torch.cuda.synchronize(); print("one batch", time.perf_counter() - t)
pr = cProfile.Profile(); pr.enable(); call_reads_mods(rs, model, md); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
