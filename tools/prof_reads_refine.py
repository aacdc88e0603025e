import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
from remora_amd import synth
from remora_amd.data_chunks import RemoraRead
from remora_amd.inference import call_reads_mods
from remora_amd.model_util import model_from_state
from remora_amd.refine_signal_map import SigMapRefiner
import bench_refine
st = synth.synth_state()
table, center, base = bench_refine.synth_reads(64, 5000, seed=5)
refiner = SigMapRefiner(_levels_array=table, center_idx=center, do_rough_rescale=True, scale_iters=0)
md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"],
          can_base="C", base_start_justify=False, offset=0, sig_map_refiner=refiner)
model = model_from_state(st, md, device=0)
def fresh():
    return [RemoraRead(dacs=base[i % 64][0], shift=400.0, scale=60.0, seq_to_sig_map=base[i % 64][1].copy(), int_seq=base[i % 64][2]) for i in range(2048)]
call_reads_mods(fresh(), model, md)
torch.cuda.synchronize()
rs = fresh()
t = time.perf_counter(); call_reads_mods(rs, model, md); torch.cuda.synchronize(); print("one batch", time.perf_counter() - t)
rs = fresh()
pr = cProfile.Profile(); pr.enable(); call_reads_mods(rs, model, md); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
