"""Repeated batched calls (refinement + motif scan + extraction + inference): device memory must stop shrinking."""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch, gc
from remora_amd import synth
from remora_amd.data_chunks import RemoraRead
from remora_amd.inference import call_reads_mods
from remora_amd.model_util import model_from_state
from remora_amd.refine_signal_map import SigMapRefiner
import bench_refine
st = synth.synth_state()
table, center, base = bench_refine.synth_reads(32, 3000, seed=5)
refiner = SigMapRefiner(_levels_array=table, center_idx=center, do_rough_rescale=True, scale_iters=0)
md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"],
          can_base="C", base_start_justify=False, offset=0, sig_map_refiner=refiner)
model = model_from_state(st, md, device=0)
def fresh(n):
    return [RemoraRead(dacs=base[i % 32][0], shift=400.0, scale=60.0, seq_to_sig_map=base[i % 32][1].copy(), int_seq=base[i % 32][2]) for i in range(n)]
frees=[]
for it in range(40):
    call_reads_mods(fresh(256 + (it % 5) * 64), model, md)
    torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()
    frees.append(torch.cuda.mem_get_info()[0] / 2**20)
print("free MiB after iterations 1,5,10,20,40:", [round(frees[i]) for i in (0, 4, 9, 19, 39)])
assert frees[39] >= frees[9] - 64, "device memory keeps shrinking"
print("no leak (single-batch path with a refiner)")
# the three-thread pipeline (batches of >= 1024 reads, no refiner): persistent streams, pinned double buffers, a second engine
md2 = dict(md, sig_map_refiner=None)
frees = []
for it in range(24):
    call_reads_mods(fresh(1024 + (it % 4) * 256), model, md2)
    torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()
    frees.append(torch.cuda.mem_get_info()[0] / 2**20)
print("pipelined: free MiB after iterations 1,4,8,16,24:", [round(frees[i]) for i in (0, 3, 7, 15, 23)])
assert frees[23] >= frees[7] - 64, "device memory keeps shrinking (pipelined path)"
import resource
print("no leak; max RSS MiB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024)
