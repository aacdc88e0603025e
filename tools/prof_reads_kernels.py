"""Per-kernel HIP-event times of one batched call_reads_mods (2048 reads x 5 kb, with a refiner)."""
import ctypes, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
from remora_amd import synth, _lib as L
from remora_amd.data_chunks import RemoraRead
from remora_amd.engine import get_engine
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
N = 2048
def fresh():
    return [RemoraRead(dacs=base[i % 64][0], shift=400.0, scale=60.0, seq_to_sig_map=base[i % 64][1].copy(), int_seq=base[i % 64][2]) for i in range(N)]
call_reads_mods(fresh(), model, md)
eng = get_engine(0); lib = L.lib()
L.check(lib.rmr_profile_enable(eng.handle, 1)); L.check(lib.rmr_profile_reset(eng.handle))
rs = fresh(); torch.cuda.synchronize(); t = time.perf_counter(); res = call_reads_mods(rs, model, md); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("batch of", N, "reads:", round(dt * 1e3, 1), "ms ->", round(N / dt), "reads/s;", sum(r[2].size for r in res), "chunks")
tot = 0
for i in range(lib.rmr_profile_num_kernels()):
    ms, cnt = ctypes.c_double(), ctypes.c_int64()
    L.check(lib.rmr_profile_get(eng.handle, i, ctypes.byref(ms), ctypes.byref(cnt)))
    if cnt.value:
        print(f"  {lib.rmr_profile_kernel_name(i).decode():18s} {ms.value:8.3f} ms  x{cnt.value}")
        tot += ms.value
print("  kernel sum", round(tot, 2), "ms")
