import cProfile, pstats, time, os, sys
import numpy as np, torch
from remora_amd import synth
from remora_amd.data_chunks import RemoraRead
from remora_amd.inference import call_reads_mods
from remora_amd.model_util import model_from_state
md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"],
          can_base="C", base_start_justify=False, offset=0, sig_map_refiner=None)
model = model_from_state(synth.synth_state("conv_lstm", 64, 9, 2, seed=2), md, device=0, dtype="bf16")
reads = []
for i in range(2048):
    r = synth.synth_read(5000, idx=i)
    reads.append(RemoraRead(dacs=r["dacs"], shift=r["shift"], scale=r["scale"], seq_to_sig_map=r["seq_to_sig_map"], int_seq=r["int_seq"], read_id=f"r{i}"))
for _ in range(2):
    call_reads_mods(reads, model, md)
t = time.perf_counter()
for _ in range(5):
    call_reads_mods(reads, model, md)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 5
print("ms per 2048 reads", dt * 1e3, "reads/s", 2048 / dt)
for sb in ():
    os.environ["RMR_READS_SUBBATCH"] = sb
    call_reads_mods(reads, model, md)
    t = time.perf_counter()
    for _ in range(5):
        call_reads_mods(reads, model, md)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print("subbatch", sb, "ms", dt * 1e3, "reads/s", 2048 / dt)
os.environ["RMR_READS_SUBBATCH"] = "512"
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    call_reads_mods(reads, model, md)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
# the worker thread is not seen by cProfile: time staging alone
from remora_amd.data_chunks import DeviceReads
t = time.perf_counter()
for _ in range(5):
    for i in range(0, 2048, 512):
        DeviceReads(reads[i:i + 512], model.engine)
torch.cuda.synchronize()
print("staging alone ms per 2048 reads", (time.perf_counter() - t) / 5 * 1e3)

from remora_amd.inference import call_read_mods
for r in reads[:8]:
    call_read_mods(r, model, md)
t = time.perf_counter()
for r in reads[:64]:
    call_read_mods(r, model, md)
print("single read ms", (time.perf_counter() - t) / 64 * 1e3)
pr = cProfile.Profile()
pr.enable()
for r in reads[:64]:
    call_read_mods(r, model, md)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
