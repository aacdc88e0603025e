import sys, time, os
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import numpy as np, torch
from remora_amd import synth
from remora_amd.data_chunks import RemoraRead
from remora_amd.inference import call_reads_mods
from remora_amd.model_util import model_from_state
md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"],
          can_base="C", base_start_justify=False, offset=0, sig_map_refiner=None)
reads = []
for i in range(2048):
    r = synth.synth_read(5000, idx=i)
    reads.append(RemoraRead(dacs=r["dacs"], shift=r["shift"], scale=r["scale"], seq_to_sig_map=r["seq_to_sig_map"], int_seq=r["int_seq"], read_id=f"r{i}"))
for dt in ("fp32", "bf16"):
    model = model_from_state(synth.synth_state("conv_lstm", 64, 9, 2, seed=2), md, device=0, dtype=dt)
    for sb in ("128", "256", "384", "512", "1024"):
        os.environ["RMR_READS_SUBBATCH"] = sb
        for _ in range(2):
            call_reads_mods(reads, model, md)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            call_reads_mods(reads, model, md)
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t) / 5
        print(dt, "subbatch", sb, "ms", round(dtm * 1e3, 2), "reads/s", round(2048 / dtm))
