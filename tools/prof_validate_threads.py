import cProfile, pstats, os, sys, tempfile, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from remora_amd.data_chunks import CoreRemoraDataset, RemoraDataset, dataset_metadata
from remora_amd.model_util import model_from_state
from remora_amd.synth import synth_chunks, synth_state
from remora_amd.validate import ValidationLogger
n = 1 << 20
td = tempfile.mkdtemp()
data = synth_chunks(n, 100, 20, (4, 4), seed=3)
md = dataset_metadata(allocate_size=n, max_seq_len=20, mod_bases=["m"], mod_long_names=["5mC"], motif_sequences=["CG"], motif_offsets=[0], chunk_context=(50, 50), kmer_context_bases=(4, 4))
ds = CoreRemoraDataset(os.path.join(td, "val"), mode="w", metadata=md)
ds.write_batch({k: data[k] for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths", "labels")}); ds.flush()
model = model_from_state(synth_state("conv_lstm", 64, 9, 2, seed=0), dict(chunk_context=(50, 50), kmer_context_bases=(4, 4)), device=0)
val = ValidationLogger(open(os.devnull, "w"))
for bs in (131072, 262144):
    for nt in (2, 4, 8, 16):
        os.environ["RMR_VALIDATE_THREADS"] = str(nt)
        rd = RemoraDataset([CoreRemoraDataset(os.path.join(td, "val"), infinite_iter=False)], [1.0], batch_size=bs, super_batch_size=1 << 20)
        val.run_validation(model, ["m"], None, rd, 0.1)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); val.run_validation(model, ["m"], None, rd, 0.1); ts.append(time.perf_counter() - t0)
        print(f"batch {bs} threads {nt}: {n / min(ts) / 1e6:.2f} M chunks/s (best of 3: {[round(t*1e3,1) for t in ts]} ms)", flush=True)
os.environ["RMR_VALIDATE_THREADS"] = "8"
rd = RemoraDataset([CoreRemoraDataset(os.path.join(td, "val"), infinite_iter=False)], [1.0], batch_size=131072, super_batch_size=1 << 20)
pr = cProfile.Profile(); pr.enable(); val.run_validation(model, ["m"], None, rd, 0.1); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
