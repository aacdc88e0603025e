"""cProfile of ValidationLogger.run_validation over a synthetic on-disk dataset (host-side costs)."""
import cProfile
import os
import pstats
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from remora_amd.data_chunks import CoreRemoraDataset, RemoraDataset, dataset_metadata
from remora_amd.model_util import model_from_state
from remora_amd.synth import synth_chunks, synth_state
from remora_amd.validate import ValidationLogger

n = 1 << 20
td = tempfile.mkdtemp()
data = synth_chunks(n, 100, 20, (4, 4), seed=3)
md = dataset_metadata(allocate_size=n, max_seq_len=20, mod_bases=["m"], mod_long_names=["5mC"], motif_sequences=["CG"],
                      motif_offsets=[0], chunk_context=(50, 50), kmer_context_bases=(4, 4))
ds = CoreRemoraDataset(os.path.join(td, "val"), mode="w", metadata=md)
ds.write_batch({k: data[k] for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths", "labels")})
ds.flush()
model = model_from_state(synth_state("conv_lstm", 64, 9, 2, seed=0), dict(chunk_context=(50, 50), kmer_context_bases=(4, 4)), device=0)
rd = RemoraDataset([CoreRemoraDataset(os.path.join(td, "val"), infinite_iter=False)], [1.0], batch_size=131072, super_batch_size=1 << 20)
val = ValidationLogger(open(os.devnull, "w"))
val.run_validation(model, ["m"], None, rd, 0.1)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
val.run_validation(model, ["m"], None, rd, 0.1)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
