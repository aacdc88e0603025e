"""Measured legs for SURVEY §8f row N4 (run on an MI355X; bench.py includes the result as "dataset_etl"):
  * validate: a synthetic on-disk CoreRemoraDataset (memmapped rows, the reference's directory format) through
    RemoraDataset iteration + ValidationLogger.run_validation (fused kernels; logits back on the host, metrics).
  * prepare: synthetic aligned io.Read objects (5 kb, all-match CIGAR) through extract_chunk_arrays_from_reads
    (focus bases, down-sampling to 15 chunks per read, upload, extraction) + write_chunk_arrays into the memmaps.
Usage: python tools/bench_dataset.py [--chunks N] [--reads N]"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synth_io_reads(n_reads, n_bases, seed=0):
    from remora_amd import io as rio

    rng = np.random.default_rng(seed)
    reads = []
    for i in range(n_reads):
        seq = "".join(np.array(list("ACGT"))[rng.integers(0, 4, n_bases)])
        q2s = np.concatenate([[0], np.cumsum(rng.integers(5, 16, n_bases))]).astype(np.int64)
        dacs = rng.integers(300, 700, int(q2s[-1])).astype(np.int16)
        r = rio.Read(read_id=f"synth-{i:08d}", dacs=dacs, seq=seq, query_to_signal=q2s, shift_dacs_to_norm=500.0,
                     scale_dacs_to_norm=80.0, ref_seq=seq, cigar=[(0, n_bases)],
                     ref_reg=rio.RefRegion("chr1", "+", 1000 * i, 1000 * i + n_bases))
        reads.append((r, None))
    return reads


def measure(n_chunks=1 << 20, n_reads=2048, n_bases=5000, device=0, batch_size=131072):
    import torch

    from remora_amd.data_chunks import CoreRemoraDataset, RemoraDataset, dataset_metadata
    from remora_amd.model_util import model_from_state
    from remora_amd.prepare_train_data import extract_chunk_arrays_from_reads
    from remora_amd.synth import synth_chunks, synth_state
    from remora_amd.util import Motif
    from remora_amd.validate import ValidationLogger

    out = {}
    td = tempfile.mkdtemp(prefix="rmr_ds_")
    try:
        # ---- validate from an on-disk dataset ----
        data = synth_chunks(n_chunks, 100, 20, (4, 4), seed=3)
        md = dataset_metadata(allocate_size=n_chunks, max_seq_len=20, mod_bases=["m"], mod_long_names=["5mC"],
                              motif_sequences=["CG"], motif_offsets=[0], chunk_context=(50, 50), kmer_context_bases=(4, 4))
        ds = CoreRemoraDataset(os.path.join(td, "val"), mode="w", metadata=md)
        t0 = time.perf_counter()
        ds.write_batch({"signal": data["signal"], "sequence": data["sequence"],
                        "sequence_to_signal_mapping": data["sequence_to_signal_mapping"],
                        "sequence_lengths": data["sequence_lengths"], "labels": data["labels"]})
        ds.flush()
        t_write = time.perf_counter() - t0
        model = model_from_state(synth_state("conv_lstm", 64, 9, 2, seed=0),
                                 dict(chunk_context=(50, 50), kmer_context_bases=(4, 4)), device=device)
        rd = RemoraDataset([CoreRemoraDataset(os.path.join(td, "val"), infinite_iter=False)], [1.0], batch_size=batch_size,
                           super_batch_size=1 << 20)
        val = ValidationLogger(open(os.devnull, "w"))
        val.run_validation(model, ["m"], None, rd, 0.1)  # warm-up (page cache, kernels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ms = val.run_validation(model, ["m"], None, rd, 0.1)
        t_val = time.perf_counter() - t0
        out["validate"] = {"chunks": n_chunks, "chunks_per_s": n_chunks / t_val, "write_chunks_per_s": n_chunks / t_write,
                           "acc": float(ms.acc), "batch_size": batch_size,
                           "note": "memmapped dataset rows -> pinned ring (reader thread) -> H2D on a copy stream -> fused kernels -> "
                                   "rmr_validation_tally on the device (softmax, calls, confusion counts, cross entropy); 5 B per "
                                   "chunk back for the filtered columns' quantile (ValidationLogger.run_validation)"}
        # ---- dataset prepare (without the file parsing) ----
        reads = synth_io_reads(n_reads, n_bases, seed=1)
        motifs = [Motif("CG", 0)]
        md = dataset_metadata(allocate_size=15 * n_reads, max_seq_len=20, mod_bases=["m"], mod_long_names=["5mC"],
                              motif_sequences=["CG"], motif_offsets=[0], chunk_context=(50, 50), kmer_context_bases=(4, 4),
                              extra_arrays={"read_ids": ("<U36", "Read identifier"),
                                            "read_focus_bases": ("int64", "Position within read training sequence")})
        for rep in range(2):
            shutil.rmtree(os.path.join(td, "prep"), ignore_errors=True)
            dsw = CoreRemoraDataset(os.path.join(td, "prep"), mode="w", metadata=md.copy())
            np.random.seed(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for st in range(0, n_reads, 256):
                arrs, ids, keep, _ = extract_chunk_arrays_from_reads(reads[st : st + 256], 1, motifs, None, None, 15, (50, 50),
                                                                    (4, 4), False, 0, False)
                dsw.write_chunk_arrays(arrs, keep=keep, read_ids=ids)
            dsw.flush()
            t_prep = time.perf_counter() - t0
        out["prepare"] = {"reads": n_reads, "bases_per_read": n_bases, "chunks_written": dsw.size,
                          "reads_per_s": n_reads / t_prep, "chunks_per_s": dsw.size / t_prep,
                          "note": "aligned reads in memory -> reference-anchored training reads (ref_to_signal, motif hits, "
                                  "15 random focus bases per read) -> upload + GPU extraction -> dataset memmaps; 256 reads "
                                  "per batch; POD5/BAM parsing not included"}
    finally:
        shutil.rmtree(td, ignore_errors=True)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=1 << 20)
    ap.add_argument("--reads", type=int, default=2048)
    a = ap.parse_args()
    print(json.dumps(measure(a.chunks, a.reads), indent=1))
