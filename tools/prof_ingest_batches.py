"""cProfile of the batch ingest alone (io.iter_ingest_batches: native BAM batches -> POD5 rows -> VBZ decode, move tables and
read assembly on the GPU) on the replicated BAM of tests/manual/prof_infer_cli.py: where the ingest thread of a
single-process `infer` spends its 35 us per record.

    python tools/prof_ingest_batches.py [REP=6000] [ref (reference-anchored, on the modified-base test alignments)] [batch=512]"""
import cProfile
import os
import pstats
import struct
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from remora_amd import io as rio  # noqa: E402

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
REF = len(sys.argv) > 2 and sys.argv[2] == "ref"
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 512
data = os.path.join(ROOT, "tests", "golden", "data")
stem = "mod" if REF else "can"
pod5, bam = os.path.join(data, f"{stem}_reads.pod5"), os.path.join(data, f"{stem}_mappings.bam")
big = os.path.join(tempfile.mkdtemp(), "big.bam")
recs = list(rio.iter_bam_records(bam, want_ref=False))
with rio.BamWriter(big, rio.read_bam_header_bytes(bam), level=1) as w:
    for _ in range(REP):
        for r in recs:
            raw = bytes(r.raw)
            w.write(struct.pack("<i", len(raw)) + raw)
n_rec = REP * len(recs)
for _ in rio.iter_ingest_batches(pod5, bam, batch=BATCH, device=0, ref_anchored=REF):
    pass
t = time.perf_counter()
k = sum(rb.n for rb, _ in rio.iter_bam_raw_batches(big, batch=BATCH, want_ref=REF))
dt = time.perf_counter() - t
print(f"native BAM batches alone: {k / dt:.0f} records/s ({dt / k * 1e6:.1f} us per record)")
t = time.perf_counter()
n = sum(len(ib) for ib in rio.iter_ingest_batches(pod5, big, batch=BATCH, device=0, ref_anchored=REF))
dt = time.perf_counter() - t
print(f"iter_ingest_batches: {n} of {n_rec} records, {n / dt:.0f} records/s ({dt / n * 1e6:.1f} us per record)")
os.environ["RMR_INFER_TIMING"] = "1"
rio.INGEST_CLOCK.clear()
t = time.perf_counter()
n = sum(len(ib) for ib in rio.iter_ingest_batches(pod5, big, batch=BATCH, device=0, ref_anchored=REF))
dt = time.perf_counter() - t
print(f"sections of _ingest_batch, us per record (wall {dt / n * 1e6:.1f}): " + ", ".join(f"{k} {v / n * 1e6:.1f}" for k, v in rio.INGEST_CLOCK.items()))
del os.environ["RMR_INFER_TIMING"]
pr = cProfile.Profile()
pr.enable()
for ib in rio.iter_ingest_batches(pod5, big, batch=BATCH, device=0, ref_anchored=REF):
    pass
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
