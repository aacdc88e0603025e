"""Profile of the POD5+BAM ingest alone (iter_reads_from_pod5_and_bam + into_remora_read) on the replicated BAM of
tools/prof_infer_big.py."""
import cProfile
import os
import pstats
import struct
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remora_amd import io as rio

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 150
data = os.path.join(ROOT, "tests", "golden", "data")
pod5, bam = os.path.join(data, "can_reads.pod5"), os.path.join(data, "can_mappings.bam")
big = os.path.join(tempfile.mkdtemp(), "big.bam")
recs = list(rio.iter_bam_records(bam, want_ref=False))
with rio.BamWriter(big, rio.read_bam_header_bytes(bam)) as w:
    for _ in range(REP):
        for r in recs:
            raw = bytes(r.raw)
            w.write(struct.pack("<i", len(raw)) + raw)
list(rio.iter_reads_from_pod5_and_bam(pod5, bam, parse_ref_align=False))
t = time.perf_counter()
n = 0
for read, err in rio.iter_reads_from_pod5_and_bam(pod5, big, parse_ref_align=False):
    read.into_remora_read(False)
    n += 1
dt = time.perf_counter() - t
print(f"ingest: {n / dt:.0f} reads/s ({dt / n * 1e3:.3f} ms per read)")
pr = cProfile.Profile()
pr.enable()
for read, err in rio.iter_reads_from_pod5_and_bam(pod5, big, parse_ref_align=False):
    read.into_remora_read(False)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
