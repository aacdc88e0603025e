"""Throughput / profile of the POD5+BAM ingest (iter_reads_from_pod5_and_bam) on the reference's test files,
read repeatedly (14 alignments per pass)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remora_amd import io as rio

data = os.path.join(ROOT, "tests", "golden", "data")
pod5, bam = os.path.join(data, "can_reads.pod5"), os.path.join(data, "can_mappings.bam")
list(rio.iter_reads_from_pod5_and_bam(pod5, bam))
t = time.perf_counter()
n = 0
for _ in range(30):
    for read, err in rio.iter_reads_from_pod5_and_bam(pod5, bam):
        n += 1
dt = time.perf_counter() - t
print(f"ingest: {n / dt:.0f} reads/s ({dt / n * 1e3:.2f} ms per read)")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    list(rio.iter_reads_from_pod5_and_bam(pod5, bam))
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
