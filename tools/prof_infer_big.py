"""File-to-file throughput of infer_from_pod5_and_bam on a BAM that holds the reference's 14 test alignments
`REP` times over (same read ids, so the POD5 side decodes 14 distinct signals per batch - the BAM parse, move-table
expansion, normalisation, chunk extraction, inference, tag formatting and BAM output are all per record)."""
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
from remora_amd import synth
from remora_amd.inference import infer_from_pod5_and_bam
from remora_amd.model_util import model_from_state

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 150
data = os.path.join(ROOT, "tests", "golden", "data")
pod5, bam = os.path.join(data, "can_reads.pod5"), os.path.join(data, "can_mappings.bam")
tmp = tempfile.mkdtemp()
big = os.path.join(tmp, "big.bam")
recs = list(rio.iter_bam_records(bam, want_ref=False))
with rio.BamWriter(big, rio.read_bam_header_bytes(bam)) as w:
    for _ in range(REP):
        for r in recs:
            raw = bytes(r.raw)
            w.write(struct.pack("<i", len(raw)) + raw)
n = REP * len(recs)
md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"],
          can_base="C", base_start_justify=False, offset=0, sig_map_refiner=None, reverse_signal=False, pa_scaling=None)
model = model_from_state(synth.synth_state(), md, device=0, dtype=os.environ.get("DT", "fp32"))
out = os.path.join(tmp, "o.bam")
infer_from_pod5_and_bam(pod5, bam, model, md, out)
for rpb in (256, 512):
    t = time.perf_counter()
    stats = infer_from_pod5_and_bam(pod5, big, model, md, out, reads_per_batch=rpb)
    dt = time.perf_counter() - t
    print(f"infer, {n} records, reads_per_batch {rpb}: {n / dt:.0f} reads/s ({dt / n * 1e3:.3f} ms per read) {dict(stats)}")
pr = cProfile.Profile()
pr.enable()
infer_from_pod5_and_bam(pod5, big, model, md, out)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
