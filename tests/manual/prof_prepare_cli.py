#!/usr/bin/env python3
"""Wall time of `python -m remora_amd dataset prepare` against the number of processes per GPU, on the reference's 14
test alignments REP times over (same read ids: the POD5 side decodes 14 distinct signals per batch; everything else is
per record).  Run by hand on a GPU box:   python tests/manual/prof_prepare_cli.py [REP=3000] [procs list=1,4,6]"""
import os
import struct
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from remora_amd import io as rio  # noqa: E402

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
PROCS = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,4,6").split(",")]
data = os.path.join(ROOT, "tests", "golden", "data")
pod5, bam = os.path.join(data, "mod_reads.pod5"), os.path.join(data, "mod_mappings.bam")
tmp = tempfile.mkdtemp()
big = os.path.join(tmp, "big.bam")
recs = list(rio.iter_bam_records(bam, want_ref=False))
with rio.BamWriter(big, rio.read_bam_header_bytes(bam), level=1) as w:
    for _ in range(REP):
        for r in recs:
            raw = bytes(r.raw)
            w.write(struct.pack("<i", len(raw)) + raw)
n = REP * len(recs)
print(f"{n} records, {os.path.getsize(big) / 1e6:.0f} MB BAM", flush=True)
for p in PROCS:
    out = os.path.join(tmp, f"chunks{p}")
    t = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "remora_amd", "dataset", "prepare", pod5, big, "--output-path", out, "--mod-base", "m", "5mC",
                        "--motif", "CG", "0", "--chunk-context", "50", "50", "--skip-shuffle"] + (["--procs-per-gpu", str(p)] if p > 1 else []),
                       cwd=ROOT, capture_output=True, text=True)
    wall = time.perf_counter() - t
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("Extracted")]
    for ln in r.stderr.splitlines():
        if ln.startswith("[prepare"):
            print("    " + ln, flush=True)
    print(f"procs/gpu {p:3d}: {n / wall:8.0f} reads/s incl. start-up ({wall:.1f} s)  {last[-1] if last else r.stderr[-400:]}", flush=True)
