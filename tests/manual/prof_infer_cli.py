#!/usr/bin/env python3
"""File-to-file throughput of `python -m remora_amd infer from_pod5_and_bam` against the number of processes per GPU
(the host side is Python: it scales with processes; src/remora/inference.py:488-572 spreads the same work over worker
processes).  Input: the reference's 14 test alignments REP times over (same read ids: the POD5 side decodes 14 distinct
signals per batch; BAM parse, move tables, normalisation, extraction, inference, MM/ML formatting and BAM output are per
record).  Test infrastructure (mints the model file from the oracle's torch restatement); run by hand on a GPU box.

    python tests/manual/prof_infer_cli.py [REP=600] [procs list=1,2,4,8,16] [dtype=fp32] [bam level=6] ["--reference-anchored"]"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import oracle as O  # noqa: E402
from oracle import torch_ref  # noqa: E402
from remora_amd import io as rio  # noqa: E402

REP = int(sys.argv[1]) if len(sys.argv) > 1 else 600
PROCS = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8,16").split(",")]
DT = sys.argv[3] if len(sys.argv) > 3 else "fp32"
LEVEL = sys.argv[4] if len(sys.argv) > 4 else "6"
EXTRA = sys.argv[5].split() if len(sys.argv) > 5 else []  # e.g. "--reference-anchored"
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
g = np.load(os.path.join(ROOT, "tests", "golden", "real_reads_can.npz"))
pt = os.path.join(tmp, "model.pt")
torch.jit.save(torch.jit.script(torch_ref.from_state(O.state_from_npz(g))), pt, _extra_files={"meta.txt": str(g["meta_txt"])})
print(f"{n} records, {os.path.getsize(big) / 1e6:.0f} MB BAM, model dtype {DT}", flush=True)
res = {}
for p in PROCS:
    out = os.path.join(tmp, f"out{p}.bam")
    t = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "remora_amd", "infer", "from_pod5_and_bam", pod5, big, "--model", pt, "--out-bam", out,
                        "--dtype", DT, "--procs-per-gpu", str(p), "--reads-per-batch", "512", "--bam-level", LEVEL] + EXTRA, cwd=ROOT, capture_output=True, text=True)
    wall = time.perf_counter() - t
    if os.environ.get("RMR_INFER_TIMING"):
        print("\n".join(ln for ln in r.stderr.splitlines() if ln.startswith("[")), flush=True)
    m = re.search(r"= (\d+) reads/s", r.stderr)
    res[p] = {"reads_per_s_excl_startup": int(m.group(1)) if m else None, "wall_s": wall, "rc": r.returncode}
    print(f"procs/gpu {p:3d}: {res[p]['reads_per_s_excl_startup']} reads/s (work only), {n / wall:.0f} reads/s incl. start-up ({wall:.1f} s)"
          f"{'' if r.returncode == 0 else ' FAILED: ' + r.stderr[-500:]}", flush=True)
    if r.returncode == 0 and p == PROCS[0]:
        first = out
    elif r.returncode == 0:
        a, b = open(first, "rb").read(), open(out, "rb").read()
        import gzip
        same = gzip.decompress(a) == gzip.decompress(b)
        print(f"              output identical to procs/gpu {PROCS[0]}: {same}", flush=True)
print("RESULT " + json.dumps({"records": n, "dtype": DT, "bam_level": LEVEL, "by_procs": res}))
