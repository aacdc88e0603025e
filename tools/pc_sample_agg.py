#!/usr/bin/env python3
"""Fold rocprofv3 PC-sampling CSVs: samples per (kernel, instruction offset, instruction text) and per stall reason.
Usage: pc_sample_agg.py <rocprofv3 output dir> <out prefix>; writes <prefix>_by_pc.csv, <prefix>_head.csv and prints the
columns it found plus the top lines.  Experiment tooling (run on the GPU box by tools/pc_sample.sh)."""
import collections
import csv
import glob
import os
import sys

src, prefix = sys.argv[1], sys.argv[2]
files = [f for f in glob.glob(os.path.join(src, "**", "*.csv"), recursive=True) if "pc_sampling" in os.path.basename(f)]
print("pc sampling files:", files)
csv.field_size_limit(1 << 30)
for f in files:
    tag = "stoch" if "stochastic" in os.path.basename(f) else "host"
    with open(f, newline="") as fh:
        rd = csv.DictReader(fh)
        print(f, "columns:", rd.fieldnames)
        cols = rd.fieldnames or []
        pick = lambda *names: next((c for c in cols for n in names if c.lower() == n.lower()), None)  # noqa: E731
        c_inst = pick("Instruction")
        c_comment = pick("Instruction_Comment")
        c_issued = pick("Wave_Issued_Instruction", "Wave_Issued")
        c_type = pick("Instruction_Type")
        c_stall = pick("Stall_Reason")
        c_corr = pick("Dispatch_Id", "Correlation_Id")
        by_pc = collections.Counter()
        by_stall = collections.Counter()
        by_type = collections.Counter()
        n = 0
        head = []
        for row in rd:
            n += 1
            if len(head) < 300:
                head.append(row)
            key = (row.get(c_comment, ""), row.get(c_inst, ""), row.get(c_issued, ""), row.get(c_type, ""), row.get(c_stall, ""))
            by_pc[key] += 1
            by_stall[(row.get(c_issued, ""), row.get(c_stall, ""))] += 1
            by_type[(row.get(c_issued, ""), row.get(c_type, ""))] += 1
    print(f"{n} samples")
    with open(f"{prefix}_{tag}_by_pc.csv", "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["count", "comment", "instruction", "issued", "type", "stall"])
        for k, v in by_pc.most_common():
            w.writerow([v, *k])
    with open(f"{prefix}_{tag}_head.csv", "w", newline="") as out:
        if head:
            w = csv.DictWriter(out, fieldnames=list(head[0].keys()))
            w.writeheader()
            w.writerows(head)
    print("by (issued, stall):")
    for k, v in by_stall.most_common(30):
        print(f"  {v:9d} {100.0 * v / max(n, 1):5.1f}%  {k}")
    print("by (issued, type):")
    for k, v in by_type.most_common(30):
        print(f"  {v:9d} {100.0 * v / max(n, 1):5.1f}%  {k}")
    print("top instructions:")
    for k, v in by_pc.most_common(40):
        print(f"  {v:9d} {100.0 * v / max(n, 1):5.1f}%  {k}")
