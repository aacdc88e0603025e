#!/usr/bin/env python3
"""GPU-side timeline of a run of tools/timeline_reads.py made under
`rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d <dir>` (no counters in that pass).

Cuts the trace into the script's segments (the script wrote their start / end on every host clock; the one the profiler
stamps with is found by where the kernels fall) and, per segment, reports: how long kernels / copies ran (union of the
intervals, so concurrent streams are not double counted), the kernels by name, the host threads' time inside HIP calls by
function, the longest stretches with no kernel running and what the host was doing in them, and an ASCII chart
(one row per stream of kernels, one per copy direction, one per host thread: S = inside a synchronising call,
c = inside a copy call, l = launching, . = other HIP call).

    python tools/timeline_summary.py <trace dir> <windows.json> > profiles/r05_reads_timeline.md
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(trace_dir, suffix):
    rows = []
    for f in glob.glob(os.path.join(trace_dir, "**", f"*_{suffix}.csv"), recursive=True):
        with open(f, newline="") as fh:
            rows += list(csv.DictReader(fh))
    return rows


def union_ms(iv):
    iv = sorted(iv)
    tot, end = 0, None
    for a, b in iv:
        if end is None or a > end:
            tot += b - a
            end = b
        elif b > end:
            tot += b - end
            end = b
    return tot / 1e6


def gaps(iv, lo, hi):
    iv = sorted(iv)
    out, end = [], lo
    for a, b in iv:
        if a > end:
            out.append((end, a))
        end = max(end, b)
    if hi > end:
        out.append((end, hi))
    return out


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void rmr::", "").replace("rmr::", "")
    return name.split("(")[0][:60]


SYNC = ("Synchronize", "hipMemcpy", "hipFree", "hipMalloc", "hipHostMalloc")


def kind_of(fn):
    if "Synchronize" in fn or fn in ("hipMemcpy", "hipMemcpyDtoH", "hipMemcpyHtoD"):
        return "S"
    if "Memcpy" in fn or "Memset" in fn:
        return "c"
    if "Launch" in fn:
        return "l"
    return "."


def main():
    trace_dir, win_path = sys.argv[1], sys.argv[2]
    kern = load(trace_dir, "kernel_trace")
    copies = load(trace_dir, "memory_copy_trace")
    api = load(trace_dir, "hip_api_trace")
    win = json.load(open(win_path))
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r) for r in kern]
    if not ks:
        print("no kernel records")
        return
    # which host clock does the profiler stamp with?  the one whose windows contain the most kernels
    best, clock = -1, None
    for c in win["windows"][0][1]:
        n = 0
        for _, w in win["windows"]:
            a, b = w[c]
            n += sum(1 for s, e, _ in ks if a <= s <= b)
        if n > best:
            best, clock = n, c
    print("# Reads pipeline on the GPU's clock (rocprofv3 --hip-trace --kernel-trace --memory-copy-trace)\n")
    print(f"Profiler clock matched: `{clock}` ({best} of {len(ks)} kernel records fall inside the script's timed windows; the rest "
          "are warm-up).  Times are unions of intervals: concurrent streams are not counted twice.\n")
    n_reads = win.get("reads", 0)
    for seg, w in win["windows"]:
        lo, hi = w[clock]
        wall = (hi - lo) / 1e6
        k_in = [(max(s, lo), min(e, hi), r) for s, e, r in ks if e > lo and s < hi]
        c_in = [(max(int(r["Start_Timestamp"]), lo), min(int(r["End_Timestamp"]), hi), r) for r in copies
                if int(r["End_Timestamp"]) > lo and int(r["Start_Timestamp"]) < hi]
        a_in = [(max(int(r["Start_Timestamp"]), lo), min(int(r["End_Timestamp"]), hi), r) for r in api
                if int(r["End_Timestamp"]) > lo and int(r["Start_Timestamp"]) < hi]
        denom = win.get("single") if seg.startswith("single") else n_reads
        print(f"## {seg}: {wall:.2f} ms wall" + (f" = {wall / denom * 1e3:.1f} us per read" if denom else "") + "\n")
        kb = union_ms([(a, b) for a, b, _ in k_in])
        print(f"* kernels running (any stream): **{kb:.2f} ms = {kb / wall:.2f} of the wall**; sum over kernels "
              f"{sum(b - a for a, b, _ in k_in) / 1e6:.2f} ms in {len(k_in)} launches")
        by_dir = defaultdict(list)
        for a, b, r in c_in:
            by_dir[r.get("Direction", r.get("Kind", "copy"))].append((a, b))
        for d, iv in sorted(by_dir.items()):
            print(f"* copies {d}: {union_ms(iv):.2f} ms in {len(iv)} copies")
        anyb = union_ms([(a, b) for a, b, _ in k_in] + [(a, b) for a, b, _ in c_in])
        print(f"* kernels or copies running: {anyb:.2f} ms = {anyb / wall:.2f} of the wall\n")
        byk = defaultdict(lambda: [0, 0])
        for a, b, r in k_in:
            v = byk[short(r["Kernel_Name"])]
            v[0] += b - a
            v[1] += 1
        print("| kernel | launches | total ms | share of kernel time |\n|---|---|---|---|")
        tot = sum(v[0] for v in byk.values()) or 1
        for name, v in sorted(byk.items(), key=lambda kv: -kv[1][0])[:10]:
            print(f"| `{name}` | {v[1]} | {v[0] / 1e6:.3f} | {v[0] / tot:.2f} |")
        print()
        bya = defaultdict(lambda: [0, 0])
        for a, b, r in a_in:
            v = bya[(r["Thread_Id"], r["Function"])]
            v[0] += b - a
            v[1] += 1
        print("| host thread | HIP call | calls | total ms | share of wall |\n|---|---|---|---|---|")
        for (th, fn), v in sorted(bya.items(), key=lambda kv: -kv[1][0])[:14]:
            print(f"| {th} | `{fn}` | {v[1]} | {v[0] / 1e6:.2f} | {v[0] / 1e6 / wall:.2f} |")
        print()
        gs = sorted(gaps([(a, b) for a, b, _ in k_in], lo, hi), key=lambda g: g[0] - g[1])[:6]
        print("Longest stretches without a running kernel, and the HIP calls in flight during them:\n")
        for a, b in gs:
            if b - a < 20000:
                continue
            during = defaultdict(int)
            for s, e, r in a_in:
                ov = min(e, b) - max(s, a)
                if ov > 0:
                    during[r["Function"]] += ov
            top = ", ".join(f"{fn} {ov / 1e3:.0f} us" for fn, ov in sorted(during.items(), key=lambda kv: -kv[1])[:3])
            print(f"* {(b - a) / 1e3:.0f} us at +{(a - lo) / 1e6:.2f} ms: {top or 'no HIP call (host Python / native code)'}")
        print()
        if seg.startswith("single"):
            continue
        width = 110
        span = max(hi - lo, 1)

        def paint(line, a, b, ch):
            i0 = int((a - lo) / span * width)
            i1 = max(i0 + 1, int((b - lo) / span * width + 0.999))
            for i in range(max(i0, 0), min(i1, width)):
                line[i] = ch

        print("```")
        streams = defaultdict(list)
        for a, b, r in k_in:
            streams[(r.get("Queue_Id", "?"), r.get("Stream_Id", "?"))].append((a, b, r))
        for key in sorted(streams):
            line = [" "] * width
            for a, b, r in streams[key]:
                nm = short(r["Kernel_Name"])
                ch = "L" if "lstm" in nm else "C" if "conv" in nm or "fused" in nm or "front" in nm else "x"
                paint(line, a, b, ch)
            print(f"kernels q{key[0]}/s{key[1]:<4} |{''.join(line)}|")
        for d, iv in sorted(by_dir.items()):
            line = [" "] * width
            for a, b in iv:
                paint(line, a, b, "#")
            print(f"{('copy ' + d)[:18]:18s} |{''.join(line)}|")
        threads = defaultdict(list)
        for a, b, r in a_in:
            threads[r["Thread_Id"]].append((a, b, r))
        for th in sorted(threads):
            line = [" "] * width
            for a, b, r in sorted(threads[th], key=lambda x: kind_of(x[2]["Function"]) == "S"):
                paint(line, a, b, kind_of(r["Function"]))
            print(f"host thread {th[-6:]:6s} |{''.join(line)}|")
        print("```")
        print("C = convolution / fused front kernels, L = LSTM kernels, x = data kernels (motif, geometry, fill, copies by kernel); "
              "# = a DMA copy; host rows: S = inside a synchronising HIP call, c = copy / memset call, l = kernel launch, . = other\n")


if __name__ == "__main__":
    main()
