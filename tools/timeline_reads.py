#!/usr/bin/env python3
"""Host-side timeline of the reads pipeline (inference.call_reads_mods and the single-read call_read_mods): which thread
spends how long in which stage of a batch, as a table and an ASCII chart.  The stages are timed by wrapping the functions
the pipeline calls (nothing in the product is instrumented); run it alone for the host view, or under
`rocprofv3 --hip-trace --kernel-trace --memory-copy-trace` for the GPU view of the same run (tools/timeline_summary.py
cuts that trace into the same segments: the script sleeps 0.4 s between them).

    python tools/timeline_reads.py [--reads 2048] [--dtypes fp32,bf16] [--single 64] [--out gpurun_out/timeline_host.md]
"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

EVENTS = []  # (segment, thread name, stage, t0, t1)
SEG = ["warmup"]
_LOCK = threading.Lock()
WINDOWS = []  # (segment, {clock name: (ns at start, ns at end)}): the profiler's clock is one of these
_CLOCKS = {"boottime": time.CLOCK_BOOTTIME, "monotonic": time.CLOCK_MONOTONIC, "monotonic_raw": time.CLOCK_MONOTONIC_RAW,
           "realtime": time.CLOCK_REALTIME}


def clocks_now():
    return {k: time.clock_gettime_ns(v) for k, v in _CLOCKS.items()}


class window:
    """Records the segment's start and end on every clock the profiler might stamp its records with."""

    def __init__(self, seg):
        self.seg = seg

    def __enter__(self):
        self.a = clocks_now()

    def __exit__(self, *exc):
        b = clocks_now()
        WINDOWS.append((self.seg, {k: (self.a[k], b[k]) for k in b}))


def timed(stage, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            t1 = time.perf_counter()
            with _LOCK:
                EVENTS.append((SEG[0], threading.current_thread().name, stage, t0, t1))

    wrapper.__wrapped__ = fn
    return wrapper


def instrument():
    from remora_amd import data_chunks as dc
    from remora_amd import engine as eng
    from remora_amd import inference as inf
    from remora_amd import util

    dc.DeviceReads.__init__ = timed("stage: pack + upload", dc.DeviceReads.__init__)
    dc.DeviceReads.wait_ready = timed("wait upload", dc.DeviceReads.wait_ready)
    dc.DeviceReads.motif_focus_bases = timed("motif scan", dc.DeviceReads.motif_focus_bases)
    dc._extract_device = timed("extract (geometry + fill)", dc._extract_device)
    dc.device_to_numpy = timed("D2H", dc.device_to_numpy)
    if hasattr(dc, "device_to_pinned_async"):
        dc.device_to_pinned_async = timed("D2H queued", dc.device_to_pinned_async)
    eng.HipModel.infer_chunks = timed("infer_chunks (launch)", eng.HipModel.infer_chunks)
    eng.Engine.wait_submitted = timed("wait kernels", eng.Engine.wait_submitted)
    if hasattr(inf, "_native_call_reads"):
        inf._native_call_reads = timed("rmr_call_reads", inf._native_call_reads)
    if hasattr(inf, "_native_call_read"):
        inf._native_call_read = timed("rmr_call_read", inf._native_call_read)
    util.find_focus_bases_in_int_sequence = timed("host motif search", util.find_focus_bases_in_int_sequence)
    dc.util.find_focus_bases_in_int_sequence = util.find_focus_bases_in_int_sequence


def chart(seg, t_lo, t_hi, width=100):
    rows = {}
    for s, th, stage, a, b in EVENTS:
        if s != seg:
            continue
        rows.setdefault(th, []).append((stage, a, b))
    letters = {}
    out = []
    span = max(t_hi - t_lo, 1e-9)
    for th in sorted(rows):
        line = [" "] * width
        for stage, a, b in rows[th]:
            ch = letters.setdefault(stage, "PwmxDiKNRh"[len(letters) % 10])
            i0 = int((a - t_lo) / span * width)
            i1 = max(i0 + 1, int((b - t_lo) / span * width))
            for i in range(max(i0, 0), min(i1, width)):
                line[i] = ch
        out.append(f"{th[:14]:14s} |{''.join(line)}|")
    out.append("legend: " + ", ".join(f"{c} = {s}" for s, c in letters.items()))
    return "\n".join(out)


def table(seg, wall, n_reads):
    agg = {}
    for s, th, stage, a, b in EVENTS:
        if s == seg:
            k = (th, stage)
            v = agg.setdefault(k, [0.0, 0])
            v[0] += b - a
            v[1] += 1
    lines = ["| thread | stage | calls | total ms | share of wall | us per read |", "|---|---|---|---|---|---|"]
    for (th, stage), (tot, n) in sorted(agg.items()):
        lines.append(f"| {th} | {stage} | {n} | {tot * 1e3:.2f} | {tot / wall:.2f} | {tot / n_reads * 1e6:.1f} |")
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2048)
    ap.add_argument("--dtypes", default="fp32,bf16")
    ap.add_argument("--single", type=int, default=64)
    ap.add_argument("--out", default="gpurun_out/timeline_host.md")
    ap.add_argument("--no-instrument", action="store_true")
    args = ap.parse_args()
    import torch

    from remora_amd import synth
    from remora_amd.data_chunks import RemoraRead
    from remora_amd.model_util import model_from_state

    if not args.no_instrument:
        instrument()
    from remora_amd import inference as inf

    st = synth.synth_state()
    md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"],
              can_base="C", base_start_justify=False, offset=0, sig_map_refiner=None, reverse_signal=False, pa_scaling=None)
    rs = []
    for i in range(args.reads):
        r = synth.synth_read(5000, idx=i)
        rs.append(RemoraRead(dacs=r["dacs"], shift=r["shift"], scale=r["scale"], seq_to_sig_map=r["seq_to_sig_map"],
                             int_seq=r["int_seq"], read_id=f"syn{i}"))
    doc = ["# Host-side timeline of the reads pipeline (tools/timeline_reads.py)", "",
           f"{args.reads} synthetic reads of 5000 bases (~312 CG chunks each); wall clocks around each call, stages timed by "
           "wrapping the functions the pipeline calls.", ""]
    segments = ["start-up (imports, read synthesis)"]
    for dt in args.dtypes.split(","):
        time.sleep(0.4)
        SEG[0] = f"warmup batched {dt}"
        segments.append(SEG[0])
        model = model_from_state(st, md, device=0, dtype=dt)
        res = inf.call_reads_mods(rs, model, md)
        nchunks = sum(x[2].size for x in res)
        inf.call_reads_mods(rs, model, md)
        torch.cuda.synchronize()
        for rep in range(2):
            time.sleep(0.4)
            seg = f"batched {dt} #{rep}"
            SEG[0] = seg
            with window(seg):
                t0 = time.perf_counter()
                inf.call_reads_mods(rs, model, md)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            segments.append(seg)
            doc += [f"## call_reads_mods, {dt} model, call {rep}: {(t1 - t0) * 1e3:.2f} ms = {args.reads / (t1 - t0) / 1e3:.1f} k reads/s "
                    f"= {nchunks / (t1 - t0) / 1e6:.1f} M chunks/s", "", "```", chart(seg, t0, t1), "```", "", table(seg, t1 - t0, args.reads), ""]
        if args.single:
            time.sleep(0.4)
            SEG[0] = f"warmup single {dt}"
            segments.append(SEG[0])
            for r in rs[:8]:
                inf.call_read_mods(r, model, md)
            torch.cuda.synchronize()
            time.sleep(0.4)
            seg = f"single {dt}"
            SEG[0] = seg
            with window(seg):
                t0 = time.perf_counter()
                for r in rs[: args.single]:
                    inf.call_read_mods(r, model, md)
                t1 = time.perf_counter()
            segments.append(seg)
            doc += [f"## call_read_mods one read at a time, {dt} model: {(t1 - t0) / args.single * 1e6:.0f} us per read = "
                    f"{args.single / (t1 - t0) / 1e3:.2f} k reads/s", "", table(seg, t1 - t0, args.single), ""]
        del model
    time.sleep(0.4)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(doc) + "\n")
    import json

    with open(os.path.splitext(args.out)[0] + "_windows.json", "w") as f:
        json.dump({"windows": WINDOWS, "reads": args.reads, "single": args.single}, f)
    print("\n".join(x for x in doc if x.startswith("## ")))


if __name__ == "__main__":
    main()
