#!/usr/bin/env python3
"""The reads paths timed as medians inside ONE process (the boxes of the pool differ by 20 % and so do consecutive calls):
single-read call_read_mods over six rounds of 64 reads, batched call_reads_mods over several calls of 2048 reads.  Knobs are
environment variables (RMR_READS_STAGERS, RMR_PACK_THREADS, RMR_READS_SUBBATCH): one process per setting.
    python tools/ab_reads.py [--dtypes fp32,bf16] [--calls 7]"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtypes", default="fp32,bf16")
    ap.add_argument("--calls", type=int, default=7)
    args = ap.parse_args()
    import torch

    from remora_amd import synth
    from remora_amd.data_chunks import RemoraRead
    from remora_amd.inference import call_read_mods, call_reads_mods
    from remora_amd.model_util import model_from_state

    st = synth.synth_state()
    md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"],
              can_base="C", base_start_justify=False, offset=0, sig_map_refiner=None, reverse_signal=False, pa_scaling=None)
    rs = []
    for i in range(2048):
        r = synth.synth_read(5000, idx=i)
        rs.append(RemoraRead(dacs=r["dacs"], shift=r["shift"], scale=r["scale"], seq_to_sig_map=r["seq_to_sig_map"], int_seq=r["int_seq"]))
    print(f"stagers {os.environ.get('RMR_READS_STAGERS', '2')}, pack threads {os.environ.get('RMR_PACK_THREADS', '8')}")
    for dt in args.dtypes.split(","):
        model = model_from_state(st, md, device=0, dtype=dt)
        for r in rs[:16]:
            call_read_mods(r, model, md)
        res = []
        for rnd in range(6):
            t = time.perf_counter()
            for r in rs[64 * rnd : 64 * rnd + 64]:
                call_read_mods(r, model, md)
            res.append((time.perf_counter() - t) / 64 * 1e6)
        print(f"{dt} single read: {statistics.median(res):.0f} us (best {min(res):.0f}) = {1e3 / statistics.median(res):.2f} k reads/s")
        call_reads_mods(rs, model, md)
        ts = []
        for _ in range(args.calls):
            torch.cuda.synchronize()
            t = time.perf_counter()
            call_reads_mods(rs, model, md)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        print(f"{dt} batched 2048 reads: median {statistics.median(ts) * 1e3:.2f} ms = {2048 / statistics.median(ts) / 1e3:.1f} k reads/s, best "
              f"{min(ts) * 1e3:.2f} ms = {2048 / min(ts) / 1e3:.1f} k reads/s, all {[round(x * 1e3, 1) for x in ts]}")
        del model


if __name__ == "__main__":
    main()
