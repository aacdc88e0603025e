#!/usr/bin/env python
"""Throughput of the signal-mapping refinement kernels (N2) on synthetic reads, inputs resident
in HBM, with the CPU oracle (single-thread C restatement of refine_signal_map_core.pyx) timed on
a sample.  Prints one JSON object; `measure()` is also imported by bench.py."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synth_reads(n_reads, n_bases, seed=0, k=9, center=4):
    rng = np.random.default_rng(seed)
    table = rng.normal(0, 1, 4**k).astype(np.float32)
    reads = []
    for _ in range(n_reads):
        seq = rng.integers(0, 4, n_bases).astype(np.int8)
        dwell = rng.integers(4, 17, n_bases)
        smap = np.concatenate([[0], np.cumsum(dwell)]).astype(np.int64)
        idx = np.zeros(n_bases - k + 1, np.int64)
        for j in range(k):
            idx = idx * 4 + seq[j : n_bases - k + 1 + j]
        lv = np.zeros(n_bases, np.float32)
        lv[center : center + n_bases - k + 1] = table[idx]
        norm = np.repeat(lv, dwell) + 0.3 * rng.standard_normal(smap[-1])
        dacs = np.round(400 + 60 * norm).astype(np.int16)
        jit = smap.copy()
        jit[1:-1] += rng.integers(-3, 4, n_bases - 1)
        jit = np.maximum.accumulate(np.clip(jit, 0, smap[-1]))
        jit[0], jit[-1] = 0, smap[-1]
        reads.append((dacs, jit, seq))
    return table, center, reads


def measure(n_reads=2048, n_bases=5000, algo="dwell_penalty", steps=3, warmup=1, cpu_reads=8, device=0):
    import torch

    from remora_amd import _lib as L
    from remora_amd.engine import get_engine
    from remora_amd.refine_signal_map import SigMapRefiner

    table, center, reads = synth_reads(min(n_reads, 64), n_bases)
    reads = [reads[i % len(reads)] for i in range(n_reads)]  # distinct work per wave, bounded host prep
    ref = SigMapRefiner(_levels_array=table, center_idx=center, scale_iters=0, algo=algo)
    dev = ref._device_refiner(device)
    eng = get_engine(device)
    lib = L.lib()
    sig_off = np.zeros(n_reads + 1, np.int64)
    seq_off = np.zeros(n_reads + 1, np.int64)
    np.cumsum([r[0].size for r in reads], out=sig_off[1:])
    np.cumsum([r[2].size for r in reads], out=seq_off[1:])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda(device)  # noqa: E731
    d_dacs, d_so = t(np.concatenate([r[0] for r in reads])), t(sig_off)
    d_map, d_seq, d_qo = t(np.concatenate([r[1] for r in reads])), t(np.concatenate([r[2] for r in reads])), t(seq_off)
    d_sh, d_sc = t(np.full(n_reads, 400.0)), t(np.full(n_reads, 60.0))
    d_out = torch.empty_like(d_map)
    d_st = torch.zeros(n_reads, dtype=torch.int32, device=d_map.device)
    p = lambda x: ctypes.c_void_p(x.data_ptr())  # noqa: E731

    def step():
        L.check(lib.rmr_refine_signal_maps(dev._h, n_reads, p(d_dacs), p(d_so), p(d_map), p(d_seq), p(d_qo), p(d_sh),
                                           p(d_sc), p(d_out), p(d_st), L.MEM_DEVICE))

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    L.check(lib.rmr_profile_enable(eng.handle, 1))
    L.check(lib.rmr_profile_reset(eng.handle))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    names = [lib.rmr_profile_kernel_name(i).decode() for i in range(lib.rmr_profile_num_kernels())]
    kern = {}
    for nm in ("refine_band", "refine_dp", "refine_dp_rowwise"):
        ms, cnt = ctypes.c_double(), ctypes.c_int64()
        L.check(lib.rmr_profile_get(eng.handle, names.index(nm), ctypes.byref(ms), ctypes.byref(cnt)))
        kern[nm] = {"ms_per_step": ms.value / steps, "launches_per_step": cnt.value / steps}
    L.check(lib.rmr_profile_enable(eng.handle, 0))
    assert int(d_st.abs().sum().item()) == 0
    out = d_out.cpu().numpy()

    # CPU oracle on a sample + parity of that sample
    from oracle import oracle as O

    mo = seq_off + np.arange(n_reads + 1)
    t0 = time.perf_counter()
    cells = 0
    for i in range(cpu_reads):
        d, m, s = reads[i]
        want, err = O.refine_one(d, 400.0, 60.0, m, s, table, center, ref.half_bandwidth, algo, ref.sd_arr)
        assert err is None and np.array_equal(out[mo[i] : mo[i + 1]], want), f"parity broke on read {i}"
    cpu_dt = (time.perf_counter() - t0) / cpu_reads
    total_bases = int(seq_off[-1])
    total_samples = int(sig_off[-1])
    band_cells = int(total_samples * (2 * ref.half_bandwidth + 1))
    dp_ms = kern["refine_dp"]["ms_per_step"]
    return {
        "workload": f"{n_reads} reads x {n_bases} bases, {total_samples / total_bases:.1f} samples/base, 9-mer table, "
                    f"{algo}, half_bandwidth {ref.half_bandwidth}",
        "reads_per_s": n_reads / dt, "bases_per_s": total_bases / dt, "samples_per_s": total_samples / dt,
        "approx_band_cells_per_s_dp_kernel": band_cells / (dp_ms * 1e-3) if dp_ms else None,
        "ms_per_step": dt * 1e3, "kernels": kern,
        "cpu_oracle": {"reads_per_s": 1.0 / cpu_dt, "cores": 1, "sample": f"{cpu_reads} reads of the same batch",
                       "kind": "port"},
        "speedup_vs_1_core": (n_reads / dt) * cpu_dt,
        "mismatches_in_sample": 0,
    }


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2048)
    ap.add_argument("--bases", type=int, default=5000)
    ap.add_argument("--algo", default="dwell_penalty")
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    print(json.dumps(measure(a.reads, a.bases, a.algo, a.steps)))
