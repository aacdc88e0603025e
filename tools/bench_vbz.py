#!/usr/bin/env python
"""Throughput of the GPU VBZ decode (N1) on synthetic signal rows resident in HBM, with the CPU oracle (C
restatement, one core) timed on a sample and checked against the GPU output.  `measure()` is imported by bench.py."""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def svb16_encode(sig):
    sig = np.asarray(sig, np.int16)
    d = np.diff(np.concatenate([[0], sig.astype(np.int64)])).astype(np.int16)
    zz = ((d.astype(np.int32) << 1) ^ (d.astype(np.int32) >> 15)).astype(np.uint32) & 0xFFFF
    two = zz > 0xFF
    keys = np.packbits(two, bitorder="little")
    data = np.empty(sig.size + int(two.sum()), np.uint8)
    offs = np.cumsum(two + 1) - (two + 1)
    data[offs] = zz & 0xFF
    data[offs[two] + 1] = zz[two] >> 8
    return np.concatenate([keys, data])


def measure(n_rows=4096, row_samples=102400, steps=5, warmup=1, cpu_rows=16, device=0):
    import torch

    from oracle import oracle as O
    from remora_amd import _lib as L
    from remora_amd.engine import get_engine

    rng = np.random.default_rng(3)
    uniq = []
    for _ in range(16):  # nanopore-like: slow level changes + noise, ~1.15 B per sample after svb16
        lv = np.repeat(rng.integers(350, 650, row_samples // 10 + 1), 10)[:row_samples]
        uniq.append(svb16_encode((lv + rng.normal(0, 12, row_samples)).astype(np.int16)))
    rows = [uniq[i % 16] for i in range(n_rows)]
    row_off = np.zeros(n_rows + 1, np.int64)
    np.cumsum([r.size for r in rows], out=row_off[1:])
    buf = np.zeros(int(row_off[-1]) + 16, np.uint8)
    for r, st in zip(rows, row_off):
        buf[st : st + r.size] = r
    rn = np.full(n_rows, row_samples, np.int32)
    eng = get_engine(device)
    lib = L.lib()
    dev = eng.torch_device
    d_buf, d_ro, d_rn = (torch.from_numpy(a).to(dev) for a in (buf, row_off, rn))
    d_out = torch.empty(n_rows * row_samples, dtype=torch.int16, device=dev)

    def step():
        L.check(lib.rmr_vbz_decode(eng.handle, d_buf.data_ptr(), d_ro.data_ptr(), d_rn.data_ptr(), n_rows,
                                   d_out.data_ptr(), L.MEM_DEVICE))

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
    ms, cnt = ctypes.c_double(), ctypes.c_int64()
    L.check(lib.rmr_profile_get(eng.handle, names.index("vbz_decode"), ctypes.byref(ms), ctypes.byref(cnt)))
    L.check(lib.rmr_profile_enable(eng.handle, 0))
    kern_ms = ms.value / max(cnt.value, 1)
    got = d_out[: cpu_rows * row_samples].cpu().numpy()
    t0 = time.perf_counter()
    for i in range(cpu_rows):
        want = O.vbz_decode(rows[i].tobytes(), row_samples)
        assert np.array_equal(got[i * row_samples : (i + 1) * row_samples], want), f"parity broke on row {i}"
    cpu_dt = (time.perf_counter() - t0) / cpu_rows
    samples = n_rows * row_samples
    alg_bytes = int(row_off[-1]) + 2 * samples
    return {
        "workload": f"{n_rows} signal rows x {row_samples} samples, {row_off[-1] / samples:.2f} compressed B/sample (svb16 layer)",
        "samples_per_s": samples / dt, "ms_per_step": dt * 1e3,
        "roofline": {"kernel": "vbz_decode", "bound": "hbm", "achieved": alg_bytes / (kern_ms * 1e-3) / 1e9, "peak": 8000.0,
                     "unit": "GB/s", "frac": alg_bytes / (kern_ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes": alg_bytes,
                     "avg_launch_ms": kern_ms},
        "cpu_oracle": {"samples_per_s": row_samples / cpu_dt, "cores": 1, "kind": "port", "sample": f"{cpu_rows} rows of the same batch"},
        "speedup_vs_1_core": (samples / dt) * cpu_dt / row_samples,
    }


if __name__ == "__main__":
    print(json.dumps(measure()))
