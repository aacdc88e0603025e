"""Several processes on ONE GPU (`--procs-per-gpu`, two bench ranks with --force-device, or simply a neighbour's job): every
pipeline has to return the same bits as it does alone.  Round 4 found two ways it did not - the LSTM kernels overwrote the x_0
tile while a delayed wave was still reading it (a missing barrier that only showed when foreign waves held a wave up), and the
packed-fp32 VALU signal producers returned damaged activations beside a process that kept the bf16 matrix cores busy
(rmr_math.h, pk_fma; profiles/NOTES_r04.md).  Before those fixes this mix failed in 10-40 % of the fp32 / bf16x6 calls and
about 1 % of the bf16 / f16 ones."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pipelines_sharing_a_gpu_stay_bit_stable():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    mix = "fp32,bf16x6,bf16,f16,f16x3"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_determinism.py"), "--mix", mix, "--reps", "150", "--n", "20000", "--json"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("JSON ")][-1][5:])
    assert [r["process"] for r in res] == mix.split(",")
    for r in res:
        assert r["failed"] is False, r
        assert r["differing_runs"] == 0, r
