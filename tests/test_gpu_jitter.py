"""The synchronisation of the model kernels, tested without needing the right neighbour on the GPU: `make jitter` builds the
same kernels with a pseudo-random per-wave sleep in front of and behind every block barrier and in front of every intra-wave
LDS hand-off (rmr_math.h, RMR_SYNC: a fifth of the waves are held for 0.2-3 us, one in thirty-two for 7-14 us -
longer than any stage of these kernels).  A hand-off through LDS that a barrier does not cover - round 4's LSTM bug was one,
and only showed beside foreign waves - then fails with the kernel alone on the GPU.  Every pipeline must return, call after
call, exactly the bits the shipped build returns."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# 'dtype[:cfg[:arch[:size[:chunks]]]]'; round 6: the streamed-weight kernels (128 / 96 channels, 176 = eleven waves' worth of units
# in the LSTM) and the four-chunk LSTM of small batches (313 chunks)
PIPELINES = ("fp32,bf16,f16,f16x3,bf16x3,bf16x6,fp32:C100:conv_only,bf16:C200,fp32:C200,"
             "fp32:C100:conv_lstm:128,fp32:C100:conv_only:96,fp32:C200:conv_lstm:176,fp32:C100:conv_lstm:64:313")


def test_jittered_barriers_change_no_bit_in_any_pipeline():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    assert os.path.exists(os.path.join(ROOT, "remora_amd", "libremora_hip_jitter.so")), "make -C remora_amd/csrc jitter (build() does)"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_determinism.py"), "--jitter", PIPELINES, "--reps", "12", "--n", "20000",
                          "--json"], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("JSON ")][-1][5:])
    assert [r["pipeline"] for r in res] == PIPELINES.split(","), res
    for r in res:
        assert r["ok"], r
        assert r["runs"] == 12 and r["differing_runs"] == 0, r
