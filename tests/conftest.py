import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the HIP library is built in-tree (git-ignored): build it if this checkout has none yet
    lib = os.path.join(ROOT, "remora_amd", "libremora_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess

        subprocess.call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "remora_amd", "csrc")])


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def code_to_onehot(code):
    """[n, K, L] int8 base code (-1 = none) -> dense f32 [n, 4K, L] one-hot."""
    n, K, L = code.shape
    return (code[:, :, None, :] == np.arange(4, dtype=np.int8)[None, None, :, None]).astype(
        np.float32
    ).reshape(n, 4 * K, L)
