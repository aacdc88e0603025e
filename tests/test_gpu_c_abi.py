"""The C ABI driven by a C program, no Python in the inferring process ("a test calls through the C-ABI").
tests/c/infer_from_c.c is compiled as pedantic C99 against include/remora_hip.h, linked with libremora_hip.so and run on
fixtures exported by tools/export_c_fixture.py from the reference-generated golden model files: rmr_engine_create ->
rmr_model_create -> rmr_infer_chunks(RMR_MEM_HOST) -> rmr_count_labels; its exit code is the verdict."""
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def infer_exe(tmp_path_factory):
    from remora_amd import _lib

    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    exe = str(tmp_path_factory.mktemp("c") / "infer_from_c")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cc = subprocess.run([gcc, "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                         os.path.join(ROOT, "tests", "c", "infer_from_c.c"), "-o", exe, "-L", libdir, "-lremora_hip", "-lm",
                         f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    return exe


@pytest.mark.parametrize("name,dtype", [("model_convlstm_s64_l100_o2", 0), ("model_convlstm_s64_l200_o3", 0),
                                        ("model_conv_s64_l100_o2", 0), ("model_convlstm_s64_l100_o2", 3),
                                        ("model_convlstm_s64_l100_o2", 1), ("model_convlstm_s64_l200_o3", 5)])
def test_c_program_infers_reference_logits(infer_exe, tmp_path, name, dtype):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import export_c_fixture

    fx = str(tmp_path / "fixture.bin")
    n, nw = export_c_fixture.export(os.path.join(ROOT, "tests", "golden", name + ".npz"), fx)
    assert n > 0 and nw > 100000
    run = subprocess.run([infer_exe, fx, str(dtype)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout, run.stderr)
    assert "OK" in run.stdout and f"chunks={n} " in run.stdout and "bad k-mer context refused" in run.stdout


def test_c_program_reports_library_errors(infer_exe, tmp_path):
    """A fixture whose weight count does not fit the descriptor: the program stops with its own message; a truncated
    file likewise — and neither crashes."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import export_c_fixture

    fx = str(tmp_path / "fixture.bin")
    export_c_fixture.export(os.path.join(ROOT, "tests", "golden", "model_convlstm_s64_l100_o2.npz"), fx)
    blob = open(fx, "rb").read()
    open(fx, "wb").write(blob[: len(blob) // 2])
    run = subprocess.run([infer_exe, fx], capture_output=True, text=True, timeout=300)
    assert run.returncode == 3 and "short read" in run.stderr
