"""The product pipeline sharded over GPUs (north_star: "reads shard embarrassingly across the 8 GPUs of one node with RCCL
... only for the final per-label count reduction"): `python -m remora_amd infer from_pod5_and_bam --gpus N` and `validate
from_remora_dataset --gpus N` start one process per GPU; each takes a contiguous share of the alignments / dataset rows,
and the per-label counts (infer) / confusion counts (validate) are all-reduced.  On a 1-GPU box both ranks are pinned to
GPU 0 (REMORA_AMD_FORCE_DEVICE) and the 16..72-byte reduction travels over gloo; kernels, sharding, part files and the
join are the real ones.  Counterparts: src/remora/inference.py:462-641, src/remora/validate.py:190-259 (single-process
in the reference)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")
TWO = {"REMORA_AMD_FORCE_DEVICE": "0", "REMORA_AMD_DIST_BACKEND": "gloo"}


def _remora(*args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-m", "remora_amd", *args], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


def _mint(tmp_path, g):
    import torch

    from oracle import oracle as O
    from oracle import torch_ref

    net = torch_ref.from_state(O.state_from_npz(g))
    pt = str(tmp_path / "model.pt")
    torch.jit.save(torch.jit.script(net), pt, _extra_files={"meta.txt": str(g["meta_txt"])})
    return pt


@pytest.mark.parametrize("prefix,anchored,split", [("can", False, "bytes"), ("mod", True, "bytes"), ("can", False, "scan")])
def test_infer_cli_two_ranks_equal_one_rank(tmp_path, prefix, anchored, split):
    """Two ranks (shares of the BAM by byte range - the default - or by an exact scan of the launcher's) write the
    single-process output byte for byte."""
    from remora_amd import io as rio

    pt = _mint(tmp_path, golden("real_reads_can.npz"))  # the CG 5mC model that calls both halves of configs[0]
    args = ["infer", "from_pod5_and_bam", os.path.join(DATA, f"{prefix}_reads.pod5"), os.path.join(DATA, f"{prefix}_mappings.bam"),
            "--model", pt, "--reads-per-batch", "3"] + (["--reference-anchored"] if anchored else [])
    one, two = str(tmp_path / "one.bam"), str(tmp_path / "two.bam")
    o1 = _remora(*args, "--out-bam", one)
    o2 = _remora(*args, "--out-bam", two, "--gpus", "2", env_extra=dict(TWO, REMORA_AMD_BAM_SHARD=split))
    assert "called 14 reads" in o1 and "called 14 reads" in o2 and "(2 GPUs)" in o2
    tally = lambda o: [ln for ln in o.splitlines() if ln.startswith("calls per label")]
    assert tally(o1) == tally(o2) and len(tally(o1)) == 1
    r1, r2 = list(rio.iter_bam_records(one)), list(rio.iter_bam_records(two))
    assert [r.query_name for r in r1] == [r.query_name for r in r2] and len(r1) == 14
    for a, b in zip(r1, r2):
        assert bytes(a.raw) == bytes(b.raw)  # same record bytes: same MM / ML, same carried-over tags, same order
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]
    total = sum(int(x.split(":")[1]) for x in tally(o1)[0].split(": ", 1)[1].split("; "))
    chunks = sum(len(r.get_tag("ML")) for r in r1)
    assert total == chunks > 500


def test_validate_cli_two_ranks_same_confusion_matrix(tmp_path):
    """`validate from_remora_dataset --gpus 2` on the reference-written core dataset: the confusion matrix, accuracy and
    call count of the summary line equal the single-process run's exactly; the filtered columns likewise (same global
    quantile); the loss agrees to rounding (mean over differently cut batches)."""
    g = golden("call_read_mods_cg_5mc.npz")
    pt = _mint(tmp_path, g)
    ddir = os.path.join(DATA, "core_dataset")
    base = ["validate", "from_remora_dataset", ddir, "--model", pt, "--batch-size", "32"]
    f1 = _remora(*base).strip().splitlines()[-1].split("\t")
    f2 = _remora(*base, "--gpus", "2", env_extra=TWO).strip().splitlines()[-1].split("\t")
    assert f1[0] == f2[0] == "val" and int(f1[6]) == int(f2[6]) == 120
    assert json.loads(f1[4]) == json.loads(f2[4]) and f1[3] == f2[3]
    assert f1[7:10] == f2[7:10]
    assert abs(float(f1[5]) - float(f2[5])) <= 2e-3 * max(1.0, abs(float(f1[5])))
    assert abs(float(f1[10]) - float(f2[10])) <= 1e-9


@pytest.mark.parametrize("procs", [2, 3])
def test_dataset_prepare_cli_ranks_write_the_single_process_dataset(tmp_path, procs):
    """`dataset prepare --gpus 1 --procs-per-gpu P` (every rank extracts its own share of the BAM, rank 0 joins the parts):
    with the shuffle off and no per-read down-sampling (both draw from numpy's global generator, as in the reference) the
    dataset directory holds the same files, byte for byte, as the single-process run's - arrays, metadata, row order."""
    base = ["dataset", "prepare", os.path.join(DATA, "mod_reads.pod5"), os.path.join(DATA, "mod_mappings.bam"), "--mod-base", "m", "5mC",
            "--motif", "CG", "0", "--chunk-context", "50", "50", "--skip-shuffle", "--max-chunks-per-read", "400"]
    one, many = str(tmp_path / "one"), str(tmp_path / "many")
    o1 = _remora(*base, "--output-path", one)
    o2 = _remora(*base, "--output-path", many, "--procs-per-gpu", str(procs), env_extra=TWO)
    tail = lambda o: [ln for ln in o.splitlines() if ln.startswith(("Extracted", "Label distribution"))]
    assert [ln.replace(many, one) for ln in tail(o2)] == tail(o1) and len(tail(o1)) == 2
    assert sorted(os.listdir(one)) == sorted(os.listdir(many))
    for name in os.listdir(one):
        a, b = open(os.path.join(one, name), "rb").read(), open(os.path.join(many, name), "rb").read()
        assert a == b, name
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]
    with pytest.raises(AssertionError, match="not available when the file is split"):
        _remora(*base, "--output-path", str(tmp_path / "x"), "--procs-per-gpu", "2", "--num-reads", "5", env_extra=TWO)
