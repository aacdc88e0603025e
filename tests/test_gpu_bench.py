"""bench.py as a multi-rank program, and the C-ABI collective — on one GPU.

The 8-GPU scaling run belongs to the driver; what can be proven on a 1-GPU box is that (a) `bench.py --gpus 2` launches
two ranks by itself, shards the data, and its one collective reduces the per-rank label counts exactly (two ranks on
the same GPU over gloo: the data path is the real one, only the transport of the 16-byte reduction differs), (b) a
world size that differs from --gpus is refused, (c) rmr_allreduce_counts runs through RCCL (a communicator of one rank)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ["--steps", "2", "--warmup", "1", "--chunks", "20000", "--no-cpu-baseline", "--no-encode", "--no-reads", "--no-others",
        "--no-refine"]


LINE_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config", "roofline", "parity", "details"}


def _bench(extra, env_extra=None, expect_rc=0):
    """Run bench.py; returns (full report = bench_details.json merged under the parsed stdout line, process).  The stdout
    line is what the driver records: it must be the ONLY thing on stdout, parse as JSON and stay below 4 KB."""
    import tempfile

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    with tempfile.TemporaryDirectory() as td:
        det = os.path.join(td, "details.json")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--details", det] + extra, capture_output=True, text=True,
                           env=env, timeout=900)
        assert p.returncode == expect_rc, (p.returncode, p.stderr[-3000:])
        if expect_rc != 0:
            return None, p
        out = p.stdout.strip().splitlines()
        assert len(out) == 1 and len(out[0]) < 4096, (len(out), [len(x) for x in out])
        line = json.loads(out[0])
        assert LINE_KEYS <= set(line), sorted(LINE_KEYS - set(line))
        assert {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
        full = json.load(open(det))
        for k in ("value", "n_gpus", "steps", "scaling", "ms_per_step"):
            assert full[k] == line[k], k
        return full, p


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_bench_two_ranks_shard_and_reduce_exactly(dtype):
    two, _ = _bench(["--gpus", "2", "--force-device", "0", "--dist-backend", "gloo", "--dtype", dtype] + FAST)
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["scaling"] == "weak"
    assert two["config"]["chunks_per_step_all_gpus"] == 40000
    assert sum(two["label_counts"]) == 2 * 20000 * 2
    per_rank = np.asarray(two["label_counts_per_rank"])
    assert per_rank.shape == (2, 2) and np.array_equal(per_rank.sum(0), two["label_counts"])
    assert two["collective"]["backend"] == "gloo" and len(two["collective"]["allreduce_ms_per_rank"]) == 2
    assert all(int(r.sum()) == 20000 * 2 for r in per_rank)
    for r in range(2):  # each rank's tally = a single-rank run over that rank's shard of the data
        one, _ = _bench(["--gpus", "1", "--shard-base", str(r), "--dtype", dtype] + FAST)
        assert one["n_gpus"] == 1 and one["label_counts"] == [int(x) for x in per_rank[r]], (r, one["label_counts"], per_rank)
    assert per_rank[0].tolist() != per_rank[1].tolist()  # the ranks really worked on different chunks


def test_bench_strong_sharding_partitions_the_same_data_set():
    """configs[3] shape (bf16, one data set sharded over the ranks) at a small size: the global label tally does not
    depend on the number of ranks."""
    args = ["--workload", "convlstm_c100_bf16_10m", "--steps", "1", "--warmup", "1", "--chunks", "30001", "--no-cpu-baseline",
            "--no-encode", "--no-reads", "--no-others", "--no-refine"]
    one, _ = _bench(["--gpus", "1"] + args)
    two, _ = _bench(["--gpus", "2", "--force-device", "0", "--dist-backend", "gloo"] + args)
    assert one["scaling"] == two["scaling"] == "strong" and two["n_gpus"] == 2
    assert sum(one["label_counts"]) == 30001 and one["label_counts"] == two["label_counts"]
    assert [sum(r) for r in two["label_counts_per_rank"]] == [15001, 15000]


def test_bench_eight_ranks_on_one_gpu_weak():
    """The driver's 8-GPU form rehearsed on what a 1-GPU lease offers: EIGHT ranks (torch.distributed.run, gloo, every rank
    on device 0) of the headline workload.  World-8 rank arithmetic of the weak branch (rank r = data blocks of rank r),
    the per-rank tallies, their sum, the placement record of every rank and the exit code."""
    n = 6000
    args = ["--steps", "2", "--warmup", "1", "--chunks", str(n), "--no-cpu-baseline", "--no-encode", "--no-reads", "--no-others", "--no-refine"]
    eight, p = _bench(["--gpus", "8", "--force-device", "0", "--dist-backend", "gloo", "--logits-hash"] + args)
    assert eight["n_gpus"] == 8 and eight["scaling"] == "weak" and eight["config"]["chunks_per_step_all_gpus"] == 8 * n
    per_rank = np.asarray(eight["label_counts_per_rank"])
    assert per_rank.shape == (8, 2) and np.array_equal(per_rank.sum(0), eight["label_counts"])
    assert all(int(r.sum()) == 2 * n for r in per_rank) and sum(eight["label_counts"]) == 8 * 2 * n
    assert len({tuple(r) for r in per_rank.tolist()}) > 4  # the ranks worked on different chunks
    coll = eight["collective"]
    assert coll["backend"] == "gloo" and len(coll["allreduce_ms_per_rank"]) == 8 and len(coll["rccl_init_ms_per_rank"]) == 8
    assert [pl["rank"] for pl in coll["placement_per_rank"]] == list(range(8)) and all(pl["device"] == 0 for pl in coll["placement_per_rank"])
    assert eight["value"] == pytest.approx(8 * n * 2 / (eight["ms_per_step"] * 2e-3), rel=1e-6)
    for r in (0, 5, 7):  # a rank's tally = a single-rank run over that rank's blocks of the data set
        one, _ = _bench(["--gpus", "1", "--shard-base", str(r)] + args)
        assert one["label_counts"] == [int(x) for x in per_rank[r]], (r, one["label_counts"], per_rank[r])


def test_bench_eight_ranks_on_one_gpu_strong_same_logits_as_one_rank():
    """configs[3] (bf16, ONE data set cut into contiguous ranges, dist.shard_range) with eight ranks: the ranges tile the
    data set, and the logits of all ranks in rank order are the single-rank run's, bit for bit."""
    total = 80_003
    args = ["--workload", "convlstm_c100_bf16_10m", "--steps", "1", "--warmup", "1", "--chunks", str(total), "--no-cpu-baseline",
            "--no-encode", "--no-reads", "--no-others", "--no-refine", "--logits-hash"]
    one, _ = _bench(["--gpus", "1"] + args)
    eight, _ = _bench(["--gpus", "8", "--force-device", "0", "--dist-backend", "gloo"] + args)
    assert eight["scaling"] == "strong" and eight["n_gpus"] == 8
    assert [sum(r) for r in eight["label_counts_per_rank"]] == [10001, 10001, 10001] + [10000] * 5
    assert one["label_counts"] == eight["label_counts"] and sum(one["label_counts"]) == total
    assert one["logits_sha256"] and one["logits_sha256"] == eight["logits_sha256"]


def test_bench_falls_back_to_gloo_when_rccl_cannot_reduce():
    """The default backend (nccl = RCCL) with a first all-reduce that fails (REMORA_AMD_DIST_FAIL_FIRST: RCCL itself refuses
    two ranks on one device, which a 1-GPU box cannot get past): every rank agrees over the gloo side channel, the run's
    collectives move there, the line is printed, `collective.backend` says what happened, rc 0."""
    two, p = _bench(["--gpus", "2", "--force-device", "0"] + FAST, env_extra={"REMORA_AMD_DIST_FAIL_FIRST": "1"})
    assert two["n_gpus"] == 2 and sum(two["label_counts"]) == 2 * 20000 * 2
    assert two["collective"]["backend"].startswith("gloo (RCCL failed: rank 0: RuntimeError"), two["collective"]["backend"]
    assert np.array_equal(np.asarray(two["label_counts_per_rank"]).sum(0), two["label_counts"])
    assert "go over gloo instead" in p.stderr


def test_bench_refuses_a_world_that_is_not_gpus():
    _, p = _bench(["--gpus", "2"] + FAST, env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, expect_rc=2)
    assert "refusing" in p.stderr and not p.stdout.strip()


def test_cabi_allreduce_counts_through_rccl():
    """rmr_comm_unique_id -> rmr_comm_init (1 rank) -> rmr_allreduce_counts on a device and on a host buffer -> destroy:
    the library's own RCCL path (dlopen, communicator on the engine's device and stream) end to end; identity for one rank.
    Before a communicator exists the call is the documented no-op."""
    import torch

    from remora_amd import dist as rdist
    from remora_amd import _lib as L
    from remora_amd.engine import get_engine

    eng = get_engine(0)
    c = torch.tensor([5, 7, 11], dtype=torch.int64, device="cuda:0")
    rdist.cabi_allreduce_counts(eng, c)
    assert c.tolist() == [5, 7, 11]
    assert rdist.init_cabi_comm(eng, 0, 1)
    try:
        rdist.cabi_allreduce_counts(eng, c)
        assert c.tolist() == [5, 7, 11]
        h = np.array([1, 2, 3, 4], np.int64)
        rdist.cabi_allreduce_counts(eng, h)
        assert h.tolist() == [1, 2, 3, 4]
        with pytest.raises(Exception, match="already has a communicator"):
            rdist.init_cabi_comm(eng, 0, 1)
    finally:
        L.check(L.lib().rmr_comm_destroy(eng.handle))
    chk = rdist.cabi_allreduce_check(eng, np.array([[3, 4]], np.int64), 0, 1)
    assert chk["status"] == "ok" and chk["got"] == [3, 4], chk


_ONE_RANK_RCCL = r'''
import json, os, sys
import numpy as np
import torch
from remora_amd import dist as rdist

rank, world, local = rdist.init_process_group("nccl", timeout_s=120)
import torch.distributed as dist
out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "first_ms": rdist.first_collective_ms()}
out["floats"] = rdist.allgather_floats([1.5, 2.5]).tolist()
out["np_counts"] = rdist.allreduce_counts(np.array([3, 4, 5], np.int64)).tolist()
c = torch.tensor([7, 8], dtype=torch.int64, device="cuda:0")
out["cuda_counts"] = rdist.allreduce_counts(c).tolist()
out["max"] = rdist.allreduce_max_float(2.25)
out["rows"] = rdist.allgather_counts(c).tolist()
out["objects"] = rdist.gather_objects({"ok": 3, None: 1})[0][None]
out["arrays"] = rdist.gather_arrays(np.arange(6, dtype=np.float32).reshape(3, 2)).tolist()
out["ints"] = rdist.gather_arrays(np.arange(4, dtype=np.int64)).tolist()
rdist.barrier()
# the sharded validation's global metrics use the same helpers
from remora_amd.validate import ValidationLogger
probs = np.array([[0.9, 0.1], [0.2, 0.8], [0.6, 0.4]], np.float32)
m = ValidationLogger._global_metrics(probs, np.array([0, 1, 1]), np.array([0.1, 0.2, 0.9]), 0.1)
out["acc"], out["calls"] = float(m.acc), int(m.num_calls)
dist.destroy_process_group()
print("\nRESULT " + json.dumps(out), flush=True)
'''


def test_torch_distributed_helpers_through_one_rank_rccl():
    """Every dist.py helper the bench and the sharded product pipeline call, through torch.distributed's nccl (= RCCL)
    backend with a process group of ONE rank (REMORA_AMD_DIST_SINGLE=1 lifts the single-process short cuts): device
    placement of the tensors, the dtypes RCCL is asked to move (int64, float64, float32), all_gather_object and barrier on
    the GPU backend.  What only more ranks can show - the ring over xGMI - stays the driver's 8-GPU run."""
    env = dict(os.environ, REMORA_AMD_DIST_SINGLE="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    env.pop("MASTER_PORT", None)
    p = subprocess.run([sys.executable, "-c", _ONE_RANK_RCCL], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    at = p.stdout.rfind("RESULT {")  # (RCCL's version banner shares stdout and does not always end its last line)
    assert at >= 0, (p.stdout[-2000:], p.stderr[-2000:])
    out = json.loads(p.stdout[at + 7:].splitlines()[0])
    assert out["backend"] == "nccl" and out["world"] == 1 and out["first_ms"] > 0
    assert out["floats"] == [[1.5, 2.5]] and out["np_counts"] == [3, 4, 5] and out["cuda_counts"] == [7, 8]
    assert out["max"] == 2.25 and out["rows"] == [[7, 8]] and out["objects"] == 1
    assert out["arrays"] == [[0.0, 1.0], [2.0, 3.0], [4.0, 5.0]] and out["ints"] == [0, 1, 2, 3]
    assert abs(out["acc"] - 2 / 3) < 1e-12 and out["calls"] == 3


def test_bench_one_rank_through_rccl_reproduces_the_plain_run():
    """bench.py itself with its process group built over nccl for ONE rank (REMORA_AMD_DIST_SINGLE=1): the timed region's
    all-reduce of the device-resident label counts, the max-over-ranks clock and the per-rank gather run through RCCL and
    leave the numbers of the plain single-process run."""
    plain, _ = _bench(["--gpus", "1"] + FAST)
    assert plain["allreduce_counts_c_abi"]["status"] == "ok", plain["allreduce_counts_c_abi"]
    forced, p = _bench(["--gpus", "1"] + FAST, env_extra={"REMORA_AMD_DIST_SINGLE": "1", "RANK": "0", "WORLD_SIZE": "1",
                                                         "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1"})
    assert forced["label_counts"] == plain["label_counts"] and forced["label_counts_per_rank"] == plain["label_counts_per_rank"]
    assert forced["n_gpus"] == 1 and forced["value"] > 0
