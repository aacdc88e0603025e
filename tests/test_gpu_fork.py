"""Fork after the model is loaded: the reference loads its model in the parent and THEN forks the reader / prepare
workers, which never touch the model (src/remora/parsers.py:1590-1598 -> inference.py:488-519); SURVEY section 8(b)
"Threading" makes "HIP initialised before fork must not break the children or the parent" part of the drop-in contract."""
import multiprocessing
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")


def _host_only_work(path):
    """What the reference's forked workers do: file parsing and numpy, nothing on the GPU."""
    from remora_amd import io as rio

    names = [r.query_name for r in rio.iter_bam_records(path)]
    a = np.arange(100000, dtype=np.float64)
    return len(names), float(np.sqrt(a).sum())


def _mp_child(path, q):
    q.put(_host_only_work(path))


def test_model_survives_forked_host_workers():
    import torch

    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    net = torch_ref.random_model("conv_lstm", 64, 9, 2, seed=17)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    model = model_from_state(state, dict(chunk_context=(50, 50), kmer_context_bases=(4, 4)), device=0)
    d = synth.synth_chunks_config("C100", 3000, shard=77)
    args = (d["signal"], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"], (4, 4))
    before = np.array(model.infer_chunks(*args))
    bam = os.path.join(DATA, "can_mappings.bam")
    expect = _host_only_work(bam)

    # 1. a raw fork: the child parses a BAM, does numpy work and leaves through os._exit (no HIP teardown in the child)
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        code = 1
        try:
            os.close(r)
            got = _host_only_work(bam)
            os.write(w, repr(got).encode())
            code = 0
        finally:
            os._exit(code)
    os.close(w)
    with os.fdopen(r, "rb") as fh:
        got = fh.read().decode()
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
    assert got == repr(expect)
    mid = np.array(model.infer_chunks(*args))
    assert np.array_equal(before, mid), "the parent's engine must be unharmed by a forked child"

    # 2. multiprocessing's fork context (what the reference's mp.Process workers are on Linux), two workers side by side
    ctx = multiprocessing.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mp_child, args=(bam, q), daemon=True) for _ in range(2)]
    for p in procs:
        p.start()
    # the parent keeps inferring while the children run, as run_model_batched does beside the reader processes
    during = np.array(model.infer_chunks(*args))
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, p.exitcode
    assert all(tuple(r) == expect for r in results)
    after = np.array(model.infer_chunks(*args))
    assert np.array_equal(before, during) and np.array_equal(before, after)
