"""GPU parity of the fp32 sig_conv3 / seq_conv2 kernels with their producers folded in (remora_amd/csrc/k_conv_front.hip):
bit-identical to the separate front + convolution kernels of the same library (RMR_CONV_FRONT=0), and within the fp32
tolerance (1e-4 on logits) of the CPU restatement of the reference network."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _env(name, value, fn):
    old = os.environ.get(name)
    os.environ[name] = value
    try:
        return fn()
    finally:
        if old is None:
            del os.environ[name]
        else:
            os.environ[name] = old


def _unfolded(fn):
    os.environ["RMR_CONV_FRONT"] = "0"
    try:
        return fn()
    finally:
        del os.environ["RMR_CONV_FRONT"]


@pytest.mark.parametrize("cfg,num_out", [("C100", 2), ("C200", 3)])
def test_folded_producers_equal_separate_kernels_and_oracle(cfg, num_out):
    import torch

    from oracle import oracle as O
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.engine import get_engine
    from remora_amd.model_util import model_from_state
    from test_gpu_fused import _awkward

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    net = torch_ref.random_model("conv_lstm", 64, 9, num_out, seed=11)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    cc = synth.CONFIGS[cfg][0]
    model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=(4, 4)), device=0, dtype="fp32")
    eng = get_engine(0)
    rng = np.random.default_rng(3)
    for n in (1, 3, 4, 5, 37, 1000, 4099):  # not multiples of the chunks per block iteration
        d = synth.synth_chunks_config(cfg, n, shard=200 + n)
        seqs, maps, lens = _awkward(d, rng) if n >= 37 else (d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])
        eng.profile_reset()
        eng.profile_enable(True)
        out = model.infer_chunks(d["signal"], seqs, maps, lens, (4, 4))
        eng.profile_enable(False)
        prof = eng.profile()
        assert "sig3_front" in prof and "seq2_front" in prof and "front_seq" not in prof, prof.keys()
        # bit identity with the separate kernels is a property of the DIRECT forms (RMR_WINOGRAD=0 on both sides); the shipped path has
        # sig_conv3 / seq_conv2 / merge_conv1 in Winograd form (tests/test_gpu_wino.py) and stays within rounding of them
        direct = _env("RMR_WINOGRAD", "0", lambda: model.infer_chunks(d["signal"], seqs, maps, lens, (4, 4)))
        out_u = _env("RMR_WINOGRAD", "0", lambda: _unfolded(lambda: model.infer_chunks(d["signal"], seqs, maps, lens, (4, 4))))
        assert np.array_equal(direct, out_u), (cfg, n, float(np.abs(direct - out_u).max()))
        assert np.abs(out - direct).max() <= 2e-5, (cfg, n, float(np.abs(out - direct).max()))
        enc = O.compute_encoded_kmer_batch(4, 4, seqs, maps, lens)
        with torch.no_grad():
            ref = net(torch.from_numpy(d["signal"]), torch.from_numpy(enc)).numpy()
        assert np.abs(out - ref).max() <= 1e-4, (cfg, n, float(np.abs(out - ref).max()))


def test_folded_producers_full_size_properties():
    """1M chunks: deterministic, independent of batch position, exact label tally, equal to the separate kernels."""
    import torch

    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    n = 1_000_000
    model = model_from_state(synth.synth_state("conv_lstm", 64, 9, 2, seed=0), dict(chunk_context=(50, 50), kmer_context_bases=(4, 4)),
                             device=0, dtype="fp32")
    d = synth.synth_chunks_config("C100", n)
    dev = [torch.from_numpy(d[k]).cuda() for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
    counts = torch.zeros(2, dtype=torch.int64, device="cuda")
    out = model.infer_chunks(*dev, (4, 4), label_counts=counts)
    assert bool(torch.isfinite(out).all())
    assert torch.equal(counts, torch.bincount(out.argmax(dim=1), minlength=2)) and int(counts.sum()) == n
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    outp = model.infer_chunks(*[t[perm].contiguous() for t in dev], (4, 4))
    assert torch.equal(outp, out[perm]), "result of a chunk depends on its batch position"
    direct = _env("RMR_WINOGRAD", "0", lambda: model.infer_chunks(*dev, (4, 4)))
    out_u = _env("RMR_WINOGRAD", "0", lambda: _unfolded(lambda: model.infer_chunks(*dev, (4, 4))))
    assert torch.equal(direct, out_u)
    assert float((out - direct).abs().max()) <= 5e-5


@pytest.mark.parametrize("arch,cfg,num_out", [("conv_only", "C100", 2), ("conv_only", "C100", 3), ("conv_lstm", "C100", 2), ("conv_lstm", "C200", 3)])
def test_signal_fold_with_matrix_core_producer_is_bit_identical(arch, cfg, num_out):
    """sig3_front_mfma_kernel (round 3): sig_conv2 as one fp32 MFMA per tap inside the staging of sig_conv3 — the shipped
    path of both architectures (RMR_SIG3_MFMA=0: the VALU producers).  An fp32 MFMA is a k-ordered fmaf
    chain, so the logits must equal those of the separate VALU front kernel + conv_mfma bit for bit, for ragged batch
    sizes; and they match the CPU restatement of the reference network within 1e-4."""
    import torch

    from oracle import oracle as O
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.engine import get_engine
    from remora_amd.model_util import model_from_state

    net = torch_ref.random_model(arch, 64, 9, num_out, seed=13)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    cc = synth.CONFIGS[cfg][0]
    model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=(4, 4)), device=0, dtype="fp32")
    eng = get_engine(0)
    for n in (1, 5, 23, 1000, 4101):
        d = synth.synth_chunks_config(cfg, n, shard=40 + n)
        args = (d["signal"], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"], (4, 4))
        eng.profile_reset()
        eng.profile_enable(True)
        out = model.infer_chunks(*args)  # the shipped path: sig_conv3 (and the 5-tap layers) in Winograd form since round 6
        eng.profile_enable(False)
        prof = eng.profile()
        assert "sig3_front" in prof and "front_sig" not in prof, sorted(prof)  # the folded kernel is what ran
        # bit identity is a property of the DIRECT forms (RMR_WINOGRAD=0 on both sides): matrix-core producer against VALU producers
        direct = _env("RMR_WINOGRAD", "0", lambda: model.infer_chunks(*args))
        plain = _env("RMR_WINOGRAD", "0", lambda: _env("RMR_SIG3_MFMA", "0", lambda: model.infer_chunks(*args)))
        assert np.array_equal(direct, plain), (arch, cfg, n)
        assert np.abs(out - direct).max() <= 2e-5, (arch, cfg, n, float(np.abs(out - direct).max()))
        if n >= 1000:
            enc = O.compute_encoded_kmer_batch(4, 4, d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])
            with torch.no_grad():
                ref = net(torch.from_numpy(d["signal"]), torch.from_numpy(enc)).numpy()
            assert np.abs(out - ref).max() <= 1e-4, (arch, cfg, n)


def test_conv_w_ref_full_size_properties():
    """BASELINE configs[1] size (1 M C100 chunks, Conv_w_ref fp32) through the round-3 kernels (tap-by-tap seq_conv1, folded
    signal branch, occupancy-sized conv blocks): deterministic, independent of batch position, exact label tally, and a
    20 k-chunk sample against the CPU restatement of the reference network."""
    import torch

    from oracle import oracle as O
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    n = 1_000_000
    net = torch_ref.random_model("conv_only", 64, 9, 2, seed=0)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    model = model_from_state(state, dict(chunk_context=(50, 50), kmer_context_bases=(4, 4)), device=0, dtype="fp32")
    d = synth.synth_chunks_config("C100", n)
    dev = [torch.from_numpy(d[k]).cuda() for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
    counts = torch.zeros(2, dtype=torch.int64, device="cuda")
    out = model.infer_chunks(*dev, (4, 4), label_counts=counts)
    out2 = model.infer_chunks(*dev, (4, 4))
    assert torch.equal(out, out2) and bool(torch.isfinite(out).all())
    assert torch.equal(counts, torch.bincount(out.argmax(dim=1), minlength=2)) and int(counts.sum()) == n
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    outp = model.infer_chunks(*[t[perm].contiguous() for t in dev], (4, 4))
    assert torch.equal(outp, out[perm]), "result of a chunk depends on its batch position"
    idx = np.sort(np.random.default_rng(1).choice(n, 20000, replace=False))
    enc = O.compute_encoded_kmer_batch(4, 4, d["sequence"][idx], d["sequence_to_signal_mapping"][idx], d["sequence_lengths"][idx])
    with torch.no_grad():
        ref = net(torch.from_numpy(d["signal"][idx]), torch.from_numpy(enc)).numpy()
    assert np.abs(out[torch.from_numpy(idx).cuda()].cpu().numpy() - ref).max() <= 1e-4
