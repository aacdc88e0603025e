"""GPU parity of the Winograd forms of k_wino.hip - F(4, 5) for the 5-tap stride-1 convolutions (merge_conv1 of
models/ConvLSTM_w_ref.py:36-37,50 and merge_conv1 / merge_conv2 of models/Conv_w_ref.py:35-38,54-55 at size 64), polyphase F(4, 3)
for Conv_w_ref's stride-3 seq_conv3 (models/Conv_w_ref.py:31-32,51): within the fp32
tolerance (1e-4 on logits) of the reference-generated golden models and of the CPU restatement of the reference network, no
further from float64 than the direct form (RMR_WINOGRAD=0, k_conv.hip) by more than a rounding-level margin, the same bits
whatever the batch a chunk arrives in."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _direct(fn):
    os.environ["RMR_WINOGRAD"] = "0"
    try:
        return fn()
    finally:
        del os.environ["RMR_WINOGRAD"]


def _profile_of(eng, fn):
    eng.profile_reset()
    eng.profile_enable(True)
    out = fn()
    eng.profile_enable(False)
    return out, eng.profile()


@pytest.mark.parametrize("name", ["convlstm_s64_l100_o2", "convlstm_s64_l200_o3", "convlstm_s64_l100_k23", "conv_s64_l100_o2", "conv_s64_l100_o3"])
def test_golden_models_through_the_winograd_kernel(name):
    """Reference-generated logits (tools/gen_golden.py): <= 1e-4 through the Winograd kernel, and the direct form agrees with
    it to rounding."""
    from oracle import oracle as O
    from remora_amd.model_util import model_from_state

    g = np.load(os.path.join(GOLD, f"model_{name}.npz"))
    state = O.state_from_npz(g)
    size, kb, ka, L, num_out = (int(x) for x in g["params"])
    model = model_from_state(state, dict(chunk_context=(L // 2, L - L // 2), kmer_context_bases=(kb, ka)), device=0, dtype="fp32")
    args = (g["sigs"], g["seqs"], g["maps"], g["lens"], (kb, ka))
    out = model.infer_chunks(*args)
    direct = _direct(lambda: model.infer_chunks(*args))
    assert np.abs(out - g["logits"]).max() <= 1e-4, (name, float(np.abs(out - g["logits"]).max()))
    assert np.abs(direct - g["logits"]).max() <= 1e-4
    assert not np.array_equal(out, direct), "RMR_WINOGRAD=0 did not select another kernel"
    assert np.abs(out - direct).max() <= 2e-5, (name, float(np.abs(out - direct).max()))


@pytest.mark.parametrize("arch,cfg,num_out", [("conv_lstm", "C100", 2), ("conv_lstm", "C200", 3), ("conv_only", "C100", 2), ("conv_only", "C100", 3)])
def test_winograd_against_float64_and_the_direct_form(arch, cfg, num_out):
    """Random networks at torch's initialisation scale and the amplified synthetic one: distance to the float64 network of the
    Winograd kernel and of the direct form; ragged batch sizes (columns = groups of four positions flattened over the chunks, 16 per
    iteration: 1, 3, 17 ... chunks end inside a tile, C200's 57 positions end one position into a group)."""
    import torch

    from oracle import oracle as O
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.engine import get_engine
    from remora_amd.model_util import model_from_state

    cc, kcb = synth.CONFIGS[cfg][0], (4, 4)
    eng = get_engine(0)
    nets = [torch_ref.random_model(arch, 64, 9, num_out, seed=21)]
    if arch == "conv_lstm":
        nets.append(torch_ref.from_state(synth.synth_state("conv_lstm", 64, 9, num_out, seed=5)))
    for k, net in enumerate(nets):
        state = {kk: v.numpy() for kk, v in net.state_dict().items()}
        model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=kcb), device=0, dtype="fp32")
        net64 = torch_ref.build(arch, 64, 9, num_out).double()
        net64.load_state_dict({kk: v.double() if v.is_floating_point() else v for kk, v in net.state_dict().items()})
        for n in (1, 3, 17, 1000, 4099):
            d = synth.synth_chunks_config(cfg, n, shard=300 + n)
            args = (d["signal"], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"], kcb)
            out, prof = _profile_of(eng, lambda: model.infer_chunks(*args))
            assert "conv_merge1" in prof, sorted(prof)
            direct = _direct(lambda: model.infer_chunks(*args))
            enc = O.compute_encoded_kmer_batch(4, 4, d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])
            with torch.no_grad():
                exact = net64(torch.from_numpy(d["signal"]).double(), torch.from_numpy(enc).double()).numpy()
            ew, ed = float(np.abs(out - exact).max()), float(np.abs(direct - exact).max())
            assert ew <= 1e-4, (arch, cfg, k, n, ew, ed)
            assert ew <= 3.0 * ed + 2e-6, (arch, cfg, k, n, ew, ed)  # F(4, 5): about twice the direct form's rounding
            if n == 4099:  # a chunk's bits do not depend on its neighbours in the batch
                for start, m in ((0, 1), (5, 2), (100, 31), (1000, 1025)):
                    part = model.infer_chunks(*[a[start : start + m] for a in args[:4]], kcb)
                    assert np.array_equal(part, out[start : start + m]), (arch, cfg, k, start, m)


def test_winograd_full_size_properties():
    """BASELINE configs[2] size (1 M C100 chunks): deterministic, independent of batch position, exact label tally; the calls of
    the Winograd and the direct form agree on all but a rounding-level handful of chunks."""
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
    assert torch.equal(out, model.infer_chunks(*dev, (4, 4)))
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    outp = model.infer_chunks(*[t[perm].contiguous() for t in dev], (4, 4))
    assert torch.equal(outp, out[perm]), "result of a chunk depends on its batch position"
    direct = _direct(lambda: model.infer_chunks(*dev, (4, 4)))
    assert float((out - direct).abs().max()) <= 5e-5
    flips = int((out.argmax(dim=1) != direct.argmax(dim=1)).sum())
    assert flips <= n // 50_000, flips  # calls differ only where the two logits tie to rounding
