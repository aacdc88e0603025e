"""GPU parity tests of the fused bf16 front kernel (remora_amd/csrc/k_fused.hip): chunk arrays -> x bf16 in one
kernel, every intermediate in LDS.  Checked against (a) the CPU restatement of the reference network (oracle, fp32;
plain-bf16 tolerance of tests/test_gpu_parity.py: 3e-2 on logits + argmax agreement on clearly separated chunks),
(b) the unfused bf16 pipeline of the same library (RMR_FUSED=0), which rounds at the same places except for
sig_conv2 / seq_conv1 (fp32 VALU there, bf16 MFMA here)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BF16_TOL = 3e-2


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


def _unfused(fn):
    os.environ["RMR_FUSED"] = "0"
    try:
        return fn()
    finally:
        del os.environ["RMR_FUSED"]


def _awkward(d, rng):
    """Edge rows on top of the synthetic generator: missing (-1) bases, zero-dwell bases, the shortest sequences,
    garbage in the padding columns."""
    seqs, maps, lens = d["sequence"].copy(), d["sequence_to_signal_mapping"].copy(), d["sequence_lengths"].copy()
    n, L = lens.size, d["chunk_len"]
    for c in range(0, n, 5):  # N bases anywhere, incl. the context columns
        seqs[c, rng.integers(0, seqs.shape[1], 3)] = -1
    for c in range(1, n, 7):  # zero-dwell: repeat a cut
        sl = int(lens[c])
        if sl >= 3:
            maps[c, 2] = maps[c, 1]
    for c in range(2, n, 11):  # one or two bases cover the whole chunk
        sl = 1 + (c % 2)
        maps[c, : sl + 1] = [0, L] if sl == 1 else [0, L // 3, L]
        lens[c] = sl
    for c in range(n):  # padding columns are uninitialised in the reference's datasets (data_chunks.py:1379-1388)
        sl = int(lens[c])
        maps[c, sl + 1 :] = rng.integers(-300, 300, maps.shape[1] - sl - 1)
        seqs[c, sl + 8 :] = rng.integers(-1, 4, seqs.shape[1] - sl - 8)
    return seqs, maps, lens


@pytest.mark.parametrize("cfg,num_out", [("C100", 2), ("C200", 3)])
def test_fused_front_vs_oracle_and_unfused(torch_cuda, O, cfg, num_out):
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.engine import get_engine
    from remora_amd.model_util import model_from_state

    torch = torch_cuda
    net = torch_ref.random_model("conv_lstm", 64, 9, num_out, seed=5)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    cc = synth.CONFIGS[cfg][0]
    model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=(4, 4)), device=0, dtype="bf16")
    eng = get_engine(0)
    rng = np.random.default_rng(9)
    for n in (1, 2, 3, 5, 37, 1000, 4099):  # not multiples of the chunks-per-iteration, of 16, of anything
        d = synth.synth_chunks_config(cfg, n, shard=100 + n)
        seqs, maps, lens = _awkward(d, rng) if n >= 37 else (d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])
        eng.profile_reset()
        eng.profile_enable(True)
        out = model.infer_chunks(d["signal"], seqs, maps, lens, (4, 4))
        eng.profile_enable(False)
        prof = eng.profile()
        assert "fused_front" in prof and "front_seq" not in prof, prof.keys()  # the fused kernel is what ran
        out_u = _unfused(lambda: model.infer_chunks(d["signal"], seqs, maps, lens, (4, 4)))
        enc = O.compute_encoded_kmer_batch(4, 4, seqs, maps, lens)
        with torch.no_grad():
            ref = net(torch.from_numpy(d["signal"]), torch.from_numpy(enc)).numpy()
        assert np.isfinite(out).all()
        err, err_u = np.abs(out - ref).max(), np.abs(out_u - ref).max()
        assert err <= BF16_TOL, (cfg, n, err, err_u)
        assert np.abs(out - out_u).max() <= BF16_TOL, (cfg, n)
        srt = np.sort(ref, axis=1)
        clear = (srt[:, -1] - srt[:, -2]) > 4 * BF16_TOL
        assert np.array_equal(out.argmax(1)[clear], ref.argmax(1)[clear])


# 16-bit error on networks with the REFERENCE's default initialisation (the golden models): measured 5.5e-4 .. 1.5e-3 in f16
# and 5.0e-3 .. 5.3e-3 in bf16 on logits of magnitude 0.9 .. 1.6 (tools/measure_16bit_parity.py, round 4); the gates sit 25-30 %
# above the worst model of each type.  (The synthetic networks of bench.py are deliberately 5-10x more sensitive: below.)
GOLDEN_TOL = {"f16": 2e-3, "bf16": 6.5e-3}


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_fused_front_golden_models(torch_cuda, O, dtype):
    """The reference-generated logits (tests/golden/model_convlstm_*.npz; weights with non-trivial BatchNorm stats)."""
    from conftest import golden
    from remora_amd.engine import get_engine
    from remora_amd.model_util import model_from_state

    for name in ("convlstm_s64_l100_o2", "convlstm_s64_l200_o3", "convlstm_s64_l100_k23"):
        g = golden(f"model_{name}.npz")
        state = O.state_from_npz(g)
        size, kb, ka, L, num_out = (int(x) for x in g["params"])
        model = model_from_state(state, dict(chunk_context=(L // 2, L - L // 2), kmer_context_bases=(kb, ka)), device=0,
                                 dtype=dtype)
        eng = get_engine(0)
        eng.profile_reset()
        eng.profile_enable(True)
        out = model.infer_chunks(g["sigs"], g["seqs"], g["maps"], g["lens"], (kb, ka))
        eng.profile_enable(False)
        assert "fused_front" in eng.profile()  # k-mer lengths 9 (4,4) and 6 (2,3) are instantiated
        assert np.abs(out - g["logits"]).max() <= GOLDEN_TOL[dtype], (name, dtype, float(np.abs(out - g["logits"]).max()))


def test_fused_front_full_size_properties(torch_cuda):
    """1M chunks (BASELINE configs[3] shape per GPU): deterministic, independent of batch position (sub-batch /
    block-iteration / tile boundaries), exact label tally, device and host-buffer entry agree."""
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    torch = torch_cuda
    n = 1_000_000
    net = torch_ref.random_model("conv_lstm", 64, 9, 2, seed=0)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    model = model_from_state(state, dict(chunk_context=(50, 50), kmer_context_bases=(4, 4)), device=0, dtype="bf16")
    d = synth.synth_chunks_config("C100", n)
    keys = ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")
    dev = [torch.from_numpy(d[k]).cuda() for k in keys]
    counts = torch.zeros(2, dtype=torch.int64, device="cuda")
    out = model.infer_chunks(*dev, (4, 4), label_counts=counts)
    out2 = model.infer_chunks(*dev, (4, 4))
    assert torch.equal(out, out2) and bool(torch.isfinite(out).all())
    assert torch.equal(counts, torch.bincount(out.argmax(dim=1), minlength=2)) and int(counts.sum()) == n
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    outp = model.infer_chunks(*[t[perm].contiguous() for t in dev], (4, 4))
    assert torch.equal(outp, out[perm]), "result of a chunk depends on its batch position"
    k = 300_001
    out_h = model.infer_chunks(*[d[key][:k] for key in keys], (4, 4))
    assert np.array_equal(out_h, out[:k].cpu().numpy())


def test_call_reads_mods_subbatch_pipeline_equals_one_batch(torch_cuda):
    """A large call_reads_mods batch is walked in sub-batches with the staging of the next one under the kernels of the
    current one (RMR_READS_SUBBATCH); the per-read results must be those of the one-batch path, bit for bit."""
    from remora_amd import synth
    from remora_amd.data_chunks import RemoraRead
    from remora_amd.inference import call_reads_mods
    from remora_amd.model_util import model_from_state

    md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"],
              can_base="C", base_start_justify=False, offset=0, sig_map_refiner=None)
    model = model_from_state(synth.synth_state("conv_lstm", 64, 9, 2, seed=2), md, device=0)
    reads = []
    for i in range(700):
        r = synth.synth_read(300 + (i % 7) * 50, idx=i)
        reads.append(RemoraRead(dacs=r["dacs"], shift=r["shift"], scale=r["scale"], seq_to_sig_map=r["seq_to_sig_map"],
                                int_seq=r["int_seq"], read_id=f"r{i}"))
    os.environ["RMR_READS_SUBBATCH"] = "0"
    try:
        whole = call_reads_mods(reads, model, md)
        os.environ["RMR_READS_SUBBATCH"] = "128"
        piped = call_reads_mods(reads, model, md)
    finally:
        del os.environ["RMR_READS_SUBBATCH"]
    assert len(whole) == len(piped) == 700
    for a, b in zip(whole, piped):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert sum(r[2].size for r in whole) > 5000


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "f16"])
@pytest.mark.parametrize("cc,msl", [((200, 200), 80), ((300, 300), 60), ((500, 500), 60), ((500, 500), 150), ((150, 150), 62)])
def test_long_chunk_contexts_run_and_match(torch_cuda, O, dtype, cc, msl):
    """Chunk contexts far beyond the benchmark shapes, starting with the reference's default chunk_context (200, 200) and its
    max_seq_len 400 // 5 = 80 (src/remora/constants.py:7-8, prepare_train_data.py:165): the reference network is
    length-agnostic (models/ConvLSTM_w_ref.py:39-58), so every shape must run.  The 16-bit pipelines stay on the fused kernel
    (a chunk that does not fit a CU's LDS goes through it in position windows of 96 outputs), the fp32 merge_conv1 stages a
    long chunk window by window.  16-bit: mean error at the format's level; the maximum grows with the number of LSTM steps
    (T = 124 / 191 / 324 here) - see test_16bit_error_is_the_arithmetic_not_the_kernel."""
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.engine import get_engine
    from remora_amd.model_util import model_from_state

    torch = torch_cuda
    L = sum(cc)
    net = torch_ref.random_model("conv_lstm", 64, 9, 2, seed=11)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=(4, 4)), device=0, dtype=dtype)
    d = synth.synth_chunks(203, L, msl, (4, 4), seed=31)
    args = (d["signal"], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"], (4, 4))
    eng = get_engine(0)
    eng.profile_reset()
    eng.profile_enable(True)
    out = model.infer_chunks(*args)
    eng.profile_enable(False)
    if dtype != "fp32":
        assert "fused_front" in eng.profile(), (cc, msl, list(eng.profile()))
    enc = O.compute_encoded_kmer_batch(4, 4, d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])
    with torch.no_grad():
        ref = net(torch.from_numpy(d["signal"]), torch.from_numpy(enc)).numpy()
    assert np.isfinite(out).all()
    err = np.abs(out - ref)
    if dtype == "fp32":
        assert err.max() <= 1e-4, (cc, err.max())
    elif dtype == "f16":
        assert err.mean() <= 5e-4 and err.max() <= 0.02, (cc, err.mean(), err.max())
    else:
        assert err.mean() <= 4e-3 and err.max() <= 0.15, (cc, err.mean(), err.max())


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("cfg", ["C100", "C200"])
def test_position_windows_are_bit_identical_to_whole_chunks(torch_cuda, cfg, dtype):
    """RMR_FUSED_WINDOWS=1 sends shapes that fit through the window kernels (C100: one short window per chunk, T 24 < 96; C200:
    T 58): every output is the same accumulation in the same order, so the logits must not move by a bit - including ragged
    batches whose last iteration holds fewer virtual chunks than the block takes."""
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    cc, kcb, _, num_out, _ = synth.CONFIGS[cfg]
    state = synth.synth_state("conv_lstm", 64, 9, num_out, seed=4)
    model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=kcb), device=0, dtype=dtype)
    for n in (1, 7, 1001):
        d = synth.synth_chunks_config(cfg, n, shard=40 + n)
        args = (d["signal"], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"], kcb)
        whole = model.infer_chunks(*args)
        os.environ["RMR_FUSED_WINDOWS"] = "1"
        try:
            windows = model.infer_chunks(*args)
        finally:
            del os.environ["RMR_FUSED_WINDOWS"]
        assert np.array_equal(whole, windows), (cfg, dtype, n, float(np.abs(whole - windows).max()))


# ---- the 16-bit pipelines (bf16: BASELINE configs[3]/[4]; f16: the same kernels on IEEE half) --------------------------
def _centred_pair(torch, cfg, n, dtype, seed=0):
    """(fp32-path logits, `dtype`-path logits) of n synthetic chunks, both minus the per-class median of the fp32 logits
    (what bench.py folds into fc.bias: random weights otherwise call one class for every chunk), as device tensors."""
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    cc, kcb, _, num_out, _ = synth.CONFIGS[cfg]
    state = synth.synth_state("conv_lstm", 64, 9, num_out, seed=seed)
    md = dict(chunk_context=cc, kmer_context_bases=kcb)
    d = synth.synth_chunks_config(cfg, n, shard=3)
    dev = [torch.from_numpy(d[k]).cuda() for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
    ref = model_from_state(state, md, device=0, dtype="fp32").infer_chunks(*dev, kcb)
    out = model_from_state(state, md, device=0, dtype=dtype).infer_chunks(*dev, kcb)
    med = ref.median(dim=0).values
    return ref - med, out - med, med


# Gates of the synthetic (deliberately amplified: synth.py scales conv x2.45, LSTM x2.5, fc x12) networks = the measured level
# x 1.25-1.3 (tools/measure_16bit_parity.py, round 4; the kernels are deterministic, the margin is for other shapes of the
# same arithmetic only).  Measured max / mean / 99.9th percentile over 100 k chunks against the fp32 path:
#   C100 f16 5.9e-3 / 1.4e-4 / 2.6e-3     C100 bf16 4.6e-2 / 1.3e-3 / 2.7e-2
#   C200 f16 8.4e-2 / 2.4e-4 / 1.4e-2     C200 bf16 0.244  / 2.1e-3 / 0.131
# and over the first 8192 chunks against the ORACLE's fp32 forward (the comparand that is not this library):
#   C100 f16 5.9e-3, bf16 3.9e-2;  C200 f16 3.8e-2, bf16 0.231
@pytest.mark.parametrize("cfg,dtype,max_abs,mean_abs,agree,p999_max,oracle_max", [
    ("C100", "f16", 8e-3, 2e-4, 0.999, 3.5e-3, 8e-3), ("C200", "f16", 0.11, 3.2e-4, 0.999, 1.9e-2, 5e-2),
    ("C100", "bf16", 6e-2, 1.75e-3, 0.999, 3.6e-2, 5.2e-2), ("C200", "bf16", 0.32, 2.8e-3, 0.995, 0.17, 0.30)])
def test_16bit_pipelines_against_the_fp32_path_100k(torch_cuda, O, cfg, dtype, max_abs, mean_abs, agree, p999_max, oracle_max):
    """100 k chunks of the configs[3] / configs[4] shapes, 16-bit pipeline against the fp32 pipeline (itself within 5e-6 of a
    float64 evaluation, bench.py `precision`) AND, on the first 8192 chunks, against the oracle's fp32 forward: maximum, mean,
    99.9th percentile of the per-chunk maximum, argmax agreement over ALL chunks whose fp32 margin exceeds 2e-2.  At C200
    (58 LSTM steps of a deliberately sensitive random network) single chunks amplify ANY 16-bit rounding (the float64
    emulation of half arithmetic shows 0.11 on 30 k chunks, of bf16 0.26; oracle/lowp_emulation.py); on networks with the
    reference's initialisation the same kernels stay below 2e-3 / 6.5e-3 (test_fused_front_golden_models).  bf16's numbers
    are those of its 8-bit mantissa, not of the kernels: test_16bit_error_is_the_arithmetic_not_the_kernel."""
    from oracle import torch_ref
    from remora_amd import synth

    torch = torch_cuda
    ref, out, med = _centred_pair(torch, cfg, 100_000, dtype)
    # the oracle as comparand: same chunks (shard 3), same state (seed 0), first 8192
    cc, kcb, _, num_out, _ = synth.CONFIGS[cfg]
    d = synth.synth_chunks_config(cfg, 100_000, shard=3)
    net = torch_ref.from_state(synth.synth_state("conv_lstm", 64, 9, num_out, seed=0))
    enc = torch.from_numpy(O.compute_encoded_kmer_batch(kcb[0], kcb[1], d["sequence"][:8192], d["sequence_to_signal_mapping"][:8192],
                                                        d["sequence_lengths"][:8192]))
    with torch.no_grad():
        ref_cpu = net(torch.from_numpy(d["signal"][:8192]), enc)
    err_oracle = float((out[:8192].cpu() - (ref_cpu - med.cpu())).abs().max())
    assert err_oracle <= oracle_max, (cfg, dtype, err_oracle)
    err = (out - ref).abs()
    top = ref.topk(2, dim=1).values
    clear = (top[:, 0] - top[:, 1]) > 2e-2
    assert int(clear.sum()) > 50_000
    agreement = float((out.argmax(1) == ref.argmax(1))[clear].float().mean())
    stats = (cfg, dtype, float(err.max()), float(err.mean()), agreement)
    assert float(err.mean()) <= mean_abs, stats
    assert agreement >= agree, stats
    assert float(err.max()) <= max_abs, stats
    p999 = float(torch.quantile(err.max(dim=1).values.float(), 0.999))
    assert p999 <= p999_max, stats + (p999,)


# ---- SPECIFIED gates of the 16-bit configurations (round-4 review item 7; SURVEY §7 "hard parts": the 16-bit configs need a
# tolerance of their own): fixed numbers, not measured levels with a margin.  Over 1 M chunks of the BASELINE configs[3] /
# configs[4] shapes a 16-bit pipeline must reproduce the fp32 path's call on at least this share of the CONFIDENT chunks
# (fp32 margin between the two largest logits > 2e-2), and on the first 20 000 chunks the same against the ORACLE's fp32
# forward.  Round 4 measured 1.0 / 0.999995 (f16, C100 / C200) and 0.99993 / 0.99813 (bf16): bf16 at configs[4] flips 0.19 %
# of the confident calls of this deliberately amplified network - the README says so and recommends f16.
SPECIFIED_AGREEMENT = {"f16": 0.9999, "bf16": 0.998}


@pytest.mark.parametrize("cfg", ["C100", "C200"])
def test_16bit_configs_meet_their_specified_argmax_gates_over_1m_chunks(torch_cuda, O, cfg):
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    torch = torch_cuda
    n, k = 1_000_000, 20_000
    cc, kcb, _, num_out, _ = synth.CONFIGS[cfg]
    state = synth.synth_state("conv_lstm", 64, 9, num_out, seed=0)
    md = dict(chunk_context=cc, kmer_context_bases=kcb)
    d = synth.synth_chunks_config(cfg, n, shard=3)
    dev = [torch.from_numpy(d[key]).cuda() for key in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
    ref = model_from_state(state, md, device=0, dtype="fp32").infer_chunks(*dev, kcb)
    med = ref.median(dim=0).values  # centred as bench.py centres them: random weights otherwise call one class throughout
    ref = ref - med
    top = ref.topk(2, dim=1).values
    clear = (top[:, 0] - top[:, 1]) > 2e-2
    assert int(clear.sum()) > 500_000
    enc = torch.from_numpy(O.compute_encoded_kmer_batch(kcb[0], kcb[1], d["sequence"][:k], d["sequence_to_signal_mapping"][:k], d["sequence_lengths"][:k]))
    with torch.no_grad():
        oracle = torch_ref.from_state(state)(torch.from_numpy(d["signal"][:k]), enc) - med.cpu()
    otop = oracle.topk(2, dim=1).values
    oclear = (otop[:, 0] - otop[:, 1]) > 2e-2
    # (sanity only: the fp32 path and the oracle are the same function - the two fp32 evaluations of this deliberately amplified
    #  network sit 1e-5 apart at C100 and 1.5e-4 at C200, 58 LSTM steps; the fp32 gate proper is on the golden models)
    assert float((ref[:k].cpu() - oracle).abs().max()) <= 5e-4
    for dtype, gate in SPECIFIED_AGREEMENT.items():
        out = model_from_state(state, md, device=0, dtype=dtype).infer_chunks(*dev, kcb) - med
        agreement = float((out.argmax(1) == ref.argmax(1))[clear].float().mean())
        sample = float((out[:k].cpu().argmax(1) == oracle.argmax(1))[oclear].float().mean())
        assert agreement >= gate, (cfg, dtype, agreement, int(clear.sum()))
        # the 20 k sample against the oracle: the same gate less two standard errors of a sample of its size (bf16 at C200:
        # 0.9978 of 15 750 confident chunks here against 0.9981 of 790 596 above)
        slack = 2.0 * (gate * (1.0 - gate) / max(int(oclear.sum()), 1)) ** 0.5
        assert sample >= gate - slack, (cfg, dtype, sample, int(oclear.sum()))



@pytest.mark.parametrize("cfg,n", [("C100", 4096), ("C200", 2048)])
def test_16bit_error_is_the_arithmetic_not_the_kernel(torch_cuda, O, cfg, n):
    """The kernels against (a) the float64 network and (b) the float64 network with round-to-nearest-even to bf16 / half at
    the kernels' rounding sites (oracle/lowp_emulation.py): the GPU's error statistics sit at the emulation's level - the
    kernels add nothing on top of what 16-bit operands cost (accumulation order and the 2-ulp exp/rcp are invisible at
    this scale).  This is what bounds the plain-bf16 parity: its error is the format's."""
    from oracle import lowp_emulation, torch_ref
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    torch = torch_cuda
    cc, kcb, _, num_out, _ = synth.CONFIGS[cfg]
    state = synth.synth_state("conv_lstm", 64, 9, num_out, seed=0)
    net = torch_ref.from_state(state)
    d = synth.synth_chunks_config(cfg, n, shard=5)
    enc = torch.from_numpy(O.compute_encoded_kmer_batch(kcb[0], kcb[1], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"]))
    sig = torch.from_numpy(d["signal"])
    with torch.no_grad():
        exact = lowp_emulation.forward(net, sig, enc, sites=()).numpy()
        for dtype in ("bf16", "f16"):
            emu = np.abs(lowp_emulation.forward(net, sig, enc, fmt=dtype).numpy() - exact)
            model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=kcb), device=0, dtype=dtype)
            gpu = np.abs(model.infer_chunks(d["signal"], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"], kcb) - exact)
            stats = (cfg, dtype, gpu.mean(), emu.mean(), np.quantile(gpu, 0.99), np.quantile(emu, 0.99), gpu.max(), emu.max())
            assert gpu.mean() <= 1.3 * emu.mean() + 2e-5, stats
            assert np.quantile(gpu, 0.99) <= 1.4 * np.quantile(emu, 0.99) + 1e-4, stats
            assert emu.mean() <= 2.5 * gpu.mean() + 2e-5, stats  # ... and the emulation is the right order of magnitude (the
            # kernels round a little less often than it assumes: measured 0.5-0.6 of its mean in f16, 0.9-1.0 in bf16)


STREAM16_SITES = ("wconv.sig3", "wconv.seq2", "wconv.merge1", "aconv.sig2", "aconv.seq1", "aconv.cat", "x", "wlstm", "h")


@pytest.mark.parametrize("cfg,size", [("C100", 128), ("C100", 96), ("C200", 160), ("C100", 256), ("C100", 100)])
def test_16bit_networks_of_more_than_64_channels(torch_cuda, O, cfg, size):
    """bf16 / f16 above 64 channels (k_stream16.hip: fp32 front kernels, then sig_conv3 / seq_conv2 / merge_conv1 and the LSTM on the
    16-bit matrix cores with streamed weights, 16-bit cat and x in HBM; a size that is no multiple of 32 runs with zero-weight
    channels).  Against the float64 network and against the float64 network rounded at THIS pipeline's 16-bit sites: the error is
    the arithmetic's, as for the fused kernels; ragged batches; chunk position does not matter; the dense-tensor entry works."""
    from oracle import lowp_emulation, torch_ref
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    torch = torch_cuda
    cc, kcb, _, num_out, _ = synth.CONFIGS[cfg]
    state = synth.synth_state("conv_lstm", size, 9, num_out, seed=3)
    net = torch_ref.from_state(state)
    n = 1500
    d = synth.synth_chunks_config(cfg, n, shard=9)
    enc_np = O.compute_encoded_kmer_batch(kcb[0], kcb[1], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])
    enc, sig = torch.from_numpy(enc_np), torch.from_numpy(d["signal"])
    keys = ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")
    with torch.no_grad():
        exact = lowp_emulation.forward(net, sig, enc, sites=()).numpy()
        for dtype in ("bf16", "f16"):
            # the emulation's error is one DRAW of the rounding errors; on these amplified networks its mean spreads 2-3 x between
            # draws (lowp_emulation.forward `prescale`; profiles/r06_stream16_parity.log), so the kernel is held to the envelope of four
            emus = [np.abs(lowp_emulation.forward(net, sig, enc, sites=STREAM16_SITES, fmt=dtype, prescale=ps).numpy() - exact)
                    for ps in (1.0, 1.19, 1.4426950408889634, 1.7)]
            emu_mean, emu_q99 = [e.mean() for e in emus], [np.quantile(e, 0.99) for e in emus]
            model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=kcb), device=0, dtype=dtype)
            assert model.kernel_size == (size + 31) // 32 * 32
            out = model.infer_chunks(*[d[k] for k in keys], kcb)
            gpu = np.abs(out - exact)
            stats = (cfg, size, dtype, gpu.mean(), emu_mean, np.quantile(gpu, 0.99), emu_q99, gpu.max())
            assert gpu.mean() <= 1.3 * max(emu_mean) + 2e-5, stats
            assert np.quantile(gpu, 0.99) <= 2.0 * max(emu_q99) + 1e-4, stats  # (the top 30 of 3000 values: a wider spread than the mean's)
            assert min(emu_mean) <= 3.0 * gpu.mean() + 2e-5, stats
            for start, m in ((0, 1), (7, 63), (100, 65), (300, 257)):  # ragged batches return the same bits
                part = model.infer_chunks(*[d[k][start : start + m] for k in keys], kcb)
                assert np.array_equal(part, out[start : start + m]), (cfg, size, dtype, start, m)
            # the dense-tensor entry sums seq_conv1 over the one-hot's 36 channels where the chunk-array entry sums table rows per
            # k-mer slot: fp32 seq1 differs in the last bits, which now and then moves one 16-bit rounding downstream
            dense = model(sig[:64].cuda(), enc[:64].cuda()).cpu().numpy()
            assert np.abs(dense - out[:64]).max() <= (2e-2 if dtype == "bf16" else 3e-3), (cfg, size, dtype, float(np.abs(dense - out[:64]).max()))


def test_f16_dtype_refuses_what_it_cannot_run(torch_cuda, O):
    """f16 exists on the fused kernels only: Conv_w_ref, size 16 and a dense one-hot input are refused with messages."""
    from conftest import golden
    from remora_amd import RemoraError
    from remora_amd.model_util import model_from_state

    with pytest.raises(RemoraError):
        model_from_state(O.state_from_npz(golden("model_conv_s64_l100_o2.npz")), dict(chunk_context=(50, 50)), device=0, dtype="f16")
    with pytest.raises(RemoraError):
        model_from_state(O.state_from_npz(golden("model_convlstm_s16_l100_o2.npz")), dict(chunk_context=(50, 50)), device=0, dtype="f16")
    g = golden("model_convlstm_s64_l100_o2.npz")
    model = model_from_state(O.state_from_npz(g), dict(chunk_context=(50, 50), kmer_context_bases=(4, 4)), device=0, dtype="f16")
    out = model.infer_chunks(g["sigs"], g["seqs"], g["maps"], g["lens"], (4, 4))
    assert np.abs(out - g["logits"]).max() <= 4e-3
    with pytest.raises(RemoraError, match="fused kernels only"):
        model(torch_cuda.from_numpy(g["sigs"][:8]).cuda(), torch_cuda.from_numpy(g["dense_seqs"]).cuda())
