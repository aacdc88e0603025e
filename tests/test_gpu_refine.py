"""N2 parity: signal-mapping refinement on the GPU (rmr_refine_signal_maps through
remora_amd.refine_signal_map.SigMapRefiner) against vectors generated from the reference
(tests/golden/refine_signal_map.npz, tools/gen_golden.py gen_refine) and against the CPU oracle
on seeded random reads.  Integer paths: bit-exact."""
import json
import os

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu

READS = ["a", "b", "c", "d"]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


@pytest.fixture(scope="module")
def G():
    return golden("refine_signal_map.npz")


def _refiner(G, **kw):
    from remora_amd.refine_signal_map import SigMapRefiner

    return SigMapRefiner(_levels_array=G["kmer_levels"], center_idx=int(G["center_idx"]), **kw)


@pytest.mark.parametrize("algo", ["Viterbi", "dwell_penalty"])
@pytest.mark.parametrize("hbw", [5, 2])
def test_refine_dp_golden_paths(torch_cuda, G, algo, hbw):
    """seq_banded_dp paths of the reference for four reads, both algorithms, two band widths,
    all four reads in one call."""
    ref = _refiner(G, scale_iters=0, algo=algo, half_bandwidth=hbw)
    outs = ref.refine_maps([G[f"{n}_dacs"] for n in READS], [505.0] * 4, [83.0] * 4, [G[f"{n}_map"] for n in READS],
                           [G[f"{n}_int_seq"] for n in READS])
    for n, o in zip(READS, outs):
        np.testing.assert_array_equal(o, G[f"{n}_hbw{hbw}_{algo}_path"], err_msg=f"read {n}")


def test_refine_rowwise_kernel_golden(torch_cuda, G, monkeypatch):
    """The row-by-row kernel (fallback of the column kernel) reproduces the same paths."""
    monkeypatch.setenv("RMR_REFINE_ROWWISE", "1")
    for algo in ("Viterbi", "dwell_penalty"):
        ref = _refiner(G, scale_iters=0, algo=algo, half_bandwidth=5)
        outs = ref.refine_maps([G[f"{n}_dacs"] for n in READS], [505.0] * 4, [83.0] * 4,
                               [G[f"{n}_map"] for n in READS], [G[f"{n}_int_seq"] for n in READS])
        for n, o in zip(READS, outs):
            np.testing.assert_array_equal(o, G[f"{n}_hbw5_{algo}_path"], err_msg=f"read {n} {algo}")


def test_refine_read_flow_golden(torch_cuda, G):
    """RemoraRead.refine_signal_mapping with four refiner settings (rough re-scale on/off and
    method, scale_iters -1/0/2, both algorithms) = the reference's maps, shift and scale."""
    from remora_amd.data_chunks import RemoraRead

    for si, st in enumerate(json.loads(str(G["settings_json"]))):
        ref = _refiner(G, **st)
        for n in READS:
            np.random.seed(1000 + si)
            read = RemoraRead(dacs=G[f"{n}_dacs"], shift=505.0, scale=83.0, seq_to_sig_map=G[f"{n}_map"].copy(),
                              int_seq=G[f"{n}_int_seq"], read_id=n)
            read.refine_signal_mapping(ref)
            np.testing.assert_array_equal(read.seq_to_sig_map, G[f"s{si}_{n}_map"], err_msg=f"setting {si} read {n}")
            np.testing.assert_allclose([read.shift, read.scale], G[f"s{si}_{n}_shift_scale"], rtol=1e-12)


def test_refine_reads_batched_equals_per_read(torch_cuda, G):
    from remora_amd.data_chunks import RemoraRead

    st = json.loads(str(G["settings_json"]))[0]
    ref = _refiner(G, **st)
    reads = [RemoraRead(dacs=G[f"{n}_dacs"], shift=505.0, scale=83.0, seq_to_sig_map=G[f"{n}_map"].copy(),
                        int_seq=G[f"{n}_int_seq"], read_id=n) for n in READS]
    errs = ref.refine_reads(reads)
    assert errs == [None] * 4
    for n, r in zip(READS, reads):
        np.testing.assert_array_equal(r.seq_to_sig_map, G[f"s0_{n}_map"])
        np.testing.assert_allclose([r.shift, r.scale], G[f"s0_{n}_shift_scale"], rtol=1e-12)


def _random_read(rng, table, k, center, nbases, zero_frac=0.0, stall=False, trim=False, noise=0.3):
    int_seq = rng.integers(0, 4, nbases).astype(np.int8)
    dwell = rng.integers(1, 14, nbases)
    if zero_frac:
        dwell[rng.random(nbases) < zero_frac] = 0
        dwell[-1] = max(dwell[-1], 1)
    if stall:
        dwell[rng.integers(5, nbases - 5, 2)] = rng.integers(200, 700, 2)
    lead, tail = (int(rng.integers(0, 50)), int(rng.integers(0, 50))) if trim else (0, 0)
    smap = lead + np.concatenate([[0], np.cumsum(dwell)]).astype(np.int64)
    idx = np.zeros(nbases - k + 1, np.int64)
    for j in range(k):
        idx = idx * 4 + int_seq[j : nbases - k + 1 + j]
    lv = np.zeros(nbases, np.float32)
    lv[center : center + nbases - k + 1] = table[idx]
    norm = np.concatenate([np.zeros(lead), np.repeat(lv, dwell), np.zeros(tail)])
    norm = norm + noise * rng.standard_normal(norm.size)
    dacs = np.round(400 + 60 * norm).astype(np.int16)
    # perturb the mapping so that the DP has something to move
    jit = smap.copy()
    jit[1:-1] += rng.integers(-4, 5, nbases - 1)
    jit = np.maximum.accumulate(np.clip(jit, smap[0], smap[-1]))
    jit[0], jit[-1] = smap[0], smap[-1]
    return dacs, jit, int_seq


@pytest.mark.parametrize("algo,sd_len", [("Viterbi", 3), ("dwell_penalty", 1), ("dwell_penalty", 3),
                                          ("dwell_penalty", 6), ("dwell_penalty", 8)])
def test_refine_random_batches_vs_oracle(torch_cuda, O, algo, sd_len):
    """Seeded random reads (ragged lengths, zero-dwell bases, stalls, maps that neither start at 0
    nor end at the signal end) in one batch: paths and validate_band errors equal the oracle's.
    sd_len 8 exceeds the register path and exercises the row-wise kernel."""
    from remora_amd.refine_signal_map import SigMapRefiner

    rng = np.random.default_rng(100 + sd_len + (algo == "Viterbi"))
    k, center = 4, 1
    table = rng.normal(0, 1, 4**k).astype(np.float32)
    sd = (0.5 * np.square(np.arange(sd_len, dtype=np.float32) - (sd_len + 1))).astype(np.float32)
    for hbw in (5, 3, 1):
        ref = SigMapRefiner(_levels_array=table, center_idx=center, scale_iters=0, algo=algo, half_bandwidth=hbw,
                            sd_arr=sd)
        reads = []
        for i in range(24):
            nb = int(rng.integers(12, 700))
            reads.append(_random_read(rng, table, k, center, nb, zero_frac=0.15 if i % 3 == 0 else 0.0,
                                      stall=(i % 4 == 1 and nb > 20), trim=(i % 2 == 0)))
        shifts = rng.uniform(380, 420, len(reads))
        scales = rng.uniform(50, 70, len(reads))
        outs, status, dev = ref._refine_batch([r[0] for r in reads], shifts, scales, [r[1] for r in reads],
                                              [r[2] for r in reads])
        n_err = 0
        for i, (d, m, s) in enumerate(reads):
            want, err = O.refine_one(d, shifts[i], scales[i], m, s, table, center, hbw, algo, sd)
            if err is not None:
                n_err += 1
                assert status[i] != 0 and dev.status_message(status[i]) == err, (i, status[i], err)
            else:
                assert status[i] == 0, (i, dev.status_message(status[i]))
                np.testing.assert_array_equal(outs[i], want, err_msg=f"hbw {hbw} read {i}")
        assert n_err < len(reads)


def test_refine_large_score_replay_vs_oracle(torch_cuda, O):
    """With a level table whose spread makes single residuals exceed LARGE_SCORE (100), the
    'LARGE_SCORE + last score of the previous row' term of the dwell-penalty step wins in many
    places; the column kernel speculates it never does, must notice when the previous row
    completes, and replay from its checkpoints with the term known, without leaving the kernel."""
    from remora_amd.engine import get_engine
    from remora_amd import _lib as L
    from remora_amd.refine_signal_map import SigMapRefiner
    import ctypes

    rng = np.random.default_rng(7)
    k, center = 3, 1
    table = rng.normal(0, 25, 4**k).astype(np.float32)
    ref = SigMapRefiner(_levels_array=table, center_idx=center, scale_iters=0, half_bandwidth=4)
    reads = [_random_read(rng, table, k, center, int(rng.integers(30, 300)), noise=8.0) for _ in range(16)]
    eng = get_engine(0)
    lib = L.lib()
    L.check(lib.rmr_profile_enable(eng.handle, 1))
    L.check(lib.rmr_profile_reset(eng.handle))
    outs, status, dev = ref._refine_batch([r[0] for r in reads], [400.0] * 16, [60.0] * 16, [r[1] for r in reads],
                                          [r[2] for r in reads])
    names = [lib.rmr_profile_kernel_name(i).decode() for i in range(lib.rmr_profile_num_kernels())]
    ms, cnt = ctypes.c_double(), ctypes.c_int64()
    L.check(lib.rmr_profile_get(eng.handle, names.index("refine_dp_rowwise"), ctypes.byref(ms), ctypes.byref(cnt)))
    L.check(lib.rmr_profile_enable(eng.handle, 0))
    for i, (d, m, s) in enumerate(reads):
        want, err = O.refine_one(d, 400.0, 60.0, m, s, table, center, 4, "dwell_penalty", ref.sd_arr)
        assert err is None and status[i] == 0
        np.testing.assert_array_equal(outs[i], want, err_msg=f"read {i}")
    assert cnt.value == 0, "replay should have repaired every read inside the column kernel"


def test_refine_long_read_properties(torch_cuda, O):
    """A 60k-base read (beyond what the per-test oracle budget covers at every setting):
    refined map is monotone, keeps its end points, and equals the oracle for one setting."""
    from remora_amd.refine_signal_map import SigMapRefiner

    rng = np.random.default_rng(11)
    k, center = 5, 2
    table = rng.normal(0, 1, 4**k).astype(np.float32)
    d, m, s = _random_read(rng, table, k, center, 60000, stall=True)
    ref = SigMapRefiner(_levels_array=table, center_idx=center, scale_iters=0)
    out = ref.refine_maps([d], [400.0], [60.0], [m], [s])[0]
    assert out[0] == m[0] and out[-1] == m[-1] and np.all(np.diff(out) >= 0)
    want, err = O.refine_one(d, 400.0, 60.0, m, s, table, center, 5, "dwell_penalty", ref.sd_arr)
    assert err is None
    np.testing.assert_array_equal(out, want)


def test_refine_errors(torch_cuda, G):
    from remora_amd import RemoraError
    from remora_amd.refine_signal_map import SigMapRefiner

    bad = G["kmer_levels"].copy()
    bad[3] = np.nan
    ref = SigMapRefiner(_levels_array=bad, center_idx=2, scale_iters=0)
    with pytest.raises(RemoraError, match="NaN"):
        ref.refine_maps([G["d_dacs"]], [505.0], [83.0], [G["d_map"]], [G["d_int_seq"]])
    with pytest.raises(RemoraError):
        SigMapRefiner(do_rough_rescale=True)  # re-scaling without a table
    ref = _refiner(G, scale_iters=0)
    m = G["d_map"].copy()
    m[:] = m[0]  # no signal assigned at all
    with pytest.raises(RemoraError):
        ref.refine_maps([G["d_dacs"]], [505.0], [83.0], [m], [G["d_int_seq"]])


@pytest.mark.parametrize("hbw,force_w", [(5, "64"), (10, None), (40, None)])
def test_refine_all_three_kernels_vs_oracle(torch_cuda, O, monkeypatch, hbw, force_w):
    """half_bandwidth 5 forced onto the 64-lane kernel, 10 (21 rows per column: the 64-lane kernel by itself)
    and 40 (81 rows per column: the row-wise kernel by itself), both algorithms."""
    from remora_amd.refine_signal_map import SigMapRefiner

    if force_w:
        monkeypatch.setenv("RMR_REFINE_W", force_w)
    rng = np.random.default_rng(900 + hbw)
    k, center = 4, 2
    table = rng.normal(0, 1, 4**k).astype(np.float32)
    reads = [_random_read(rng, table, k, center, int(rng.integers(100, 500)), stall=(i == 2)) for i in range(6)]
    for algo in ("dwell_penalty", "Viterbi"):
        ref = SigMapRefiner(_levels_array=table, center_idx=center, scale_iters=0, algo=algo, half_bandwidth=hbw)
        outs, status, dev = ref._refine_batch([r[0] for r in reads], [400.0] * 6, [60.0] * 6, [r[1] for r in reads],
                                              [r[2] for r in reads])
        for i, (d, m, s) in enumerate(reads):
            want, err = O.refine_one(d, 400.0, 60.0, m, s, table, center, hbw, algo, ref.sd_arr)
            assert err is None and status[i] == 0, (i, err, status[i])
            np.testing.assert_array_equal(outs[i], want, err_msg=f"{algo} hbw {hbw} read {i}")


def test_refine_tiny_reads_and_device_pointers(torch_cuda, O):
    """Reads of 2..12 bases, and the RMR_MEM_DEVICE flavour of the entry point (torch tensors in, torch out)."""
    import ctypes

    import torch

    from remora_amd import _lib as L
    from remora_amd.refine_signal_map import SigMapRefiner

    rng = np.random.default_rng(4)
    k, center = 3, 1
    table = rng.normal(0, 1, 4**k).astype(np.float32)
    ref = SigMapRefiner(_levels_array=table, center_idx=center, scale_iters=0, half_bandwidth=2)
    reads = []
    for nb in (2, 3, 4, 5, 7, 12):
        seq = rng.integers(0, 4, nb).astype(np.int8)
        dwell = rng.integers(3, 9, nb)
        m = np.concatenate([[0], np.cumsum(dwell)]).astype(np.int64)
        d = np.round(400 + 60 * rng.standard_normal(m[-1])).astype(np.int16)
        reads.append((d, m, seq))
    outs, status, dev = ref._refine_batch([r[0] for r in reads], [400.0] * 6, [60.0] * 6, [r[1] for r in reads],
                                          [r[2] for r in reads])
    want = []
    for i, (d, m, s) in enumerate(reads):
        w, err = O.refine_one(d, 400.0, 60.0, m, s, table, center, 2, "dwell_penalty", ref.sd_arr)
        if err is None:
            assert status[i] == 0
            np.testing.assert_array_equal(outs[i], w, err_msg=f"read {i}")
        else:
            assert status[i] != 0 and dev.status_message(status[i]) == err
        want.append(w)
    # the same batch through device pointers
    n = len(reads)
    so = np.concatenate([[0], np.cumsum([r[0].size for r in reads])]).astype(np.int64)
    qo = np.concatenate([[0], np.cumsum([r[2].size for r in reads])]).astype(np.int64)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    d_d, d_so, d_m = t(np.concatenate([r[0] for r in reads])), t(so), t(np.concatenate([r[1] for r in reads]))
    d_s, d_qo, d_sh, d_sc = t(np.concatenate([r[2] for r in reads])), t(qo), t(np.full(n, 400.0)), t(np.full(n, 60.0))
    d_out, d_st = d_m.clone(), torch.zeros(n, dtype=torch.int32, device="cuda")
    p = lambda x: ctypes.c_void_p(x.data_ptr())  # noqa: E731
    L.check(L.lib().rmr_refine_signal_maps(dev._h, n, p(d_d), p(d_so), p(d_m), p(d_s), p(d_qo), p(d_sh), p(d_sc),
                                           p(d_out), p(d_st), L.MEM_DEVICE))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(d_st.cpu().numpy(), status)
    got = d_out.cpu().numpy()
    mo = qo + np.arange(n + 1)
    for i in range(n):
        if status[i] == 0:
            np.testing.assert_array_equal(got[mo[i] : mo[i + 1]], want[i])


def _plain_read(rng, nb, idx=0):
    from remora_amd.data_chunks import RemoraRead

    seq = rng.integers(0, 4, nb)
    m = np.concatenate([[0], np.cumsum(rng.integers(1 if nb > 3 else 2, 9, nb))]).astype(np.int64)
    d = np.round(500 + 80 * rng.standard_normal(m[-1])).astype(np.int16)
    return RemoraRead(dacs=d, shift=498.0 + nb % 3, scale=79.5, seq_to_sig_map=m, int_seq=seq, read_id=f"x{nb}_{idx}")


@pytest.mark.parametrize("path", ["kernel", "general"])
@pytest.mark.parametrize("method", ["least_squares", "theil_sen"])
def test_rough_rescale_device_equals_host(torch_cuda, G, method, path, monkeypatch):
    """The batched GPU-side rough re-scale (level lookup, centre samples, sorts, numpy's quantile arithmetic: the
    hand-written rmr_rescale_quantiles kernel, and the torch formulation kept for reads too long for its LDS sort)
    returns bit-identical (shift, scale) to the per-read host method, for reads shorter and longer than the
    2 x 10 clipped bases and at the powers of two of the sort, and to the reference's values for the golden reads."""
    from remora_amd.data_chunks import DeviceReads, RemoraRead
    from remora_amd.engine import get_engine
    from remora_amd.refine_signal_map import SigMapRefiner

    if path == "general":
        monkeypatch.setenv("RMR_RESCALE_GENERAL", "1")
    ref = SigMapRefiner(_levels_array=G["kmer_levels"], center_idx=int(G["center_idx"]), do_rough_rescale=True,
                        rough_rescale_method=method)
    rng = np.random.default_rng(2)
    reads = []
    for n in READS:
        reads.append(RemoraRead(dacs=G[f"{n}_dacs"], shift=505.0, scale=83.0, seq_to_sig_map=G[f"{n}_map"].copy(),
                                int_seq=G[f"{n}_int_seq"], read_id=n))
    # (2- and 3-base reads have no k-mer level at all: a zero Theil-Sen slope, which that method refuses on the host too)
    for nb in ((2, 3) if method == "least_squares" else ()) + (7, 20, 21, 22, 33, 84, 85, 900, 1044, 1045, 5000, 8212, 8213):
        reads.append(_plain_read(rng, nb))
    eng = get_engine(0)
    eng.profile_reset()
    eng.profile_enable(True)
    want = [ref.rough_rescale(r.shift, r.scale, r.seq_to_sig_map, r.int_seq, r.dacs) for r in reads]
    dr = DeviceReads(reads)
    ref.rough_rescale_device(dr, reads)
    eng.profile_enable(False)
    assert ("rescale_quantiles" in eng.profile()) == (path == "kernel")
    for r, (sh, sc) in zip(reads, want):  # (a 7-base read has too few distinct quantiles for Theil-Sen: NaN on both sides)
        np.testing.assert_array_equal(np.array([r.shift, r.scale]), np.array([sh, sc]), err_msg=r.read_id)
    np.testing.assert_array_equal(dr.shift.cpu().numpy(), [w[0] for w in want])
    if method == "theil_sen":  # setting 1 of the golden flow is exactly this (no DP pass)
        for i, n in enumerate(READS):
            np.testing.assert_allclose([reads[i].shift, reads[i].scale], G[f"s1_{n}_shift_scale"], rtol=1e-12)


def test_rough_rescale_device_long_read_takes_general_path(torch_cuda, G):
    """A read with more kept bases than the kernel sorts in LDS (16384) is reported by status and the batch is
    evaluated by the general formulation: same bits as the host method."""
    from remora_amd.data_chunks import DeviceReads
    from remora_amd.refine_signal_map import SigMapRefiner

    ref = SigMapRefiner(_levels_array=G["kmer_levels"], center_idx=int(G["center_idx"]), do_rough_rescale=True)
    rng = np.random.default_rng(5)
    reads = [_plain_read(rng, nb, i) for i, nb in enumerate((16404, 300, 16405, 40000))]
    want = [ref.rough_rescale(r.shift, r.scale, r.seq_to_sig_map, r.int_seq, r.dacs) for r in reads]
    for keep in (slice(0, 2), slice(0, 4)):  # the first pair fits the kernel exactly (16404 - 20 = 2^14), the rest does not
        sub = reads[keep]
        for r in sub:
            r.shift, r.scale = 498.0 + r.int_seq.size % 3, 79.5
        dr = DeviceReads(sub)
        ref.rough_rescale_device(dr, sub)
        for r, (sh, sc) in zip(sub, want[keep]):
            np.testing.assert_array_equal(np.array([r.shift, r.scale]), np.array([sh, sc]), err_msg=r.read_id)


def test_call_reads_mods_pipelined_with_refiner_equals_one_batch(torch_cuda, monkeypatch):
    """With a single-pass refiner (rough re-scale + one DP pass) a large batch can be walked in sub-batches by worker
    threads (inference._call_reads_mods_pipelined, opt-in: the DP is latency-bound, see call_reads_mods); calls, refined mappings and new scalings must be those of the
    one-batch path, bit for bit."""
    from remora_amd import synth
    from remora_amd.data_chunks import RemoraRead
    from remora_amd.inference import call_reads_mods
    from remora_amd.model_util import model_from_state
    from remora_amd.refine_signal_map import SigMapRefiner

    rng = np.random.default_rng(77)
    k, center = 5, 2
    table = rng.normal(0, 1, 4**k).astype(np.float32)
    base = [_random_read(rng, table, k, center, int(rng.integers(200, 900))) for i in range(40)]
    refiner = SigMapRefiner(_levels_array=table, center_idx=center, do_rough_rescale=True, scale_iters=0)
    md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"],
              can_base="C", base_start_justify=False, offset=0, sig_map_refiner=refiner)
    model = model_from_state(synth.synth_state("conv_lstm", 64, 9, 2, seed=3), md, device=0)

    def fresh():
        return [RemoraRead(dacs=base[i % 40][0], shift=400.0 + i % 3, scale=60.0, seq_to_sig_map=base[i % 40][1].copy(),
                           int_seq=base[i % 40][2], read_id=f"p{i}") for i in range(300)]

    monkeypatch.setenv("RMR_READS_SUBBATCH", "0")
    ra = fresh()
    whole = call_reads_mods(ra, model, md)
    monkeypatch.setenv("RMR_READS_SUBBATCH", "64")
    monkeypatch.setenv("RMR_READS_PIPELINE_REFINER", "1")
    rb = fresh()
    piped = call_reads_mods(rb, model, md)
    assert len(whole) == len(piped) == 300 and sum(r[2].size for r in whole) > 1000
    assert sum(not np.array_equal(x.seq_to_sig_map, base[i % 40][1]) for i, x in enumerate(ra)) > 200  # the DP moved the maps
    for a, b, x, y in zip(whole, piped, ra, rb):
        assert all(np.array_equal(p, q) for p, q in zip(a, b))
        assert (x.shift, x.scale) == (y.shift, y.scale)
        np.testing.assert_array_equal(x.seq_to_sig_map, y.seq_to_sig_map)
