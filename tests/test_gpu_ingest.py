"""The batch ingest of `infer from_pod5_and_bam` (io.iter_ingest_batches: a whole BAM batch trimmed, mapped, scaled and laid
out as read arrays on the GPU - rmr_assemble_reads - with no Python object per read) against the per-read path it replaces
(io.Read.from_pod5 + add_alignment + into_remora_read, src/remora/io.py:1972-2177, which is pinned on reference-generated
goldens): the same arrays bit for bit, the same reasons for the reads that cannot be called, the same output file."""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "golden", "data")


def _golden(name):
    return np.load(os.path.join(HERE, "golden", name), allow_pickle=False)


def _patch_tag(rec, name, value=None):
    """Record bytes with tag `name` removed (value None) or its integer value replaced."""
    raw = bytearray(rec.raw)
    for tname, s, e in rec.tag_spans:
        if tname != name:
            continue
        s, e = rec.tags_offset + s, rec.tags_offset + e
        if value is None:
            del raw[s:e]
        else:
            typ = chr(raw[s + 2])
            fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[typ]
            raw[s + 3 : e] = struct.pack(fmt, value)
        return bytes(raw)
    raise KeyError(name)


def _reparsed(raw, header):
    """The record object of record bytes (for a second _patch_tag on the same record)."""
    import tempfile

    from remora_amd import io as rio

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "one.bam")
        with rio.BamWriter(path, header) as w:
            w.write(struct.pack("<i", len(raw)) + raw)
        return next(iter(rio.iter_bam_records(path)))


def _dirty_bam(path, prefix, with_missing_moves=True, without_scaling_tags=False):
    """The reference's test alignments plus records of every kind the ingest has to turn away, in between them."""
    from remora_amd import io as rio

    src = os.path.join(DATA, f"{prefix}_mappings.bam")
    recs = list(rio.iter_bam_records(src))
    src_header = rio.read_bam_header_bytes(src)
    out = []
    for k, r in enumerate(recs):
        raw = bytes(r.raw)
        if k == 1:  # a secondary alignment: skipped
            raw = raw[:14] + struct.pack("<H", r.flag | 0x100) + raw[16:]
        elif k == 3:  # unmapped and reverse: "Unmapped reads cannot map to reverse strand."
            raw = struct.pack("<i", -1) + raw[4:14] + struct.pack("<H", r.flag | 16 | 4) + raw[16:]
        elif k == 5 and with_missing_moves:  # no move table
            raw = _patch_tag(r, "mv")
        elif k == 7:  # a read the POD5 file does not hold: skipped
            name = b"0" * (raw[8] - 1)
            raw = raw[:32] + name + raw[32 + len(name) :]
        elif k == 9:  # a trim that leaves too little signal for the move table
            raw = _patch_tag(r, "ts", 100)  # (ts is a one-byte tag here)
        if without_scaling_tags and k in (2, 6, 8, 10):  # no sm, neither, no sd, neither: median / MAD of the trimmed signal
            stripped = r
            for name in {2: ("sm",), 6: ("sm", "sd"), 8: ("sd",), 10: ("sd", "sm")}[k]:
                raw = _patch_tag(stripped, name)
                stripped = _reparsed(raw, src_header)
        out.append(raw)
        if k == 4:  # the same read twice in one batch (one decode, two alignments)
            out.append(bytes(r.raw))
    with rio.BamWriter(path, rio.read_bam_header_bytes(src)) as w:
        for raw in out:
            w.write(struct.pack("<i", len(raw)) + raw)
    return len(out)


@pytest.mark.parametrize("ref_anchored", [False, True])
@pytest.mark.parametrize("prefix", ["can", "mod"])
def test_ingest_batches_equal_the_per_read_path(prefix, ref_anchored, tmp_path):
    """Both anchors: the basecalls (move table) and the reference bases of the alignment (move table composed with the CIGAR,
    reference sequence from MD; a record without a move table sends its whole batch down the per-read path, so the file of
    this case has none)."""
    import torch

    from remora_amd import io as rio

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    pod5 = os.path.join(DATA, f"{prefix}_reads.pod5")
    bam = str(tmp_path / "dirty.bam")
    n_rec = _dirty_bam(bam, prefix, with_missing_moves=not ref_anchored)
    for pa_scaling in (None, (87.5, 14.25)):
        slow = list(rio.iter_reads_from_pod5_and_bam(pod5, bam, pa_scaling=pa_scaling, parse_ref_align=ref_anchored, decode_batch=4))
        want, want_err = [], []
        for read, err in slow:
            if err is None:
                try:
                    want.append(read.into_remora_read(ref_anchored))
                    want_err.append(None)
                    continue
                except rio.RemoraError as e:
                    err = f"Read prep error: {e}"
            want_err.append(err)
        assert len(slow) == n_rec - 2 and sum(e is not None for e in want_err) == (2 if ref_anchored else 3)
        got_err, k = [], 0
        for ib in rio.iter_ingest_batches(pod5, bam, pa_scaling=pa_scaling, batch=4, ref_anchored=ref_anchored):
            assert isinstance(ib, rio.IngestBatch)
            got_err.extend(ib.err)
            if not ib.good.size:
                continue
            dr = ib.dr
            dacs, s2s, iseq = dr.dacs.cpu().numpy(), dr.s2s.cpu().numpy(), dr.iseq.cpu().numpy()
            shift, scale = dr.shift.cpu().numpy(), dr.scale.cpu().numpy()
            assert np.array_equal(dr.d_sig_off.cpu().numpy(), dr.sig_off) and np.array_equal(dr.d_seq_off.cpu().numpy(), dr.seq_off)
            for g in range(ib.good.size):
                rr = want[k]
                k += 1
                assert np.array_equal(dacs[dr.sig_off[g] : dr.sig_off[g + 1]], rr.dacs)
                assert np.array_equal(s2s[dr.seq_off[g] + g : dr.seq_off[g + 1] + g + 1], rr.seq_to_sig_map)
                assert np.array_equal(iseq[dr.seq_off[g] : dr.seq_off[g + 1]], rr.int_seq)
                assert shift[g] == rr.shift and scale[g] == rr.scale  # the same float64 operations: equal, not close
                assert ib.seq[ib.seq_off[g] : ib.seq_off[g + 1]].decode() == rr.str_seq
                assert ib.reads[g].shift == rr.shift and ib.reads[g].scale == rr.scale
                if ref_anchored:  # the forward-strand reference bases the output record is rewritten with
                    kk = int(ib.good[g])
                    fwd = ib.ref_fwd[ib.ref_fwd_off[kk] : ib.ref_fwd_off[kk + 1]].decode()
                    rec_rev = bool(ib.rb.flag[ib.keep[kk]] & 16)
                    assert fwd == (rio.revcomp(rr.str_seq) if rec_rev else rr.str_seq)
            if ref_anchored:
                assert ib.ref_fwd_off.size == len(ib) + 1 and int(ib.ref_fwd_off[-1]) == len(ib.ref_fwd)
        assert k == len(want) and got_err == want_err


@pytest.mark.parametrize("ref_anchored", [False, True])
@pytest.mark.parametrize("prefix", ["can", "mod"])
def test_reads_without_scaling_tags_stay_on_the_batch_ingest(prefix, ref_anchored, tmp_path):
    """Records without sm / sd (either or both) are scaled by median and MAD of their trimmed signal
    (io.Read.compute_pa_to_norm_scaling, src/remora/io.py:1851-1856).  The batch ingest counts on the GPU
    (rmr_signal_histograms) and does the float64 order statistics on the occupied bins: shift and scale equal the per-read
    path's np.median values, every batch stays an IngestBatch."""
    import torch

    from remora_amd import io as rio

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    pod5 = os.path.join(DATA, f"{prefix}_reads.pod5")
    bam = str(tmp_path / "untagged.bam")
    _dirty_bam(bam, prefix, with_missing_moves=False, without_scaling_tags=True)
    untagged_seen = 0
    for rec in rio.iter_bam_records(bam):
        tags = dict(rec.tags)
        untagged_seen += not ("sm" in tags and "sd" in tags)
    assert untagged_seen == 4
    for pa_scaling in (None, (87.5, 14.25)):
        want = []
        for read, err in rio.iter_reads_from_pod5_and_bam(pod5, bam, pa_scaling=pa_scaling, parse_ref_align=ref_anchored, decode_batch=4):
            if err is None:
                try:
                    want.append(read.into_remora_read(ref_anchored))
                except rio.RemoraError:
                    pass
        k = 0
        for ib in rio.iter_ingest_batches(pod5, bam, pa_scaling=pa_scaling, batch=5, ref_anchored=ref_anchored):
            assert isinstance(ib, rio.IngestBatch)
            if not ib.good.size:
                continue
            dr = ib.dr
            dacs, shift, scale = dr.dacs.cpu().numpy(), dr.shift.cpu().numpy(), dr.scale.cpu().numpy()
            for g in range(ib.good.size):
                rr = want[k]
                k += 1
                assert np.array_equal(dacs[dr.sig_off[g] : dr.sig_off[g + 1]], rr.dacs)
                assert shift[g] == rr.shift and scale[g] == rr.scale, (k, shift[g], rr.shift, scale[g], rr.scale)
        assert k == len(want) and k >= 10


@pytest.mark.parametrize("prefix", ["can", "mod"])
def test_reference_anchored_infer_output_is_the_same_file_with_and_without_the_batch_ingest(prefix, tmp_path, monkeypatch):
    """`infer --reference-anchored`: reads anchored on the reference bases of their alignments, output records rewritten to
    `<len>M` + those bases (src/remora/inference.py:452-458) - batch ingest + native record rewrite against the per-read path,
    byte for byte, on the reference's test files and on the file with records that have to be turned away."""
    import torch

    from oracle import oracle as O
    from remora_amd import io as rio
    from remora_amd.inference import infer_from_pod5_and_bam
    from remora_amd.model_util import load_model
    from test_gpu_parity import _mint_pt, _real_reads_golden

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    model, md = load_model(_mint_pt(tmp_path, _real_reads_golden(prefix), O), device=0)
    pod5 = os.path.join(DATA, f"{prefix}_reads.pod5")
    dirty = str(tmp_path / "dirty.bam")
    _dirty_bam(dirty, prefix, with_missing_moves=False)
    for bam in (os.path.join(DATA, f"{prefix}_mappings.bam"), dirty):
        outs, stats, counts = [], [], []
        for mode in ("1", "0"):
            monkeypatch.setenv("RMR_INFER_BATCH_INGEST", mode)
            out = str(tmp_path / f"ra{mode}.bam")
            lc = {}
            stats.append(infer_from_pod5_and_bam(pod5, bam, model, md, out, reads_per_batch=5, ref_anchored=True, label_counts_out=lc))
            outs.append(open(out, "rb").read())
            counts.append({k: v.tolist() for k, v in lc.items()})
        assert stats[0] == stats[1] and counts[0] == counts[1]
        assert outs[0] == outs[1]
        assert stats[0][None] >= 10
        called = [r for r in rio.iter_bam_records(str(tmp_path / "ra1.bam")) if "MM" in dict(r.tags)]
        assert len(called) == stats[0][None] and all(len(r.cigartuples) == 1 and r.cigartuples[0][0] == 0 for r in called)
    # a limit that ends inside a batch
    for mode in ("1", "0"):
        monkeypatch.setenv("RMR_INFER_BATCH_INGEST", mode)
        infer_from_pod5_and_bam(pod5, dirty, model, md, str(tmp_path / f"rl{mode}.bam"), reads_per_batch=5, num_reads=7, ref_anchored=True)
    assert open(tmp_path / "rl1.bam", "rb").read() == open(tmp_path / "rl0.bam", "rb").read()


@pytest.mark.parametrize("prefix", ["can", "mod"])
def test_infer_output_is_the_same_file_with_and_without_the_batch_ingest(prefix, tmp_path, monkeypatch):
    import torch

    from oracle import oracle as O
    from remora_amd import io as rio
    from remora_amd.inference import infer_from_pod5_and_bam
    from remora_amd.model_util import load_model
    from test_gpu_parity import _mint_pt, _real_reads_golden

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    model, md = load_model(_mint_pt(tmp_path, _real_reads_golden(prefix), O), device=0)
    pod5 = os.path.join(DATA, f"{prefix}_reads.pod5")
    dirty = str(tmp_path / "dirty.bam")
    _dirty_bam(dirty, prefix)
    for bam, n_ok in ((os.path.join(DATA, f"{prefix}_mappings.bam"), 14), (dirty, None)):
        outs, stats, counts = [], [], []
        for mode in ("1", "0"):
            monkeypatch.setenv("RMR_INFER_BATCH_INGEST", mode)
            out = str(tmp_path / f"o{mode}.bam")
            lc = {}
            stats.append(infer_from_pod5_and_bam(pod5, bam, model, md, out, reads_per_batch=5, label_counts_out=lc))
            outs.append(open(out, "rb").read())
            counts.append({k: v.tolist() for k, v in lc.items()})
        assert outs[0] == outs[1] and stats[0] == stats[1] and counts[0] == counts[1]
        if n_ok is not None:
            assert stats[0][None] == n_ok
        else:  # the turned-away records are in the output, untouched, at their place
            assert stats[0]["Unmapped reads cannot map to reverse strand."] == 1
            assert stats[0]["Read prep error: Missing query_to_signal (move table)"] == 1
            assert stats[0]["Move table discordant with signal"] == 1
            names_in = [r.query_name for r in rio.iter_bam_records(bam) if not r.is_secondary and r.query_name != "0" * 36]
            assert [r.query_name for r in rio.iter_bam_records(str(tmp_path / "o1.bam"))] == names_in
        # a limit that ends inside a batch
        for mode in ("1", "0"):
            monkeypatch.setenv("RMR_INFER_BATCH_INGEST", mode)
            infer_from_pod5_and_bam(pod5, bam, model, md, str(tmp_path / f"l{mode}.bam"), reads_per_batch=5, num_reads=7)
        assert open(tmp_path / "l1.bam", "rb").read() == open(tmp_path / "l0.bam", "rb").read()


@pytest.mark.parametrize("ref_anchored", [False, True])
def test_several_models_share_the_batch_ingest(ref_anchored, tmp_path, monkeypatch):
    """One model per canonical base (src/remora/inference.py:286,311-315: `models[can_base]`): a C model (the golden 5mC CG one)
    and an A model with a rare motif (reads without a hit carry the C model's tags only; a batch can hold reads no model calls)
    run over the SAME resident reads of an ingest batch, their MM / ML strings joined per read in model order - the output
    file, the per-reason counts and the per-model label tallies are those of the read-by-read path."""
    import torch

    from oracle import oracle as O
    from oracle import torch_ref
    from remora_amd.inference import infer_from_pod5_and_bam
    from remora_amd.model_util import load_model, model_from_state
    from test_gpu_parity import _mint_pt, _real_reads_golden

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    model_c, md_c = load_model(_mint_pt(tmp_path, _real_reads_golden("can"), O), device=0)
    net = torch_ref.random_model("conv_lstm", 64, 9, 3, seed=31)
    md_a = dict(md_c, motifs=[("GATCA", 1)], can_base="A", mod_bases=["a", "b"], mod_long_names=["6mA", "other"], sig_map_refiner=None)
    model_a = model_from_state({k: v.numpy() for k, v in net.state_dict().items()}, md_a, device=0)
    pod5, bam = os.path.join(DATA, "can_reads.pod5"), os.path.join(DATA, "can_mappings.bam")
    outs, stats, counts = [], [], []
    for mode in ("1", "0"):
        monkeypatch.setenv("RMR_INFER_BATCH_INGEST", mode)
        out = str(tmp_path / f"m{mode}.bam")
        lc = {}
        stats.append(infer_from_pod5_and_bam(pod5, bam, [model_c, model_a], [md_c, md_a], out, reads_per_batch=5, ref_anchored=ref_anchored,
                                             label_counts_out=lc))
        outs.append(open(out, "rb").read())
        counts.append({k: np.asarray(v).tolist() for k, v in lc.items()})
    assert stats[0] == stats[1] and counts[0] == counts[1] and set(counts[0]) == {"C", "A"}
    assert outs[0] == outs[1]
    assert sum(counts[0]["A"]) > 0, "the second model called something: its tags are in the joined strings"


@pytest.mark.parametrize("scale_iters", [-1, 0])
def test_batch_ingest_with_a_signal_mapping_refiner(scale_iters, tmp_path, monkeypatch):
    """Models with a k-mer level table re-scale (and with scale_iters 0 re-map) every read before the extraction: the
    refiner's device passes work on the assembled batch as they do on an uploaded one - same output file either way."""
    import torch

    from oracle import oracle as O
    from remora_amd.inference import infer_from_pod5_and_bam
    from remora_amd.model_util import load_model
    from remora_amd.refine_signal_map import SigMapRefiner
    from test_gpu_parity import _mint_pt, _real_reads_golden

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    model, md = load_model(_mint_pt(tmp_path, _real_reads_golden("can"), O), device=0)
    md = dict(md, sig_map_refiner=SigMapRefiner(kmer_model_filename=os.path.join(DATA, "levels_4mer.txt"), do_rough_rescale=True,
                                                scale_iters=scale_iters, do_fix_guage=True))
    assert md["sig_map_refiner"].is_loaded
    pod5, bam = os.path.join(DATA, "can_reads.pod5"), os.path.join(DATA, "can_mappings.bam")
    outs, stats = [], []
    for mode in ("1", "0"):
        monkeypatch.setenv("RMR_INFER_BATCH_INGEST", mode)
        out = str(tmp_path / f"r{mode}.bam")
        stats.append(infer_from_pod5_and_bam(pod5, bam, model, md, out, reads_per_batch=5))
        outs.append(open(out, "rb").read())
    assert stats[0] == stats[1] and stats[0][None] == 14 and outs[0] == outs[1]


def test_median_mad_scaling_from_gpu_histograms_equals_numpy_on_the_samples():
    """io._median_mad_scaling (rmr_signal_histograms + float64 order statistics on the occupied bins) against
    io.Read.compute_pa_to_norm_scaling's numpy on the samples themselves (src/remora/io.py:1851-1856) - equal, not close - on
    synthetic spans: narrow ranges (counted in LDS), the full int16 range (more than 8192 bins: global atomics), one sample,
    odd and even lengths, a constant signal (MAD 0 -> scale 1.0), spans that overlap, a negative calibration scale."""
    import torch

    from remora_amd import io as rio
    from remora_amd.engine import get_ingest_engine

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    eng = get_ingest_engine(0)
    rng = np.random.RandomState(5)
    pieces = [rng.randint(300, 900, 50001), rng.randint(-32768, 32768, 70000), np.full(4000, 417), rng.randint(-5, 6, 1),
              (rng.standard_normal(30000) * 90 + 500).astype(np.int64), rng.randint(-20000, 20000, 12345)]
    flat_host = np.concatenate(pieces).astype(np.int16)
    bounds = np.concatenate([[0], np.cumsum([p.size for p in pieces])])
    start = list(bounds[:-1]) + [10, 50000, 60000]
    length = [p.size for p in pieces] + [50000, 30001, 2]
    cal_off = np.asarray([3.0, -241.0, 10.5, 0.0, -17.25, 100.0, 3.0, -1.5, 8.0])
    cal_scale = np.asarray([0.17, 0.21, 0.5, 1.0, 0.125, -0.3, 0.17, 0.19, 0.2])
    flat = torch.from_numpy(flat_host).to(eng.torch_device)
    sm, sd = rio._median_mad_scaling(flat, np.asarray(start), np.asarray(length), cal_off, cal_scale, eng)
    for i, (s0, n) in enumerate(zip(start, length)):
        pa = (flat_host[s0 : s0 + n] - float(cal_off[i])) / float(cal_scale[i])
        med = np.median(pa)
        assert sm[i] == med, (i, sm[i], med)
        assert sd[i] == max(1.0, np.median(np.abs(pa - med)) * rio.PA_TO_NORM_SCALING_FACTOR), i
    assert sd[2] == 1.0 and sd[3] == 1.0
