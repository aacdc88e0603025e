"""Parity of the HIP path (through the C ABI) against the golden vectors generated from the
reference and against the CPU oracle.  Needs a real MI355X: run with `-m gpu`."""
import json
import os

import numpy as np
import pytest

from conftest import code_to_onehot, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


def test_native_library_loaded(torch_cuda):
    from remora_amd import _lib
    from remora_amd.engine import get_engine

    assert os.path.exists(_lib.LIB_PATH)
    assert get_engine(0).handle
    assert b"gfx950" in _lib.lib().rmr_version()


# ---- E1 -----------------------------------------------------------------------------------
def test_encode_kmers_golden_bit_exact(torch_cuda):
    from remora_amd.encoded_kmers import compute_encoded_kmer_batch

    torch = torch_cuda
    g = golden("encode_kmers.npz")
    for i in range(int(g["num_cases"])):
        kb, ka, L = (int(x) for x in g[f"c{i}_args"])
        ref = code_to_onehot(g[f"c{i}_enc_code"])
        enc = compute_encoded_kmer_batch(kb, ka, g[f"c{i}_seqs"], g[f"c{i}_maps"], g[f"c{i}_lens"])
        assert enc.dtype == np.float32 and enc.shape == ref.shape
        assert np.array_equal(enc, ref), f"case {i} (host path)"
        dev = compute_encoded_kmer_batch(kb, ka, torch.from_numpy(g[f"c{i}_seqs"]).cuda(),
                                         torch.from_numpy(g[f"c{i}_maps"]).cuda(),
                                         torch.from_numpy(g[f"c{i}_lens"]).cuda())
        assert np.array_equal(dev.cpu().numpy(), ref), f"case {i} (device path)"


@pytest.mark.parametrize("cfg", ["C100", "C200"])
def test_encode_kmers_oracle_synth(torch_cuda, O, cfg):
    from remora_amd import synth
    from remora_amd.encoded_kmers import compute_encoded_kmer_batch

    d = synth.synth_chunks_config(cfg, 3000, shard=3)
    kb, ka = d["kmer_context_bases"]
    enc = compute_encoded_kmer_batch(kb, ka, d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])
    ref = O.compute_encoded_kmer_batch(kb, ka, d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])
    assert np.array_equal(enc, ref)
    # one 1.0 per (k-mer slot, signal position): SURVEY appendix A.9 checksum
    assert np.all(enc.sum(axis=(1, 2)) == (kb + ka + 1) * d["chunk_len"])


@pytest.mark.parametrize("L,max_seq,kcb", [(100, 20, (4, 4)), (200, 40, (4, 4)), (100, 20, (2, 3)), (200, 40, (2, 3)),  # unrolled stores
                                           (60, 12, (4, 4)), (300, 100, (4, 4)), (400, 200, (1, 1)), (640, 300, (4, 4))])
def test_encode_kmers_every_kernel_form(torch_cuda, O, L, max_seq, kcb, monkeypatch):
    """Every instantiation of the encode kernel against the oracle, bit for bit: rows prefetched in registers one, two
    and four elements per lane (widths up to 64 / 128 / 256), rows read where needed (wider), the store loop unrolled
    for the shipped shapes and plain for the others - and the round-2 form (RMR_ENCODE_FORM=0) on the same inputs."""
    from remora_amd import synth
    from remora_amd.encoded_kmers import compute_encoded_kmer_batch

    torch = torch_cuda
    d = synth.synth_chunks(777, chunk_len=L, max_seq_len=max_seq, kmer_context_bases=kcb, cg_context=False, shard=11)
    kb, ka = kcb
    args = (d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])
    ref = O.compute_encoded_kmer_batch(kb, ka, *args)
    dev = [torch.from_numpy(a).cuda() for a in args]
    got = compute_encoded_kmer_batch(kb, ka, *dev).cpu().numpy()
    assert got.shape == ref.shape == (777, 4 * (kb + ka + 1), L) and np.array_equal(got, ref)
    monkeypatch.setenv("RMR_ENCODE_FORM", "1")
    assert np.array_equal(compute_encoded_kmer_batch(kb, ka, *dev).cpu().numpy(), ref)
    monkeypatch.setenv("RMR_ENCODE_FORM", "0")
    assert np.array_equal(compute_encoded_kmer_batch(kb, ka, *dev).cpu().numpy(), ref)


def test_encode_kmers_full_size_checksum(torch_cuda):
    """1M C100 chunks (BASELINE config size): per-chunk sum == kmer_len * L, every value in {0,1}."""
    from remora_amd import synth
    from remora_amd.encoded_kmers import compute_encoded_kmer_batch

    torch = torch_cuda
    n = 1_000_000
    d = synth.synth_chunks_config("C100", n, shard=5)
    seqs = torch.from_numpy(d["sequence"]).cuda()
    maps = torch.from_numpy(d["sequence_to_signal_mapping"]).cuda()
    lens = torch.from_numpy(d["sequence_lengths"]).cuda()
    step = 250_000
    for st in range(0, n, step):
        enc = compute_encoded_kmer_batch(4, 4, seqs[st : st + step], maps[st : st + step], lens[st : st + step])
        sums = enc.sum(dim=(1, 2))
        assert bool((sums == 900).all())
        assert bool(((enc == 0) | (enc == 1)).all())
        # each 4-row group holds exactly one 1 per column
        assert bool((enc.view(-1, 9, 4, 100).sum(dim=2) == 1).all())
        del enc


# ---- T1 / M1 --------------------------------------------------------------------------------
def test_trim_golden(torch_cuda):
    from remora_amd.data_chunks_core import trim_sb_chunk_context_core

    g = golden("trim_chunk_context.npz")
    for i in range(int(g["num_cases"])):
        scc0, scc1, cc0, cc1, tsc = (int(x) for x in g[f"c{i}_args"])
        seqs, lens = g[f"c{i}_in_seqs"].copy(), g[f"c{i}_in_lens"].copy()
        maps = (g[f"c{i}_in_maps"] - (scc0 - cc0)).astype(np.int16)
        trim_sb_chunk_context_core(scc0, scc1, cc0, cc1, tsc, seqs, maps, lens)
        assert np.array_equal(lens, g[f"c{i}_out_lens"])
        assert np.array_equal(maps, g[f"c{i}_out_maps"])
        assert np.array_equal(seqs, g[f"c{i}_out_seqs"])


def test_parse_move_tag_golden(torch_cuda):
    from remora_amd import RemoraError
    from remora_amd.io import parse_move_tag

    g = golden("parse_move_tag.npz")
    for i in range(int(g["num_cases"])):
        mv = g[f"c{i}_mv"]
        sig_len, seq_len, rev, check = (int(x) for x in g[f"c{i}_args"])
        err = str(g[f"c{i}_err"])
        kw = dict(seq_len=None if seq_len < 0 else seq_len, check=bool(check), reverse_signal=bool(rev))
        if err:
            with pytest.raises(RemoraError, match=err):
                parse_move_tag(mv, sig_len, **kw)
        else:
            q2s, mvt, stride = parse_move_tag(mv, sig_len, **kw)
            assert np.array_equal(q2s, g[f"c{i}_q2s"]) and q2s.dtype == np.int64
            assert stride == mv[0] and np.array_equal(mvt, mv[1:])


def test_parse_move_tags_batch_golden(torch_cuda, O):
    """rmr_parse_moves_batch (one launch for the move tables of a whole batch, as the POD5+BAM ingest uses it): per
    read the golden result or the reference's error, for the golden cases grouped by (check, reverse) and for a
    ragged random batch against the oracle."""
    from remora_amd import RemoraError
    from remora_amd.io import parse_move_tags

    g = golden("parse_move_tag.npz")
    groups = {}
    for i in range(int(g["num_cases"])):
        sig_len, seq_len, rev, check = (int(x) for x in g[f"c{i}_args"])
        groups.setdefault((check, rev), []).append((i, sig_len, None if seq_len < 0 else seq_len))
    for (check, rev), cases in groups.items():
        res = parse_move_tags([g[f"c{i}_mv"] for i, _, _ in cases], [sl for _, sl, _ in cases], [ql for _, _, ql in cases],
                              check=bool(check), reverse_signal=bool(rev))
        for (i, _, _), r in zip(cases, res):
            err = str(g[f"c{i}_err"])
            if err:
                assert isinstance(r, RemoraError) and err in str(r)
            else:
                q2s, mvt, stride = r
                assert np.array_equal(q2s, g[f"c{i}_q2s"]) and q2s.dtype == np.int64
                assert stride == g[f"c{i}_mv"][0] and np.array_equal(mvt, g[f"c{i}_mv"][1:])
    rng = np.random.default_rng(8)
    for rev in (False, True):
        tags, sls, qls = [], [], []
        for n in (1, 2, 63, 64, 65, 1023, 1024, 1025, 5000, 70_001):
            mv = (rng.random(n) < 0.35).astype(np.int8)
            mv[0] = 1
            stride = int(rng.integers(1, 13))
            tags.append(np.concatenate([[stride], mv]).astype(np.int8))
            sls.append(n * stride + int(rng.integers(0, stride)))
            qls.append(int(mv.sum()))
        res = parse_move_tags(tags, sls, qls, reverse_signal=rev)
        for t, sl, ql, r in zip(tags, sls, qls, res):
            qo, _, _ = O.parse_move_tag(t, sl, seq_len=ql, reverse_signal=rev)
            assert np.array_equal(r[0], qo)
    assert parse_move_tags([], []) == []


def test_parse_move_tag_large_vs_oracle(torch_cuda, O):
    from remora_amd.io import parse_move_tag

    rng = np.random.default_rng(5)
    for rev in (False, True):
        mv = (rng.random(300_000) < 0.4).astype(np.int8)
        mv[0] = 1
        tag = np.concatenate([[6], mv]).astype(np.int8)
        sig_len = mv.size * 6 + 3
        q, _, _ = parse_move_tag(tag, sig_len, seq_len=int(mv.sum()), reverse_signal=rev)
        qo, _, _ = O.parse_move_tag(tag, sig_len, seq_len=int(mv.sum()), reverse_signal=rev)
        assert np.array_equal(q, qo)
        assert np.all(np.diff(q) > 0) and q[-1] == sig_len or rev


# ---- X1 / X2 / X3 -----------------------------------------------------------------------------
def test_extract_chunks_golden(torch_cuda):
    from remora_amd.data_chunks import RemoraRead
    from remora_amd.util import Motif

    g = golden("extract_chunks.npz")
    for rname in g["read_names"]:
        rname = str(rname)
        shift, scale = (float(x) for x in g[f"{rname}_shift_scale"])
        read = RemoraRead(dacs=g[f"{rname}_dacs"], shift=shift, scale=scale, seq_to_sig_map=g[f"{rname}_map"],
                          int_seq=g[f"{rname}_int_seq"], read_id=rname)
        read.check()
        assert np.array_equal(read.sig.view(np.uint32), g[f"{rname}_sig"].view(np.uint32)), "signal bits"
        for mname in ("CG", "C"):
            read.set_motif_focus_bases([Motif(mname, 0)])
            assert np.array_equal(read.focus_bases, g[f"{rname}_{mname}_focus"])
            for ci, cfg in enumerate(g["configs"]):
                cc, kcb, bsj, off = (int(cfg[0]), int(cfg[1])), (int(cfg[2]), int(cfg[3])), bool(cfg[4]), int(cfg[5])
                pre = f"{rname}_{mname}_c{ci}_"
                arrs = read.extract_chunk_arrays(cc, kcb, bsj, off)
                n = read.focus_bases.size
                assert len(arrs) == n
                if n == 0:
                    continue
                sig = arrs.signal.cpu().numpy().reshape(n, -1)
                assert np.array_equal(sig.view(np.uint32), g[pre + "signal"].view(np.uint32))
                sl = g[pre + "seq_len"]
                assert np.array_equal(arrs.lengths.cpu().numpy(), sl)
                seqs, maps = arrs.sequence.cpu().numpy(), arrs.mapping.cpu().numpy()
                assert seqs.shape[1] == sl.max() + sum(kcb) and maps.shape[1] == sl.max() + 1
                for i in range(n):
                    assert np.array_equal(seqs[i, : sl[i] + sum(kcb)], g[pre + "seq_w_context"][i, : sl[i] + sum(kcb)])
                    assert np.array_equal(maps[i, : sl[i] + 1], g[pre + "seq_to_sig_map"][i, : sl[i] + 1])
                    assert np.all(seqs[i, sl[i] + sum(kcb):] == -1) and np.all(maps[i, sl[i] + 1:] == 0)
                misc, geo = g[pre + "misc"], arrs.geo.cpu().numpy()
                assert np.array_equal(geo[:, 1], misc[:, 0])
                assert np.array_equal(geo[:, 2], misc[:, 1])
                assert np.array_equal(arrs.read_focus_bases.cpu().numpy(), misc[:, 2])
                # host Chunk view agrees too
                chunks = list(read.iter_chunks(cc, kcb, bsj, off))
                assert len(chunks) == n and chunks[0].seq_len == sl[0]


def test_extract_chunk_one_signal_position_golden(torch_cuda):
    """RemoraRead.extract_chunk(focus_sig_idx, chunk_context, kmer_context_bases, label, read_focus_base, check_chunk) -
    the reference's own method (src/remora/data_chunks.py:331-423), one chunk around a SIGNAL index - against the chunks the
    reference cut with it from the golden reads (iter_chunks calls it once per focus base, :455): first, last and middle
    chunk of every read / configuration, which covers both padding branches of the short reads."""
    from remora_amd import RemoraError
    from remora_amd.data_chunks import RemoraRead

    g = golden("extract_chunks.npz")
    checked = padded = 0
    for rname in g["read_names"]:
        rname = str(rname)
        shift, scale = (float(x) for x in g[f"{rname}_shift_scale"])
        smap = g[f"{rname}_map"]
        read = RemoraRead(dacs=g[f"{rname}_dacs"], shift=shift, scale=scale, seq_to_sig_map=smap, int_seq=g[f"{rname}_int_seq"],
                          read_id=rname)
        for ci, cfg in enumerate(g["configs"]):
            cc, kcb, bsj = (int(cfg[0]), int(cfg[1])), (int(cfg[2]), int(cfg[3])), bool(cfg[4])
            pre = f"{rname}_CG_c{ci}_"
            if pre + "misc" not in g:
                continue
            misc, sl = g[pre + "misc"], g[pre + "seq_len"]
            n = misc.shape[0]
            for i in sorted({0, n // 2, n - 1}):
                fb = int(misc[i, 2])
                fsig = int(smap[fb]) if bsj else int((smap[fb] + smap[fb + 1]) // 2)
                ch = read.extract_chunk(fsig, cc, kcb, label=1, read_focus_base=fb, check_chunk=True)
                assert np.array_equal(ch.signal.view(np.uint32), g[pre + "signal"][i].view(np.uint32)), (rname, ci, i)
                assert np.array_equal(ch.seq_w_context, g[pre + "seq_w_context"][i, : sl[i] + sum(kcb)])
                assert ch.seq_to_sig_map.dtype == np.int32 and np.array_equal(ch.seq_to_sig_map, g[pre + "seq_to_sig_map"][i, : sl[i] + 1])
                assert (ch.chunk_sig_focus_idx, ch.chunk_focus_base, ch.read_focus_base) == tuple(int(x) for x in misc[i, :3])
                assert ch.label == 1 and ch.read_id == rname and ch.seq_len == sl[i]
                checked += 1
                padded += int(fsig - cc[0] < 0 or fsig + cc[1] > read.dacs.size)
    assert checked >= 30 and padded >= 4
    # a signal index that is no base's centre: the window of the index itself, and the default read_focus_base of -1
    rname = str(g["read_names"][0])
    shift, scale = (float(x) for x in g[f"{rname}_shift_scale"])
    read = RemoraRead(dacs=g[f"{rname}_dacs"], shift=shift, scale=scale, seq_to_sig_map=g[f"{rname}_map"], int_seq=g[f"{rname}_int_seq"])
    mid = int(read.dacs.size // 2) + 1
    ch = read.extract_chunk(mid, (50, 50), (4, 4))
    assert np.array_equal(ch.signal.view(np.uint32), read.sig[mid - 50 : mid + 50].view(np.uint32)) and ch.read_focus_base == -1
    s0 = int(np.searchsorted(read.seq_to_sig_map, mid - 50, side="right") - 1)
    s1 = int(np.searchsorted(read.seq_to_sig_map, mid + 50, side="left"))
    assert ch.seq_len == s1 - s0 and ch.chunk_sig_focus_idx == 50 and ch.chunk_focus_base == -1 - s0
    assert np.array_equal(ch.seq_w_context[4:-4], read.int_seq[s0:s1])
    # signal_padding=True (:357-363): the mirrored signal in place of the zeros, against the reference's own chunks - padding in
    # front, behind, on both sides, none; a read too short to mirror from is refused by numpy there and here
    gp = golden("extract_chunk_padding.npz")
    ok = refused = 0
    for key in gp["cases"]:
        key = str(key)
        rname, f, c0, c1 = key.split("_")[0], int(key.split("_")[1][1:]), int(key.split("_")[2][1:]), int(key.split("_")[3])
        shift, scale = (float(x) for x in gp[f"{rname}_shift_scale"])
        rd = RemoraRead(dacs=gp[f"{rname}_dacs"], shift=shift, scale=scale, seq_to_sig_map=gp[f"{rname}_map"], int_seq=gp[f"{rname}_int_seq"],
                        read_id=rname)
        if str(gp[key + "_err"]):
            with pytest.raises(ValueError):
                rd.extract_chunk(f, (c0, c1), (4, 4), label=0, read_focus_base=5, signal_padding=True)
            refused += 1
            continue
        ch = rd.extract_chunk(f, (c0, c1), (4, 4), label=0, read_focus_base=5, signal_padding=True)
        assert np.array_equal(ch.signal.view(np.uint32), gp[key + "_signal"].view(np.uint32)), key
        assert np.array_equal(ch.seq_to_sig_map, gp[key + "_map"]) and np.array_equal(ch.seq_w_context, gp[key + "_seq"]), key
        if f - c0 >= 0 and f + c1 <= rd.dacs.size:  # nothing to pad: the plain chunk
            assert np.array_equal(rd.extract_chunk(f, (c0, c1), (4, 4), label=0, read_focus_base=5).signal, ch.signal)
        ok += 1
    assert ok >= 20 and refused == 2


def test_extract_multi_read_batch_vs_oracle(torch_cuda, O):
    from remora_amd import synth
    from remora_amd.data_chunks import RemoraRead, extract_chunk_arrays
    from remora_amd.util import Motif

    reads = []
    for i in range(6):
        r = synth.synth_read(800 + 37 * i, idx=i)
        rd = RemoraRead(dacs=r["dacs"], shift=r["shift"] + i, scale=r["scale"] - i, seq_to_sig_map=r["seq_to_sig_map"],
                        int_seq=r["int_seq"], read_id=f"r{i}")
        rd.set_motif_focus_bases([Motif("CG", 0)])
        reads.append(rd)
    arrs, sig = extract_chunk_arrays(reads, (50, 50), (4, 4))
    st = 0
    sig = sig.cpu().numpy()
    s_off = 0
    for rd in reads:
        osig = O.normalise_signal(rd.dacs, rd.shift, rd.scale)
        assert np.array_equal(sig[s_off : s_off + osig.size].view(np.uint32), osig.view(np.uint32))
        s_off += osig.size
        ch = O.extract_chunks(osig, rd.seq_to_sig_map, rd.int_seq, rd.focus_bases, (50, 50), (4, 4))
        n = rd.focus_bases.size
        sl = ch["sequence_lengths"]
        assert np.array_equal(arrs.lengths[st : st + n].cpu().numpy(), sl)
        assert np.array_equal(arrs.signal[st : st + n].cpu().numpy().view(np.uint32), ch["signal"].view(np.uint32))
        seqs, maps = arrs.sequence[st : st + n].cpu().numpy(), arrs.mapping[st : st + n].cpu().numpy()
        for i in range(n):
            assert np.array_equal(seqs[i, : sl[i] + 8], ch["sequence"][i, : sl[i] + 8])
            assert np.array_equal(maps[i, : sl[i] + 1], ch["sequence_to_signal_mapping"][i, : sl[i] + 1])
        assert np.array_equal(arrs.read_focus_bases[st : st + n].cpu().numpy(), ch["read_focus_bases"])
        st += n
    assert st == len(arrs)


# ---- F1 / F2 ------------------------------------------------------------------------------------
MODELS = ["convlstm_s64_l100_o2", "convlstm_s64_l200_o3", "convlstm_s16_l100_o2",
          "convlstm_s64_l100_k23", "conv_s64_l100_o2", "conv_s64_l100_o3",
          # any `--size` (src/remora/parsers.py:858-862): streamed-weight kernels above 64, zero-padded channels otherwise
          "convlstm_s96_l100_o2", "convlstm_s128_l100_o2", "conv_s96_l100_o2", "convlstm_s40_l100_o2", "conv_s24_l100_o3"]


def _model_from_golden(g, O):
    from remora_amd.model_util import model_from_state

    state = O.state_from_npz(g)
    size, kb, ka, L, num_out = (int(x) for x in g["params"])
    md = dict(chunk_context=(L // 2, L - L // 2), kmer_context_bases=(kb, ka))
    return model_from_state(state, md, device=0), state, (kb, ka)


@pytest.mark.parametrize("name", MODELS)
def test_fused_logits_golden(torch_cuda, O, name):
    """fp32 tolerance from the north star: <= 1e-4 on per-chunk class logits."""
    torch = torch_cuda
    g = golden(f"model_{name}.npz")
    model, state, kcb = _model_from_golden(g, O)
    out = model.infer_chunks(g["sigs"], g["seqs"], g["maps"], g["lens"], kcb)
    assert out.shape == g["logits"].shape
    assert np.abs(out - g["logits"]).max() <= 1e-4
    dev = model.infer_chunks(torch.from_numpy(g["sigs"]).cuda(), torch.from_numpy(g["seqs"]).cuda(),
                             torch.from_numpy(g["maps"]).cuda(), torch.from_numpy(g["lens"]).cuda(), kcb)
    assert np.array_equal(dev.cpu().numpy(), out), "host-staged and device paths must agree bit for bit"


@pytest.mark.parametrize("name", MODELS)
def test_dense_forward_golden(torch_cuda, O, name):
    """model(sigs, enc_kmers) contract with a materialised one-hot AND with arbitrary dense seqs."""
    torch = torch_cuda
    g = golden(f"model_{name}.npz")
    model, state, (kb, ka) = _model_from_golden(g, O)
    enc = O.compute_encoded_kmer_batch(kb, ka, g["seqs"], g["maps"], g["lens"])
    out = model(torch.from_numpy(g["sigs"]).cuda(), torch.from_numpy(enc).cuda())
    assert out.is_cuda and out.dtype == torch.float32
    assert np.abs(out.cpu().numpy() - g["logits"]).max() <= 1e-4
    out_d = model(torch.from_numpy(g["sigs"][:8]).cuda(), torch.from_numpy(g["dense_seqs"]).cuda())
    assert np.abs(out_d.cpu().numpy() - g["dense_logits"]).max() <= 1e-4
    assert next(model.parameters()).device.type == "cuda" and model.eval() is model


@pytest.mark.parametrize("arch,cfg,num_out", [("conv_lstm", "C100", 2), ("conv_lstm", "C200", 3), ("conv_only", "C100", 2)])
def test_fused_logits_oracle_synth(torch_cuda, O, arch, cfg, num_out):
    """Ragged batch sizes (not multiples of any tile) on the SURVEY §8(d) generator."""
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    torch = torch_cuda
    net = torch_ref.random_model(arch, 64, 9, num_out, seed=7)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    cc = synth.CONFIGS[cfg][0]
    model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=(4, 4)), device=0)
    for n in (1, 17, 1000, 4099):
        d = synth.synth_chunks_config(cfg, n, shard=n)
        out = model.infer_chunks(d["signal"], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"], (4, 4))
        enc = O.compute_encoded_kmer_batch(4, 4, d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])
        with torch.no_grad():
            ref = net(torch.from_numpy(d["signal"]), torch.from_numpy(enc)).numpy()
        assert np.abs(out - ref).max() <= 1e-4, (arch, cfg, n)


@pytest.mark.parametrize("arch,cfg,size,num_out", [("conv_lstm", "C100", 128, 2), ("conv_lstm", "C100", 80, 3), ("conv_lstm", "C200", 96, 3),
                                                   ("conv_lstm", "C100", 256, 2), ("conv_lstm", "C100", 100, 2), ("conv_only", "C100", 128, 2),
                                                   ("conv_only", "C100", 176, 3), ("conv_lstm", "C100", 50, 2), ("conv_only", "C100", 7, 2)])
def test_any_model_size_oracle_synth(torch_cuda, O, arch, cfg, size, num_out):
    """`--size` is any int in the reference (src/remora/parsers.py:858-862, models/ConvLSTM_w_ref.py:11-37): above 64 the
    streamed-weight kernels (k_stream.hip: every multiple of 16 up to 256, both architectures, ragged batch sizes around the
    16- and 64-chunk groups of the LSTM kernel), sizes in between through zero-weight channels.  Random networks against the
    oracle's torch.nn restatement, the north star's 1e-4; position independence and determinism over a batch larger than a grid."""
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    torch = torch_cuda
    net = torch_ref.random_model(arch, size, 9, num_out, seed=size)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    cc = synth.CONFIGS[cfg][0]
    model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=(4, 4)), device=0)
    assert model.size == size and model.kernel_size >= size and model.kernel_size % 16 == 0
    for n in (1, 33, 63, 64, 65, 1000):
        d = synth.synth_chunks_config(cfg, n, shard=n)
        out = model.infer_chunks(d["signal"], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"], (4, 4))
        enc = O.compute_encoded_kmer_batch(4, 4, d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])
        with torch.no_grad():
            ref = net(torch.from_numpy(d["signal"]), torch.from_numpy(enc)).numpy()
        assert out.shape == ref.shape and np.abs(out - ref).max() <= 1e-4, (arch, cfg, size, n, float(np.abs(out - ref).max()))
        if n == 1000:  # the dense-tensor entry (model(sigs, enc_kmers)) of the same network
            dense = model(torch.from_numpy(d["signal"]).cuda(), torch.from_numpy(enc).cuda()).cpu().numpy()
            assert np.abs(dense - ref).max() <= 1e-4
    if size in (128, 176):
        n = 40_000
        d = synth.synth_chunks_config(cfg, n, shard=3)
        dev = [torch.from_numpy(d[k]).cuda() for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
        a, b = model.infer_chunks(*dev, (4, 4)), model.infer_chunks(*dev, (4, 4))
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())
        perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
        assert torch.equal(model.infer_chunks(*[t[perm].contiguous() for t in dev], (4, 4)), a[perm])


@pytest.mark.parametrize("cc,msl,size", [((200, 200), 80, 128), ((500, 500), 60, 96), ((300, 300), 60, 256)])
def test_large_models_on_long_chunk_contexts(torch_cuda, O, cc, msl, size):
    """The reference's default chunk_context is (200, 200) (src/remora/constants.py:7-8) and its networks are length-agnostic:
    networks of more than 64 channels on chunks whose merge_conv1 input no longer fits a block's LDS go through the position
    windows of the streamed convolution (k_stream.hip) and an LSTM of several hundred steps."""
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    torch = torch_cuda
    # (a) torch's default weight scale (what a reference-initialised network looks like): north_star's fixed 1e-4 against float64;
    # (b) the amplified network (LSTM x 2.5, fc x 12: every layer's error reaches the logits, and so does fp32 rounding over
    #     several hundred steps - the oracle's OWN fp32 forward is measured against float64 and sets the scale of the allowance)
    for amplified in (False, True):
        if amplified:
            net = torch_ref.random_model("conv_lstm", size, 9, 2, seed=5)
            state = {k: v.numpy() for k, v in net.state_dict().items()}
        else:
            state = synth.synth_state("conv_lstm", size, 9, 2, seed=5, amplify=False)
            net = torch_ref.from_state(state)
        model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=(4, 4)), device=0)
        for n in (3, 70):
            d = synth.synth_chunks(n, sum(cc), msl, (4, 4), 2, True, shard=n)
            out = model.infer_chunks(d["signal"], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"], (4, 4))
            enc = O.compute_encoded_kmer_batch(4, 4, d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])
            with torch.no_grad():
                ref32 = net(torch.from_numpy(d["signal"]), torch.from_numpy(enc)).numpy()
                ref = net.double()(torch.from_numpy(d["signal"]).double(), torch.from_numpy(enc).double()).numpy()
            net.float()
            tol = 1e-4 + (3.0 * float(np.abs(ref32 - ref).max()) if amplified else 0.0)
            assert np.abs(out - ref).max() <= tol, (cc, size, n, amplified, float(np.abs(out - ref).max()), tol)
            assert out.std(axis=0).min() > (1e-3 if amplified else 1e-6)  # the chunks are told apart


@pytest.mark.parametrize("cfg,num_out", [("C100", 2), ("C200", 3)])
def test_small_batches_take_the_four_chunk_lstm_and_return_the_same_bits(torch_cuda, O, cfg, num_out):
    """Batches of up to 1024 chunks (one read per call: rmr_call_read, call_read_mods) run the LSTM four chunks per block on
    v_mfma_f32_4x4x1_16b (lstm_small_kernel, k_lstm.hip) instead of sixteen per block: the same k-ordered fmaf chains, so the
    logits of a chunk are the same bits whether it arrives alone, in a read's few hundred chunks or in a batch of thousands -
    and within 1e-4 of the oracle."""
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    torch = torch_cuda
    net = torch_ref.random_model("conv_lstm", 64, 9, num_out, seed=21)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    cc = synth.CONFIGS[cfg][0]
    model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=(4, 4)), device=0)
    d = synth.synth_chunks_config(cfg, 3000, shard=77)
    keys = ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")
    big = model.infer_chunks(*[d[k] for k in keys], (4, 4))  # sixteen chunks per block
    enc = O.compute_encoded_kmer_batch(4, 4, d["sequence"][:600], d["sequence_to_signal_mapping"][:600], d["sequence_lengths"][:600])
    with torch.no_grad():
        ref = net(torch.from_numpy(d["signal"][:600]), torch.from_numpy(enc)).numpy()
    assert np.abs(big[:600] - ref).max() <= 1e-4
    for start, n in ((0, 1), (5, 3), (8, 4), (100, 5), (300, 313), (1000, 1024), (17, 1023)):
        part = model.infer_chunks(*[d[k][start : start + n] for k in keys], (4, 4))
        assert np.array_equal(part.view(np.uint32), big[start : start + n].view(np.uint32)), (cfg, start, n, float(np.abs(part - big[start : start + n]).max()))


def test_model_sizes_the_engine_refuses(torch_cuda):
    from oracle import torch_ref
    from remora_amd import RemoraError
    from remora_amd.model_util import model_from_state

    md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4))
    st = lambda size: {k: v.numpy() for k, v in torch_ref.random_model("conv_lstm", size, 9, 2, seed=1).state_dict().items()}  # noqa: E731
    with pytest.raises(RemoraError, match="size 1..256"):
        model_from_state(st(272), md, device=0)
    for dtype in ("f16x3", "bf16x3", "bf16x6"):  # the split dtypes stop at 64 channels (bf16 / f16 stream their weights above: k_stream16.hip)
        with pytest.raises(RemoraError, match="split dtypes stop at 64"):
            model_from_state(st(96), md, device=0, dtype=dtype)
    # a padded size in the 16-bit pipeline: 40 channels run the fused kernels at 64
    m = model_from_state(st(40), md, device=0, dtype="f16")
    assert m.kernel_size == 64


def test_full_size_properties(torch_cuda, O):
    """BASELINE config 3 size (1M C100 chunks, ConvLSTM_w_ref fp32): determinism, permutation
    equivariance across sub-batch / tile boundaries, exact label tally, and a 20k-chunk sample
    against the CPU restatement of the reference network."""
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    torch = torch_cuda
    n = 1_000_000
    net = torch_ref.random_model("conv_lstm", 64, 9, 2, seed=0)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    model = model_from_state(state, dict(chunk_context=(50, 50), kmer_context_bases=(4, 4)), device=0)
    d = synth.synth_chunks_config("C100", n)
    dev = [torch.from_numpy(d[k]).cuda() for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
    counts = torch.zeros(2, dtype=torch.int64, device="cuda")
    out = model.infer_chunks(*dev, (4, 4), label_counts=counts)
    out2 = model.infer_chunks(*dev, (4, 4))
    torch.cuda.synchronize()
    assert torch.equal(out, out2), "non-deterministic"
    assert bool(torch.isfinite(out).all())
    host_counts = torch.bincount(out.argmax(dim=1), minlength=2)
    assert torch.equal(counts, host_counts) and int(counts.sum()) == n
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    outp = model.infer_chunks(*[t[perm].contiguous() for t in dev], (4, 4))
    assert torch.equal(outp, out[perm]), "result of a chunk depends on its batch position"
    idx = np.sort(np.random.default_rng(0).choice(n, 20000, replace=False))
    enc = O.compute_encoded_kmer_batch(4, 4, d["sequence"][idx], d["sequence_to_signal_mapping"][idx], d["sequence_lengths"][idx])
    with torch.no_grad():
        ref = net(torch.from_numpy(d["signal"][idx]), torch.from_numpy(enc)).numpy()
    assert np.abs(out[torch.from_numpy(idx).cuda()].cpu().numpy() - ref).max() <= 1e-4


def test_count_labels_golden(torch_cuda):
    import ctypes

    from remora_amd import _lib as L
    from remora_amd.engine import get_engine

    g = golden("post_process.npz")
    logits = np.ascontiguousarray(g["tally_logits"])
    counts = np.zeros(3, np.int64)
    L.check(L.lib().rmr_count_labels(get_engine(0).handle, logits.ctypes.data, logits.shape[0], 3, counts.ctypes.data, L.MEM_HOST))
    assert np.array_equal(counts, g["tally_pred_counts"])


# ---- boundary: load_model + call_read_mods -----------------------------------------------------------
def _mint_pt(tmp_path, g, O):
    """TorchScript file in the reference's format (model_util.py:115-176): the oracle's torch
    restatement scripted + the reference-written meta.txt string from the golden file."""
    import torch
    from oracle import torch_ref

    net = torch_ref.from_state(O.state_from_npz(g))
    pt = str(tmp_path / "model.pt")
    torch.jit.save(torch.jit.script(net), pt, _extra_files={"meta.txt": str(g["meta_txt"])})
    return pt


def test_load_model_dtype_from_the_environment(torch_cuda, O, tmp_path, monkeypatch):
    """REMORA_HIP_DTYPE: the arithmetic of a model loaded through the reference's load_model signature, which has no dtype
    argument (src/remora/model_util.py:566-578) - a caller that cannot pass one selects the 16-bit pipeline this way; an
    explicit dtype= wins."""
    from remora_amd.model_util import load_model, load_torchscript_model

    pt = _mint_pt(tmp_path, golden("call_read_mods_cg_5mc.npz"), O)
    assert load_model(pt, device=0, quiet=True, eval_only=True)[0].dtype == "fp32"
    monkeypatch.setenv("REMORA_HIP_DTYPE", "f16")
    assert load_model(pt, device=0, quiet=True, eval_only=True)[0].dtype == "f16"
    assert load_torchscript_model(pt, device=0, dtype="bf16")[0].dtype == "bf16"


@pytest.mark.parametrize("name", ["cg_5mc", "allc_5hmc_5mc", "conv_cg", "cg_5mc_refine"])
def test_load_model_and_call_read_mods_golden(torch_cuda, O, tmp_path, name):
    """cg_5mc_refine carries a k-mer level table (base_start_justify, offset 1): call_read_mods then
    re-scales and re-maps every read (rough re-scale + dwell-penalty DP on the GPU) before extraction."""
    from remora_amd.data_chunks import RemoraRead
    from remora_amd.inference import call_read_mods
    from remora_amd.model_util import load_model

    g = golden(f"call_read_mods_{name}.npz")
    model, md = load_model(_mint_pt(tmp_path, g, O), device=0, quiet=True, eval_only=True)
    ref_md = json.loads(str(g["derived_md_json"]))
    for k, v in ref_md.items():
        got = md[k]
        if isinstance(v, list):
            got = json.loads(json.dumps(got))
        assert got == v, (k, got, v)
    assert md["sig_map_refiner"].is_loaded is name.endswith("_refine")
    for rname in g["read_names"]:
        rname = str(rname)
        shift, scale = (float(x) for x in g[f"{rname}_shift_scale"])

        def mk():
            return RemoraRead(dacs=g[f"{rname}_dacs"].copy(), shift=shift, scale=scale,
                              seq_to_sig_map=g[f"{rname}_map"].copy(), int_seq=g[f"{rname}_int_seq"].copy(), read_id=rname)

        nn_out, labels, pos = call_read_mods(mk(), model, md)
        assert np.array_equal(pos, g[f"{rname}_pos"]), "chunk order must follow the reference's set order"
        assert np.array_equal(labels, g[f"{rname}_labels"])
        if pos.size == 0:
            assert nn_out.size == 0
            res = call_read_mods(mk(), model, md, return_mm_ml_tags=True)
            assert len(res) == 3 and str(g[f"{rname}_mm"]) == "<EMPTY3>"
            continue
        assert nn_out.dtype == np.float32 and pos.dtype == np.int64
        assert np.abs(nn_out - g[f"{rname}_nn_out"]).max() <= 1e-4
        probs, _, _ = call_read_mods(mk(), model, md, return_mod_probs=True)
        assert probs.dtype == np.float64 and np.abs(probs - g[f"{rname}_probs"]).max() <= 1e-4
        mm, ml = call_read_mods(mk(), model, md, return_mm_ml_tags=True)
        assert mm == str(g[f"{rname}_mm"])
        ml = np.asarray(list(ml), np.uint8).astype(int)
        assert np.abs(ml - g[f"{rname}_ml"].astype(int)).max() <= 1  # floor(p*256) next to a bin edge
    if "r_long_focus_offset" not in g:
        return
    fo = int(g["r_long_focus_offset"])
    shift, scale = (float(x) for x in g["r_long_shift_scale"])
    rd = RemoraRead(dacs=g["r_long_dacs"], shift=shift, scale=scale, seq_to_sig_map=g["r_long_map"], int_seq=g["r_long_int_seq"])
    o2, _, p2 = call_read_mods(rd, model, md, focus_offset=fo)
    assert np.array_equal(p2, g["r_long_focus_pos"]) and np.abs(o2 - g["r_long_focus_nn_out"]).max() <= 1e-4


def test_prepare_batches_reference_tuple(torch_cuda):
    """read.batches elements unpack to (signal, enc_kmers, labels, read_focus_bases) like the reference's."""
    from remora_amd.data_chunks import RemoraRead
    from remora_amd.util import Motif

    g = golden("prepare_batches.npz")
    shift, scale = (float(x) for x in g["shift_scale"])
    read = RemoraRead(dacs=g["dacs"], shift=shift, scale=scale, seq_to_sig_map=g["map"], int_seq=g["int_seq"])
    read.set_motif_focus_bases([Motif("CG", 0)])
    md = dict(chunk_context=[50, 50], kmer_context_bases=[4, 4], base_start_justify=False, offset=0, sig_map_refiner=None)
    read.prepare_batches(md, 2048)
    assert len(read.batches) == 1
    sig, enc, labels, rfb = read.batches[0]
    assert np.array_equal(sig.view(np.uint32), g["signal"].view(np.uint32))
    assert np.array_equal(enc, code_to_onehot(g["enc_code"]))
    assert np.array_equal(labels, g["labels"]) and np.array_equal(rfb, g["read_focus_bases"])


def test_errors_surface_as_remora_error(torch_cuda):
    from remora_amd import RemoraError
    from remora_amd.model_util import load_model, model_from_state

    with pytest.raises(RemoraError, match="not found"):
        load_model("/nonexistent/model.pt")
    with pytest.raises(RemoraError):
        load_model(None)
    from oracle import torch_ref

    net = torch_ref.random_model("conv_lstm", 64, 9, 2)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    model = model_from_state(state, dict(chunk_context=(50, 50)), device=0)
    import torch

    with pytest.raises(RemoraError, match="do not match"):
        model(torch.zeros(2, 1, 90).cuda(), torch.zeros(2, 36, 90).cuda())
    with pytest.raises(RemoraError, match="kmer"):
        model.infer_chunks(np.zeros((2, 1, 100), np.float32), np.zeros((2, 24), np.int8), np.zeros((2, 21), np.int16),
                           np.ones(2, np.int16), (2, 3))


# ---- bf16-MFMA modes (split operands): own tolerances ---------------------------------------------
# bf16x6 is fp32-class (as close to a float64 evaluation as the fp32-MFMA path, tests/manual/precision_vs_float64.py) and
# holds the fp32 gate of 1e-4; bf16x3 (two-part operands, ~2^-16 relative) is a 5e-4 mode: 1.4e-4 measured on 8192
# synthetic chunks, 1.8e-4 over 1 M - it does NOT meet 1e-4 and does not claim to; plain bf16 and f16 are the 16-bit
# pipelines, gated in tests/test_gpu_fused.py against the float64 network and the emulated 16-bit arithmetic
# f16x3 (round 4): two IEEE-half parts, three products - 22 significand bits, the fp32 gate at bf16x3's cost
SPLIT_TOL = {"bf16x6": 1e-4, "f16x3": 1e-4, "bf16x3": 5e-4, "bf16": 3e-2, "f16": 4e-3}


@pytest.mark.parametrize("dtype", ["bf16x6", "f16x3", "bf16x3", "bf16", "f16"])
@pytest.mark.parametrize("name", ["convlstm_s64_l100_o2", "convlstm_s64_l200_o3", "convlstm_s64_l100_k23"])
def test_split_bf16_logits_golden(torch_cuda, O, name, dtype):
    """The reference-generated logits under every reduced-precision mode: bf16x6 within the fp32 gate (1e-4), bf16x3
    within 5e-4, plain bf16 within 3e-2 and f16 within 4e-3 on these 24-48 chunks, each with argmax agreement on the chunks
    whose margin exceeds four tolerances."""
    from remora_amd.model_util import model_from_state

    g = golden(f"model_{name}.npz")
    state = O.state_from_npz(g)
    size, kb, ka, L, num_out = (int(x) for x in g["params"])
    md = dict(chunk_context=(L // 2, L - L // 2), kmer_context_bases=(kb, ka))
    model = model_from_state(state, md, device=0, dtype=dtype)
    out = model.infer_chunks(g["sigs"], g["seqs"], g["maps"], g["lens"], (kb, ka))
    err = np.abs(out - g["logits"]).max()
    assert err <= SPLIT_TOL[dtype], (dtype, err)
    ref = g["logits"]
    srt = np.sort(ref, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 4 * SPLIT_TOL[dtype]
    assert np.array_equal(out.argmax(1)[clear], ref.argmax(1)[clear])


def test_f16x3_is_fp32_class_on_the_synthetic_benchmark_network(torch_cuda, O):
    """dtype f16x3 on 8192 chunks of the amplified synthetic network (the one bench.py runs) against a
    float64 evaluation of the same network: max |logit error| <= 2e-5 (measured 6e-6: the class of the fp32 path and of
    bf16x6), and the calls of the fp32 path wherever the float64 margin exceeds 1e-4."""
    import torch

    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    cc, kcb, _, num_out, _ = synth.CONFIGS["C100"]
    state = synth.synth_state("conv_lstm", 64, 9, num_out, seed=0)
    d = synth.synth_chunks_config("C100", 8192, shard=3)
    args = (d["signal"], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"], kcb)
    md = dict(chunk_context=cc, kmer_context_bases=kcb)
    got = np.array(model_from_state(state, md, device=0, dtype="f16x3").infer_chunks(*args))
    fp32 = np.array(model_from_state(state, md, device=0).infer_chunks(*args))
    enc = O.compute_encoded_kmer_batch(kcb[0], kcb[1], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])
    with torch.no_grad():
        ref = torch_ref.from_state(state).double()(torch.from_numpy(d["signal"]).double(), torch.from_numpy(enc).double()).numpy()
    assert np.abs(got - ref).max() <= 2e-5, np.abs(got - ref).max()
    srt = np.sort(ref, axis=1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-4
    assert clear.mean() > 0.99 and np.array_equal(got.argmax(1)[clear], fp32.argmax(1)[clear])


def test_split_bf16_rejects_unsupported(torch_cuda, O):
    from remora_amd import RemoraError
    from remora_amd.model_util import model_from_state

    g = golden("model_conv_s64_l100_o2.npz")
    with pytest.raises(RemoraError):
        model_from_state(O.state_from_npz(g), dict(chunk_context=(50, 50)), device=0, dtype="bf16x6")
    with pytest.raises(RemoraError):
        model_from_state(O.state_from_npz(g), dict(chunk_context=(50, 50)), device=0, dtype="fp8")


# ---- BASELINE configs[0] shape: the reference's own POD5 + BAM test data end to end ------------------
def _real_reads_golden(prefix):
    """Per-read results of the reference on tests/data/<prefix>_reads.pod5 + <prefix>_mappings.bam; the CG 5mC model's
    weights travel once, in the `can` file (tools/gen_golden.py: the same network calls both halves of configs[0])."""
    class G(dict):
        files = property(lambda self: list(self.keys()))

    g = G(golden(f"real_reads_{prefix}.npz"))
    if prefix != "can":
        w = golden("real_reads_can.npz")
        g.update({k: w[k] for k in w.files if k.startswith("w__")})
    return g


@pytest.mark.parametrize("prefix,n_chunks", [("can", 922), ("mod", 1264)])
def test_real_reads_pod5_bam_end_to_end(torch_cuda, O, tmp_path, prefix, n_chunks):
    """BASELINE configs[0], both halves: tests/data/{can,mod}_reads.pod5 + {can,mod}_mappings.bam (copied under
    tests/golden/data) -> remora_amd.io ingest -> Read.add_alignment -> into_remora_read -> call_read_mods, against the
    reference's own Read / call_read_mods run on the same parsed records (tools/gen_golden.py)."""
    from remora_amd import io as rio
    from remora_amd.inference import call_read_mods
    from remora_amd.model_util import load_model

    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
    g = _real_reads_golden(prefix)
    model, md = load_model(_mint_pt(tmp_path, g, O), device=0)
    n = 0
    for i, (read, err) in enumerate(rio.iter_reads_from_pod5_and_bam(os.path.join(data, f"{prefix}_reads.pod5"),
                                                                     os.path.join(data, f"{prefix}_mappings.bam"))):
        assert err is None and read.read_id == str(g[f"r{i}_name"])
        rr = read.into_remora_read(False)
        assert [rr.shift, rr.scale] == list(g[f"r{i}_shift_scale"]), "scaling composition must match bit for bit"
        assert rr.dacs.size == int(g[f"r{i}_ndacs"])
        crc = int(np.bitwise_xor.reduce(rr.dacs.astype(np.int64) * (np.arange(rr.dacs.size) % 251 + 1)))
        assert crc == int(g[f"r{i}_dacs_crc"])
        assert np.array_equal(rr.seq_to_sig_map, g[f"r{i}_map"]) and rr.str_seq == str(g[f"r{i}_seq"])
        nn_out, labels, pos = call_read_mods(rr, model, md)
        assert np.array_equal(pos, g[f"r{i}_pos"])
        assert np.abs(nn_out - g[f"r{i}_nn_out"]).max() <= 1e-4
        mm, ml = call_read_mods(read.into_remora_read(False), model, md, return_mm_ml_tags=True)
        assert mm == str(g[f"r{i}_mm"])
        assert np.abs(np.asarray(list(ml), np.uint8).astype(int) - g[f"r{i}_ml"].astype(int)).max() <= 1
        n += pos.size
    assert i == 13 and n == n_chunks


@pytest.mark.parametrize("prefix", ["can", "mod"])
def test_add_alignment_branches_the_test_files_never_take(torch_cuda, O, tmp_path, prefix):
    """Read.add_alignment / into_remora_read on the branches no record of tests/data exercises — reverse_signal (the double
    flip around the sp / ts / ns trims and the reversed move table, src/remora/io.py:1995-2010), no sm / sd tags (median /
    MAD scaling, :1851-1856, :2035-2040), split reads (sp + pi, the child's id, and the mismatch error, :2001-2020),
    pa_scaling (:442-461, :2159-2167) — against the reference's own add_alignment run on the same records through the same
    edited tags (tests/golden_util.py: real_read_branch; tools/gen_golden.py: gen_real_read_branches)."""
    from golden_util import REAL_READ_BRANCHES, RecordWithTags, pod5_reads_cpu, real_read_branch
    from remora_amd import RemoraError
    from remora_amd import io as rio
    from remora_amd.inference import call_read_mods
    from remora_amd.model_util import load_model

    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
    g = golden("real_read_branches.npz")
    model, md = load_model(_mint_pt(tmp_path, _real_reads_golden("can"), O), device=0)
    pods = {p.read_id: p for p in pod5_reads_cpu(os.path.join(data, f"{prefix}_reads.pod5"))}
    recs = list(rio.iter_bam_records(os.path.join(data, f"{prefix}_mappings.bam")))
    assert len(recs) == 14
    for i, rec in enumerate(recs):
        pod = pods[rec.query_name]
        for variant in REAL_READ_BRANCHES:
            read_id, dacs, rec_v, kw, _ = real_read_branch(variant, pod, rec, 1000 + i)
            read = rio.Read(read_id=read_id, dacs=dacs, shift_dacs_to_pa=pod.calibration_offset, scale_dacs_to_pa=pod.calibration_scale)
            read.add_alignment(rec_v, parse_ref_align=False, **kw)
            rr = read.into_remora_read(False)
            key = f"{prefix}_r{i}_{variant}"
            assert [rr.shift, rr.scale] == list(g[f"{key}_shift_scale"]), key
            assert rr.dacs.size == int(g[f"{key}_ndacs"]), key
            assert int(np.bitwise_xor.reduce(rr.dacs.astype(np.int64) * (np.arange(rr.dacs.size) % 251 + 1))) == int(g[f"{key}_dacs_crc"]), key
            assert np.array_equal(rr.seq_to_sig_map, g[f"{key}_map"]), key
            assert [read.read_id, read.child_read_id] == [str(x) for x in g[f"{key}_read_ids"]], key
            nn_out, labels, pos = call_read_mods(rr, model, md)
            assert np.array_equal(pos, g[f"{key}_pos"]), key
            assert np.abs(nn_out - g[f"{key}_nn_out"]).max() <= 1e-4, key
    pod, rec = pods[recs[0].query_name], recs[0]
    read = rio.Read(read_id=pod.read_id, dacs=pod.signal, shift_dacs_to_pa=pod.calibration_offset, scale_dacs_to_pa=pod.calibration_scale)
    with pytest.raises(RemoraError) as ei:
        read.add_alignment(RecordWithTags(rec, add={"pi": "somebody-else"}), parse_ref_align=False)
    assert str(ei.value) == str(g[f"{prefix}_split_mismatch_error"])
    # the batched ingest takes the same branches (reverse_signal through the batched move-table expansion)
    for i, (read, err) in enumerate(rio.iter_reads_from_pod5_and_bam(os.path.join(data, f"{prefix}_reads.pod5"),
                                                                     os.path.join(data, f"{prefix}_mappings.bam"),
                                                                     reverse_signal=True, parse_ref_align=False)):
        assert err is None
        assert np.array_equal(read.into_remora_read(False).seq_to_sig_map, g[f"{prefix}_r{i}_rev_map"])


def test_batched_call_reads_mods_matches_single_read_api(torch_cuda, O):
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.data_chunks import RemoraRead
    from remora_amd.inference import call_read_mods, call_reads_mods
    from remora_amd.model_util import model_from_state

    net = torch_ref.random_model("conv_lstm", 64, 9, 2, seed=3)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"],
              mod_long_names=["5mC"], can_base="C", base_start_justify=False, offset=0, sig_map_refiner=None)
    model = model_from_state(state, md, device=0)

    def mk(i, nb):
        r = synth.synth_read(nb, idx=i)
        return RemoraRead(dacs=r["dacs"], shift=r["shift"], scale=r["scale"], seq_to_sig_map=r["seq_to_sig_map"],
                          int_seq=r["int_seq"], read_id=f"r{i}")

    sizes = [700, 5, 1200, 64, 3000]
    sizes_no_cg = np.array([0, 0, 3, 3, 0, 0])  # a read without any CG
    reads = [mk(i, nb) for i, nb in enumerate(sizes)]
    reads.insert(2, RemoraRead(dacs=np.full(60, 500, np.int16), shift=500.0, scale=80.0,
                               seq_to_sig_map=np.arange(0, 61, 10), int_seq=sizes_no_cg, read_id="nocg"))
    batched = call_reads_mods(reads, model, md)
    assert len(batched) == len(reads) and batched[2][0].size == 0
    for rd, (o, l, p) in zip(reads, batched):
        so, sl, sp = call_read_mods(rd.copy(), model, md)
        order = np.argsort(sp, kind="stable")
        assert np.array_equal(p, sp[order])
        if p.size:
            assert np.array_equal(o, so[order]), "same chunks must give bit-identical logits in any batch"


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "f16x3"])
def test_native_call_read_equals_prepare_batches_and_run_model(torch_cuda, O, monkeypatch, dtype):
    """call_read_mods through the ONE native entry (rmr_call_read: staging, normalisation, geometry, rows, network, two
    stream synchronisations) against the two-step path of the reference's shape (RemoraRead.prepare_batches + run_model,
    src/remora/data_chunks.py:468-540): the same bits, positions and labels - for reads shorter than a chunk (both padding
    branches), with N bases, int8 / int32 bases, labels, an offset, base_start_justify, a single focus_offset, no motif hit;
    and the fp32 logits against the CPU oracle."""
    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.data_chunks import RemoraRead
    from remora_amd.inference import call_read_mods
    from remora_amd.model_util import model_from_state

    net = torch_ref.random_model("conv_lstm", 64, 9, 2, seed=5)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    rng = np.random.default_rng(11)

    def mk(i, nb, seq_dtype=np.int64, n_bases=0, labels=False):
        r = synth.synth_read(nb, idx=i)
        seq = r["int_seq"].astype(seq_dtype)
        if n_bases:
            seq[rng.choice(nb, n_bases, replace=False)] = -1
        return RemoraRead(dacs=r["dacs"], shift=r["shift"] + i, scale=r["scale"] - i, seq_to_sig_map=r["seq_to_sig_map"], int_seq=seq,
                          read_id=f"r{i}", labels=rng.integers(0, 2, nb) if labels else None)

    reads = [mk(0, 900), mk(1, 6), mk(2, 40, np.int8), mk(3, 2500, np.int32, n_bases=60), mk(4, 700, labels=True), mk(5, 5000),
             RemoraRead(dacs=np.full(60, 500, np.int16), shift=500.0, scale=80.0, seq_to_sig_map=np.arange(0, 61, 10),
                        int_seq=np.array([0, 0, 3, 3, 0, 0]), read_id="nocg")]
    for bsj, offset in ((False, 0), (True, 1), (False, -2)):
        md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"],
                  can_base="C", base_start_justify=bsj, offset=offset, sig_map_refiner=None)
        model = model_from_state(state, md, device=0, dtype=dtype)
        for rd in reads:
            monkeypatch.setenv("RMR_NATIVE_CALL_READ", "1")
            a = call_read_mods(rd.copy(), model, md)
            monkeypatch.setenv("RMR_NATIVE_CALL_READ", "0")
            b = call_read_mods(rd.copy(), model, md)
            assert len(a) == len(b) == 3 and a[0].shape == b[0].shape, rd.read_id
            for x, y in zip(a, b):
                assert np.array_equal(x, y), (rd.read_id, bsj, offset)
            if rd.labels is not None:
                assert (a[1] >= 0).all()
            if rd.int_seq.size > 100:
                monkeypatch.setenv("RMR_NATIVE_CALL_READ", "1")
                fo = int(rd.int_seq.size // 2)
                a1 = call_read_mods(rd.copy(), model, md, focus_offset=fo)
                monkeypatch.setenv("RMR_NATIVE_CALL_READ", "0")
                b1 = call_read_mods(rd.copy(), model, md, focus_offset=fo)
                assert a1[0].shape == (1, 2) and all(np.array_equal(x, y) for x, y in zip(a1, b1))
        if dtype == "fp32" and not bsj and offset == 0:
            monkeypatch.setenv("RMR_NATIVE_CALL_READ", "1")
            rd = reads[0]
            out, _, pos = call_read_mods(rd.copy(), model, md)
            o_out, _, o_pos = O.call_read_mods(rd.dacs, rd.shift, rd.scale, rd.seq_to_sig_map, rd.int_seq, state, md)
            assert np.array_equal(pos, o_pos) and np.abs(out - o_out).max() <= 1e-4
        del model


def test_fused_edge_cases_vs_oracle(torch_cuda, O):
    """One base per sample (max_seq_len == chunk_len), zero-dwell bases, missing (-1) bases,
    garbage in the padding columns, wide arrays, and empty batches."""
    from oracle import torch_ref
    from remora_amd.model_util import model_from_state

    torch = torch_cuda
    net = torch_ref.random_model("conv_lstm", 64, 9, 2, seed=11)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    model = model_from_state(state, dict(chunk_context=(50, 50), kmer_context_bases=(4, 4)), device=0)
    rng = np.random.default_rng(3)
    n, L, msl = 37, 100, 100
    seqs = rng.integers(-1, 4, (n, msl + 8 + 5)).astype(np.int8)      # 5 extra garbage columns
    maps = rng.integers(-50, 150, (n, msl + 1 + 3)).astype(np.int16)  # garbage everywhere first
    lens = np.zeros(n, np.int16)
    for c in range(n):
        sl = [100, 1, 2, 57][c % 4] if c < 8 else int(rng.integers(1, 101))
        cuts = np.sort(rng.integers(0, L + 1, sl - 1)) if c % 3 else np.sort(rng.choice(np.arange(1, L), sl - 1, replace=False))
        maps[c, : sl + 1] = np.concatenate([[0], cuts, [L]])
        lens[c] = sl
    sig = rng.standard_normal((n, 1, L)).astype(np.float32)
    out = model.infer_chunks(sig, seqs, maps, lens, (4, 4))
    enc = O.compute_encoded_kmer_batch(4, 4, seqs, maps, lens)
    with torch.no_grad():
        ref = net(torch.from_numpy(sig), torch.from_numpy(enc)).numpy()
    assert np.abs(out - ref).max() <= 1e-4
    # the standalone encode agrees bit for bit on the same awkward rows
    from remora_amd.encoded_kmers import compute_encoded_kmer_batch

    assert np.array_equal(compute_encoded_kmer_batch(4, 4, seqs, maps, lens), enc)
    # empty batch: shapes preserved, nothing launched
    e = model.infer_chunks(sig[:0], seqs[:0], maps[:0], lens[:0], (4, 4))
    assert e.shape == (0, 2)
    assert model(torch.zeros(0, 1, 100).cuda(), torch.zeros(0, 36, 100).cuda()).shape == (0, 2)


def test_concurrent_calls_from_threads(torch_cuda, O):
    """The reference calls the model from several Python threads (src/remora/inference.py:973-982);
    the engine serialises calls internally, results must be unaffected."""
    import threading

    from oracle import torch_ref
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    net = torch_ref.random_model("conv_lstm", 64, 9, 2, seed=5)
    state = {k: v.numpy() for k, v in net.state_dict().items()}
    model = model_from_state(state, dict(chunk_context=(50, 50), kmer_context_bases=(4, 4)), device=0)
    datas = [synth.synth_chunks_config("C100", 3000 + 101 * i, shard=40 + i) for i in range(4)]
    args = [(d["signal"], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"]) for d in datas]
    expect = [model.infer_chunks(*a, (4, 4)) for a in args]
    got = [[None] * 6 for _ in args]

    def work(i):
        for rep in range(6):
            got[i][rep] = model.infer_chunks(*args[i], (4, 4))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(args))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for i in range(len(args)):
        for rep in range(6):
            assert np.array_equal(got[i][rep], expect[i])


@pytest.mark.parametrize("prefix", ["can", "mod"])
def test_infer_from_pod5_and_bam_cli(torch_cuda, O, tmp_path, prefix):
    """`python -m remora_amd infer from_pod5_and_bam` on the reference's test data (both halves of BASELINE configs[0]):
    every input record comes back with the MM/ML tags the reference's call_read_mods produces for that read."""
    import subprocess
    import sys

    from remora_amd import io as rio

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = os.path.join(root, "tests", "golden", "data")
    g = _real_reads_golden(prefix)
    pt = _mint_pt(tmp_path, g, O)
    out = str(tmp_path / "out.bam")
    res = subprocess.run([sys.executable, "-m", "remora_amd", "infer", "from_pod5_and_bam", os.path.join(data, f"{prefix}_reads.pod5"),
                          os.path.join(data, f"{prefix}_mappings.bam"), "--model", pt, "--out-bam", out, "--reads-per-batch", "5"],
                         cwd=root, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "called 14 reads" in res.stdout
    recs = list(rio.iter_bam_records(out))
    assert len(recs) == 14
    for i, rec in enumerate(recs):
        assert rec.query_name == str(g[f"r{i}_name"])
        assert rec.get_tag("MM") == str(g[f"r{i}_mm"])
        ml = np.asarray(list(rec.get_tag("ML")), int)
        assert np.abs(ml - g[f"r{i}_ml"].astype(int)).max() <= 1
        assert "mv" in dict(rec.tags)  # input tags are carried over


@pytest.mark.parametrize("tag", ["stored", "trim", "trimcc", "trimk"])
def test_core_dataset_on_disk(torch_cuda, O, tag):
    """Memory-mapped CoreRemoraDataset directory written by the reference: rows (after the dynamic
    k-mer / chunk-context trimming, which runs the T1 kernel) must encode to exactly what the
    reference's own iteration produced, and the fused path must agree with the oracle forward."""
    from oracle import torch_ref
    from remora_amd.data_chunks import CoreRemoraDataset, validate_dataset
    from remora_amd.encoded_kmers import compute_encoded_kmer_batch
    from remora_amd.model_util import model_from_state

    torch = torch_cuda
    g = golden("core_dataset.npz")
    override = json.loads(str(g[f"{tag}_override"]))
    ddir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data", "core_dataset")
    ds = CoreRemoraDataset(ddir, override_metadata=override, batch_size=64, infinite_iter=False)
    assert ds.size == int(g["num_chunks"]) == 120
    sig, code, labs = [], [], []
    for b in ds.iter_batches():
        enc = compute_encoded_kmer_batch(*ds.kmer_context_bases, b["sequence"], b["sequence_to_signal_mapping"], b["sequence_lengths"])
        sig.append(b["signal"]); labs.append(b["labels"]); code.append(enc)
    assert np.array_equal(np.concatenate(sig).view(np.uint32), g[f"{tag}_signal"].view(np.uint32))
    assert np.array_equal(np.concatenate(labs), g[f"{tag}_labels"])
    assert np.array_equal(np.concatenate(code), code_to_onehot(g[f"{tag}_enc_code"]))
    if ds.chunk_len == 100:  # a ConvLSTM for this context: fused path + tallies
        K = sum(ds.kmer_context_bases) + 1
        net = torch_ref.random_model("conv_lstm", 64, K, 2, seed=9)
        state = {k: v.numpy() for k, v in net.state_dict().items()}
        model = model_from_state(state, dict(chunk_context=ds.chunk_context, kmer_context_bases=ds.kmer_context_bases), device=0)
        res = validate_dataset(ds, model)
        with torch.no_grad():
            ref = net(torch.from_numpy(np.concatenate(sig)), torch.from_numpy(np.concatenate(code))).numpy()
        assert np.abs(res["logits"] - ref).max() <= 1e-4
        assert np.array_equal(res["pred_counts"], np.bincount(ref.argmax(1), minlength=2))
        assert res["confusion"].sum() == 120 and np.array_equal(res["confusion"].sum(0), res["pred_counts"])
        assert np.array_equal(ds.get_label_counts(), np.bincount(g[f"{tag}_labels"], minlength=2))


def test_staged_infer_pipeline_on_real_reads(torch_cuda, O, tmp_path):
    """The reference's `remora infer` stages under their own names (prepare_reads -> prep_nn_input ->
    batch_reads -> run_model_batched -> unbatch -> post_process_reads) on the reference's POD5 + BAM test
    data, batch size 64 so that reads straddle batches: per-read positions, logits and MM strings equal
    the reference's call_read_mods on the same reads."""
    import queue

    from remora_amd import io as rio
    from remora_amd.inference import (batch_reads, post_process_reads, prep_nn_input, prepare_reads, run_model_batched,
                                      unbatch)
    from remora_amd.model_util import load_model

    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
    g = golden("real_reads_can.npz")
    model, md = load_model(_mint_pt(tmp_path, g, O), device=0)
    read_errs = list(rio.iter_reads_from_pod5_and_bam(os.path.join(data, "can_reads.pod5"),
                                                      os.path.join(data, "can_mappings.bam")))
    names = [r.read_id for r, _ in read_errs]
    prepped = [prep_nn_input(prepare_reads([re], [md])) for re in read_errs]
    bq, cq, rq = queue.Queue(), queue.Queue(), queue.Queue()
    batch_reads(iter(prepped), bq, 64, [md])
    run_model_batched(bq, cq, {md["can_base"]: model}, [md], 64)
    unbatch(cq, rq, [md])
    done = []
    while True:
        it = rq.get()
        if it is StopIteration:
            break
        done.append(it)
    assert [d[0].read_id for d in done] == names
    total = 0
    for i, rm in enumerate(done):
        io_read, mod_calls, err = rm
        assert err is None and len(mod_calls) == 1
        _, nn_out, pos = mod_calls[0]
        assert np.array_equal(pos, g[f"r{i}_pos"])
        assert np.abs(nn_out - g[f"r{i}_nn_out"]).max() <= 1e-4
        _, mm, ml = post_process_reads(rm, [md])
        assert mm == str(g[f"r{i}_mm"])
        assert np.abs(np.asarray(list(ml), np.uint8).astype(int) - g[f"r{i}_ml"].astype(int)).max() <= 1
        total += pos.size
    assert total == 922


def test_core_dataset_writer_matches_reference_files(torch_cuda, tmp_path):
    """CoreRemoraDataset(mode="w") fed by the GPU extraction kernels writes the directory the reference wrote
    from the same two labelled reads (tests/golden/data/core_dataset): identical metadata.jsn text, identical
    signal / lengths / labels files, identical sequence / mapping rows up to each chunk's length (the reference
    leaves the padding columns uninitialised), and the result reads back through the trimming reader."""
    from remora_amd.data_chunks import CoreRemoraDataset, RemoraRead, dataset_metadata, extract_chunk_arrays
    from remora_amd.util import Motif

    g = golden("core_dataset.npz")
    ref_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data", "core_dataset")
    md = dataset_metadata(allocate_size=int(g["num_chunks"]) + 7, max_seq_len=20, mod_bases=["m"], mod_long_names=["5mC"],
                          motif_sequences=["CG"], motif_offsets=[0], chunk_context=(50, 50), kmer_context_bases=(4, 4))
    out_dir = str(tmp_path / "ds")
    ds = CoreRemoraDataset(out_dir, mode="w", metadata=md)
    for ri in range(2):
        read = RemoraRead(dacs=g[f"ds{ri}_dacs"], shift=500.0, scale=80.0, seq_to_sig_map=g[f"ds{ri}_map"],
                          int_seq=g[f"ds{ri}_int_seq"], read_id=f"ds{ri}", labels=g[f"ds{ri}_labels"])
        read.set_motif_focus_bases([Motif("CG", 0)])
        arrs, _ = extract_chunk_arrays([read], (50, 50), (4, 4), False, 0)
        ds.write_chunk_arrays(arrs)
    ds.flush()
    n = int(g["num_chunks"])
    assert int(ds.metadata["dataset_end"]) == n
    assert open(os.path.join(out_dir, "metadata.jsn")).read() == str(g["metadata_jsn"])
    for name in ("signal", "sequence_lengths", "labels"):
        nb = n * ds.arrays[name][0:1].nbytes
        mine = open(os.path.join(out_dir, f"{name}.npy"), "rb").read()[:nb]
        assert mine == open(os.path.join(ref_dir, f"{name}.npy"), "rb").read()[:nb], name
    ref = CoreRemoraDataset(ref_dir)
    lens = np.asarray(ref.arrays["sequence_lengths"][:n]).astype(int)
    for i in range(n):
        np.testing.assert_array_equal(ds.arrays["sequence"][i, : lens[i] + 8], ref.arrays["sequence"][i, : lens[i] + 8])
        np.testing.assert_array_equal(ds.arrays["sequence_to_signal_mapping"][i, : lens[i] + 1],
                                      ref.arrays["sequence_to_signal_mapping"][i, : lens[i] + 1])
    back = CoreRemoraDataset(out_dir, override_metadata={"chunk_context": (30, 25), "kmer_context_bases": (2, 3)})
    b = back.load_batch(0, n)
    np.testing.assert_array_equal(b["signal"], g["trim_signal"])
    with pytest.raises(Exception):
        ds.write_batch({"signal": np.zeros((100, 1, 100), np.float32)})  # beyond the allocation / missing arrays


@pytest.mark.parametrize("prefix,n_chunks", [("can", 847), ("mod", 1060)])
def test_real_reads_reference_anchored(torch_cuda, O, tmp_path, prefix, n_chunks):
    """Reference-anchored flavour of the same reads: Read.add_alignment(parse_ref_align=True) ->
    ref_to_signal / ref region / ref_seq -> into_remora_read(True) -> call_read_mods, and the
    `--reference-anchored` pipeline output (records rewritten to <len>M + reference sequence + MM/ML),
    against the reference's results on the same records."""
    from remora_amd import io as rio
    from remora_amd.inference import call_read_mods, infer_from_pod5_and_bam
    from remora_amd.model_util import load_model

    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
    pod5, bam = os.path.join(data, f"{prefix}_reads.pod5"), os.path.join(data, f"{prefix}_mappings.bam")
    g = _real_reads_golden(prefix)
    model, md = load_model(_mint_pt(tmp_path, g, O), device=0)
    n = 0
    for i, (read, err) in enumerate(rio.iter_reads_from_pod5_and_bam(pod5, bam)):
        assert err is None
        np.testing.assert_array_equal(read.ref_to_signal, g[f"r{i}_ra_ref_to_signal"])
        assert [read.ref_reg.ctg, read.ref_reg.strand, str(read.ref_reg.start), str(read.ref_reg.end)] == \
            [str(x) for x in g[f"r{i}_ra_region"]]
        rr = read.into_remora_read(True)
        assert rr.str_seq == str(g[f"r{i}_ra_ref_seq"]) and rr.dacs.size == int(g[f"r{i}_ra_ndacs"])
        np.testing.assert_array_equal(rr.seq_to_sig_map, g[f"r{i}_ra_map"])
        nn_out, _, pos = call_read_mods(rr, model, md)
        np.testing.assert_array_equal(pos, g[f"r{i}_ra_pos"])
        assert np.abs(nn_out - g[f"r{i}_ra_nn_out"]).max() <= 1e-4
        mm, _ = call_read_mods(read.into_remora_read(True), model, md, return_mm_ml_tags=True)
        assert mm == str(g[f"r{i}_ra_mm"])
        n += pos.size
    assert n == n_chunks
    out_bam = str(tmp_path / "ra.bam")
    stats = infer_from_pod5_and_bam(pod5, bam, model, md, out_bam, ref_anchored=True)
    assert stats.get(None) == 14
    for i, rec in enumerate(rio.iter_bam_records(out_bam)):
        ref_len = len(str(g[f"r{i}_ra_ref_seq"]))
        assert rec.cigartuples == [(0, ref_len)] and len(rec.query_sequence) == ref_len
        fwd = str(g[f"r{i}_ra_ref_seq"])
        assert rec.query_sequence == (rio.revcomp(fwd) if rec.is_reverse else fwd)
        assert dict(rec.tags)["MM"] == str(g[f"r{i}_ra_mm"])


def test_motif_scan_kernel_vs_host(torch_cuda):
    """rmr_motif_flags (through DeviceReads.motif_focus_bases) against Motif.findall on every read: IUPAC
    codes, several motifs, N bases in the reads, a focus position left of the motif after N-stripping, reads
    shorter than the motif, and no hit may straddle two reads."""
    from remora_amd.data_chunks import DeviceReads, RemoraRead
    from remora_amd.util import Motif

    rng = np.random.default_rng(12)
    reads = []
    for n in (3000, 1, 2, 777, 5, 1200, 64, 65, 4096):
        seq = rng.integers(0, 4, n).astype(np.int64)
        if n > 100:
            seq[rng.integers(0, n, n // 50)] = -1
        dw = rng.integers(1, 5, n)
        m = np.concatenate([[0], np.cumsum(dw)]).astype(np.int64)
        reads.append(RemoraRead(dacs=np.zeros(m[-1], np.int16), shift=0.0, scale=1.0, seq_to_sig_map=m, int_seq=seq))
    dr = DeviceReads(reads)
    for spec in ([("CG", 0)], [("C", 0)], [("DRACH", 2), ("CG", 1)], [("NCG", 0)], [("CHH", 0), ("CHG", 0), ("CG", 0)],
                 [("ACGTACGTACGTAC", 5)], [("N", 0)]):
        motifs = [Motif(*m) for m in spec]
        focus, foc_off = dr.motif_focus_bases(motifs)
        focus = focus.cpu().numpy()
        for i, r in enumerate(reads):
            want = set()
            for mot in motifs:
                for st in mot.findall(r.int_seq):
                    fb = int(st) + mot.focus_pos
                    if 0 <= fb < r.int_seq.size:
                        want.add(fb)
            got = focus[foc_off[i] : foc_off[i + 1]]
            assert np.array_equal(got, np.array(sorted(want), dtype=np.int64)), (spec, i)


def _svb16_encode(sig):
    """Test-side encoder of the VBZ layer below zstd: int16 samples -> deltas -> zigzag -> streamvbyte16."""
    sig = np.asarray(sig, np.int16)
    d = np.diff(np.concatenate([[0], sig.astype(np.int64)])).astype(np.int16)  # wraps like the decoder's sum
    zz = ((d.astype(np.int32) << 1) ^ (d.astype(np.int32) >> 15)).astype(np.uint32) & 0xFFFF
    two = zz > 0xFF
    keys = np.packbits(two, bitorder="little")
    data = np.empty(sig.size + int(two.sum()), np.uint8)
    offs = np.cumsum(two + 1) - (two + 1)
    data[offs] = zz & 0xFF
    data[offs[two] + 1] = zz[two] >> 8
    return keys.tobytes() + data.tobytes()


def test_vbz_decode_kernel(torch_cuda):
    """rmr_vbz_decode against the numpy decoder (itself pinned on the reference's POD5 file through the signal
    checksums of real_reads_can.npz): the rows of the real file, and synthetic rows from 1 sample to 150 k,
    all-one-byte / all-two-byte deltas, int16 wrap-around; host and device pointer flavours; corrupt input."""
    import pyarrow as pa

    from remora_amd import RemoraError
    from remora_amd import io as rio

    f = rio.Pod5File(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data", "can_reads.pod5"))
    from golden_util import pod5_reads_cpu
    from oracle import oracle as OR

    cpu = {r.read_id: r for r in pod5_reads_cpu(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data",
                                                             "can_reads.pod5"))}
    for batch in (f.read_ids, f.read_ids[:1], f.read_ids[3:9]):
        got = f.get_many(batch)
        for rid, g in zip(batch, got):
            np.testing.assert_array_equal(g.signal, cpu[rid].signal)
            np.testing.assert_array_equal(f.get(rid).signal, cpu[rid].signal)
    rng = np.random.default_rng(8)
    rows = []
    for n in (1, 7, 8, 9, 255, 2048, 2049, 4096, 40000, 102400, 150001):
        walk = np.cumsum(rng.integers(-40, 41, n)) + 500
        rows.append(walk.astype(np.int16))
    rows.append(np.full(5000, 77, np.int16))                                   # every delta fits one byte
    rows.append((rng.integers(0, 2, 5000) * 20000 - 10000).astype(np.int16))    # every delta needs two bytes
    rows.append(np.cumsum(rng.integers(-30000, 30000, 9000)).astype(np.int16))  # wraps around int16 many times
    comp = [pa.compress(_svb16_encode(r), codec="zstd", asbytes=True) for r in rows]
    for r, c in zip(rows[:6], comp[:6]):  # the encoder agrees with the host decoder
        np.testing.assert_array_equal(OR.vbz_decode_numpy(bytes(rio._zstd_decompress(c)), r.size), r)
    flat, off = rio.vbz_decode_batch(comp, [r.size for r in rows])
    for i, r in enumerate(rows):
        np.testing.assert_array_equal(flat[off[i] : off[i + 1]], r, err_msg=f"row {i}")
    dflat, _ = rio.vbz_decode_batch(comp, [r.size for r in rows], to_host=False)
    np.testing.assert_array_equal(dflat.cpu().numpy(), flat)
    bad = pa.compress(_svb16_encode(rows[4])[:-3], codec="zstd", asbytes=True)
    with pytest.raises(RemoraError, match="corrupt VBZ"):
        rio.vbz_decode_batch([comp[0], bad], [rows[0].size, rows[4].size])


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_whole_read_pipeline_vs_oracle(torch_cuda, O, seed):
    """Random model shapes (size 16/32/64, chunk length 40..260 with unequal contexts, k-mer contexts 0..6,
    2..4 classes), random motifs (IUPAC, focus anywhere), base_start_justify / offset, and random reads with N
    bases and zero-dwell bases: the batched GPU pipeline (upload, motif scan, extraction, fused inference) must
    give the oracle's positions (as a set: the batch path is ascending) and logits for every read."""
    from remora_amd import synth
    from remora_amd.data_chunks import RemoraRead
    from remora_amd.inference import call_reads_mods
    from remora_amd.model_util import model_from_state

    rng = np.random.default_rng(5000 + seed)
    size = int(rng.choice([16, 32, 64]))
    kb, ka = int(rng.integers(0, 7)), int(rng.integers(0, 7))
    cc = (int(rng.integers(20, 131)), int(rng.integers(20, 131)))
    num_out = int(rng.integers(2, 5))
    state = synth.synth_state("conv_lstm", size=size, kmer_len=kb + ka + 1, num_out=num_out, seed=seed)
    motif = [("CG", 0), ("C", 0), ("DRACH", 2), ("GATC", 1), ("CHH", 0), ("NCG", 0)][seed % 6]
    md = dict(chunk_context=cc, kmer_context_bases=(kb, ka), motifs=[motif], mod_bases=list("abc"[: num_out - 1]),
              mod_long_names=list("xyz"[: num_out - 1]), can_base=motif[0][motif[1]] if motif[0][motif[1]] in "ACGT" else "C",
              base_start_justify=bool(seed % 2), offset=int(rng.integers(0, 3)), sig_map_refiner=None)
    model = model_from_state(state, md, device=0)
    reads, raw = [], []
    for i in range(5):
        nb = int(rng.integers(3, 400))
        seq = rng.integers(0, 4, nb).astype(np.int64)
        if i % 2:
            seq[rng.integers(0, nb, max(nb // 30, 1))] = -1
        dw = rng.integers(0 if i == 2 else 1, 18, nb)
        dw[-1] = max(dw[-1], 1)
        m = np.concatenate([[0], np.cumsum(dw)]).astype(np.int64)
        d = rng.integers(300, 700, m[-1]).astype(np.int16)
        raw.append((d, m, seq))
        reads.append(RemoraRead(dacs=d, shift=500.0 + i, scale=80.0 - i, seq_to_sig_map=m, int_seq=seq, read_id=f"f{i}"))
    res = call_reads_mods(reads, model, md)
    total = 0
    for i, ((d, m, seq), (nn_out, _, pos)) in enumerate(zip(raw, res)):
        want_out, _, want_pos = O.call_read_mods(d, 500.0 + i, 80.0 - i, m, seq, state, md)
        assert sorted(np.asarray(want_pos).tolist()) == np.asarray(pos).tolist(), (seed, i)
        if len(pos):
            order = np.argsort(want_pos, kind="stable")
            assert np.abs(nn_out - want_out[order]).max() <= 1e-4, (seed, i, float(np.abs(nn_out - want_out[order]).max()))
        total += len(pos)
    assert total > 0


def test_infer_with_two_models(torch_cuda, O, tmp_path):
    """One model per canonical base (the reference's repeated --model): the 5mC CG model of the golden plus an
    all-context adenine model; every record carries both MM entries in model order, each equal to what the
    single-read API gives for that model alone (the C entry also equals the reference's)."""
    from remora_amd import io as rio
    from remora_amd import synth
    from remora_amd.inference import call_read_mods, infer_from_pod5_and_bam
    from remora_amd.model_util import load_model, model_from_state

    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
    pod5, bam = os.path.join(data, "can_reads.pod5"), os.path.join(data, "can_mappings.bam")
    g = golden("real_reads_can.npz")
    model_c, md_c = load_model(_mint_pt(tmp_path, g, O), device=0)
    md_a = dict(md_c, motifs=[("A", 0)], motif=("A", 0), can_base="A", mod_bases=["a"], mod_long_names=["6mA"],
                chunk_context=(40, 60), chunk_len=100, kmer_context_bases=(2, 3), kmer_len=6)
    model_a = model_from_state(synth.synth_state("conv_lstm", size=32, kmer_len=6, num_out=2, seed=9), md_a, device=0)
    out = str(tmp_path / "two.bam")
    stats = infer_from_pod5_and_bam(pod5, bam, [model_c, model_a], [md_c, md_a], out, reads_per_batch=6)
    assert stats.get(None) == 14
    reads = [r for r, _ in rio.iter_reads_from_pod5_and_bam(pod5, bam)]
    for i, rec in enumerate(rio.iter_bam_records(out)):
        mm = rec.get_tag("MM")
        mm_a, ml_a = call_read_mods(reads[i].into_remora_read(False), model_a, md_a, return_mm_ml_tags=True)
        assert mm == str(g[f"r{i}_mm"]) + mm_a
        assert len(rec.get_tag("ML")) == g[f"r{i}_ml"].size + len(ml_a)


def test_validate_from_remora_dataset_cli(torch_cuda, O, tmp_path):
    """`python -m remora_amd validate from_remora_dataset` on the reference-written dataset with a model whose
    contexts are smaller than the stored ones (so the trimming path runs): header + one summary line in the
    reference's format, equal to ValidationLogger.run_validation and to validate_dataset called directly."""
    import subprocess
    import sys

    from remora_amd.data_chunks import CoreRemoraDataset, RemoraDataset, validate_dataset
    from remora_amd.model_util import load_model
    from remora_amd.validate import ValidationLogger, mat_to_str

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ds_dir = os.path.join(root, "tests", "golden", "data", "core_dataset")
    g = golden("call_read_mods_cg_5mc.npz")
    pt = _mint_pt(tmp_path, g, O)
    full = str(tmp_path / "full.tsv")
    res = subprocess.run([sys.executable, "-m", "remora_amd", "validate", "from_remora_dataset", ds_dir, "--model", pt,
                          "--full-results-filename", full], cwd=root, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    model, md = load_model(pt, device=0)
    over = {"kmer_context_bases": md["kmer_context_bases"], "chunk_context": md["chunk_context"]}
    ds = CoreRemoraDataset(ds_dir, infinite_iter=False, override_metadata=dict(over))
    want = validate_dataset(ds, model)
    lines = res.stdout.strip().splitlines()
    assert lines[0] == ValidationLogger.HEADER
    f = lines[1].split("\t")
    assert f[:3] == ["val", "0", "0"] and int(f[6]) == ds.size
    assert abs(float(f[3]) - want["acc"]) < 1e-6 and f[4] == mat_to_str(want["confusion"])
    rd = RemoraDataset([CoreRemoraDataset(ds_dir, infinite_iter=False, override_metadata={"extra_arrays": {}, **over})], [1.0])
    ms = ValidationLogger(open(os.devnull, "w")).run_validation(model, md["mod_bases"], None, rd, 0.1)
    assert abs(float(f[5]) - ms.loss) < 1e-5 and f[8] == f"{ms.filt_acc:.6f}" and f[9] == mat_to_str(ms.filt_conf_mat)
    rows = open(full).read().strip().splitlines()
    assert rows[0] == ValidationLogger.FULL_HEADER and len(rows) == ds.size + 1
    labs = np.array([int(r.split("\t")[0]) for r in rows[1:]])
    calls = np.array([int(r.split("\t")[1]) for r in rows[1:]])
    assert np.array_equal(np.bincount(calls, minlength=2), want["pred_counts"])
    assert np.array_equal(np.bincount(labs, minlength=2), ds.get_label_counts())


# ---- N4: `remora dataset prepare` and RemoraDataset validation on the GPU path ---------------------------------
def _prep_args(name):
    g = golden("prepared_datasets.npz")
    which, mod_base, kw = json.loads(str(g["configs_json"]))[name]
    return g, which, mod_base, kw


def _run_prepare(name, out_dir, skip_shuffle=False, reads_per_batch=5):
    from remora_amd import io as rio
    from remora_amd.prepare_train_data import extract_chunk_dataset
    from remora_amd.refine_signal_map import SigMapRefiner
    from remora_amd.util import Motif

    g, which, mod_base, kw = _prep_args(name)
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
    refiner = SigMapRefiner()
    if kw["refine"]:
        refiner = SigMapRefiner(kmer_model_filename=os.path.join(data, "levels_4mer.txt"), do_rough_rescale=True,
                                scale_iters=0, do_fix_guage=True)
    np.random.seed(11)
    return extract_chunk_dataset(
        bam_path=os.path.join(data, f"{which}_mappings.bam"), pod5_path=os.path.join(data, f"{which}_reads.pod5"),
        out_path=out_dir, mod_base=mod_base, mod_base_control=mod_base is None, motifs=[Motif(*m) for m in kw["motifs"]],
        focus_ref_pos=None if kw["bed"] is None else rio.parse_bed(os.path.join(data, kw["bed"])),
        chunk_context=kw["chunk_context"], min_samps_per_base=kw["min_samps_per_base"],
        max_chunks_per_read=kw["max_chunks_per_read"], pa_scaling=None, sig_map_refiner=refiner,
        kmer_context_bases=kw["kmer_context_bases"], base_start_justify=kw["base_start_justify"], offset=kw["offset"],
        num_reads=None, basecall_anchor=kw["basecall_anchor"], skip_shuffle=skip_shuffle, reads_per_batch=reads_per_batch)


@pytest.mark.parametrize("name", ["can_ctrl", "mod_m", "mod_h", "can_bc_bed", "can_bc", "can_ref_bed", "can_refine",
                                  "can_default"])
def test_dataset_prepare_matches_reference(torch_cuda, tmp_path, name):
    """`remora dataset prepare` (extract_chunk_dataset) on the reference's POD5 + BAM test data against the
    dataset the reference's own extract_chunks + CoreRemoraDataset wrote under the same numpy seed: same
    metadata.jsn text, the same rows in the same (shuffled) order for every array - signal bit for bit - and the
    same pre-shuffle order; control and modified labels, reference- and basecall-anchored reads, BED-selected
    positions, down-sampling, signal-mapping refinement, two motifs / offset / base-start justification."""
    from golden_util import dataset_rows

    g, which, mod_base, kw = _prep_args(name)
    out = str(tmp_path / "ds")
    ds, errs = _run_prepare(name, out)
    assert open(os.path.join(out, "metadata.jsn")).read() == str(g[f"{name}__metadata_jsn"])
    assert errs == json.loads(str(g[f"{name}__errs_json"]))
    md, rows = dataset_rows(out)
    for k, v in rows.items():
        want = g[f"{name}__{k}"]
        assert v.shape == want.shape, k
        if k == "signal":
            assert np.array_equal(v.view(np.uint32), want.view(np.uint32)), k
        else:
            np.testing.assert_array_equal(v, want, err_msg=k)
    if f"{name}__kmer_table" in g:
        np.testing.assert_array_equal(np.load(os.path.join(out, "kmer_table.npy")), g[f"{name}__kmer_table"])
    _run_prepare(name, str(tmp_path / "ns"), skip_shuffle=True, reads_per_batch=256)
    _, pre = dataset_rows(str(tmp_path / "ns"))
    np.testing.assert_array_equal(pre["read_ids"], g[f"{name}_preshuffle__read_ids"])
    np.testing.assert_array_equal(pre["read_focus_bases"], g[f"{name}_preshuffle__read_focus_bases"])
    # the labels are the constant of the sample; each read contributes at most max_chunks_per_read chunks
    assert set(rows["labels"].tolist()) == {0 if mod_base is None else 1}
    assert max(np.unique(rows["read_ids"], return_counts=True)[1]) <= kw["max_chunks_per_read"]


@pytest.mark.parametrize("name", ["can_ctrl", "can_refine", "mod_h"])
def test_dataset_prepare_batch_ingest_and_per_read_path_write_the_same_dataset(torch_cuda, tmp_path, monkeypatch, name):
    """Reference-anchored, motif-selected `dataset prepare` runs on the batch ingest since round 5 (reads assembled on the GPU,
    focus bases and their down-sampling still per read on the host, in the reference's order): it is the path the golden test
    above exercises for these configurations - checked here by counting its calls - and the per-read path
    (RMR_PREPARE_BATCH_INGEST=0) still writes the same bytes."""
    import remora_amd.prepare_train_data as ptd
    from golden_util import dataset_rows

    calls = []
    real = ptd.extract_chunk_arrays_from_ingest
    monkeypatch.setattr(ptd, "extract_chunk_arrays_from_ingest", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    _run_prepare(name, str(tmp_path / "batch"))
    assert len(calls) >= 2, "the batch ingest did not run"
    n_batch_calls = len(calls)
    monkeypatch.setenv("RMR_PREPARE_BATCH_INGEST", "0")
    _run_prepare(name, str(tmp_path / "reads"))
    assert len(calls) == n_batch_calls
    (md_a, rows_a), (md_b, rows_b) = dataset_rows(str(tmp_path / "batch")), dataset_rows(str(tmp_path / "reads"))
    assert open(tmp_path / "batch" / "metadata.jsn").read() == open(tmp_path / "reads" / "metadata.jsn").read()
    assert sorted(rows_a) == sorted(rows_b)
    for k in rows_a:
        assert rows_a[k].shape == rows_b[k].shape and (np.array_equal(rows_a[k].view(np.uint32), rows_b[k].view(np.uint32))
                                                       if k == "signal" else np.array_equal(rows_a[k], rows_b[k])), k


def test_extract_chunks_reference_signature(torch_cuda):
    """prepare_train_data.extract_chunks with the reference's arguments and return shape: per read a list of
    Chunk objects (or an error), equal to the rows the dataset writer got."""
    from remora_amd import io as rio
    from remora_amd.prepare_train_data import extract_chunks
    from remora_amd.util import Motif

    g = golden("prepared_datasets.npz")
    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
    read_errs = list(rio.iter_reads_from_pod5_and_bam(os.path.join(data, "can_reads.pod5"), os.path.join(data, "can_mappings.bam")))
    read_errs.insert(2, (read_errs[0][0], "made-up alignment error"))
    np.random.seed(11)
    res = extract_chunks(read_errs, 0, [Motif("CG", 0)], None, None, 15, (50, 50), (4, 4), False, 0, False)
    assert len(res) == len(read_errs) and res[2] == (None, "made-up alignment error")
    ids = [c.read_id for chunks, err in res if chunks for c in chunks]
    fbs = [c.read_focus_base for chunks, err in res if chunks for c in chunks]
    keep = np.array([c.seq_len <= 20 for chunks, err in res if chunks for c in chunks])
    np.testing.assert_array_equal(np.asarray(ids)[keep], g["can_ctrl_preshuffle__read_ids"])
    np.testing.assert_array_equal(np.asarray(fbs)[keep], g["can_ctrl_preshuffle__read_focus_bases"])
    ch = res[0][0][0]
    assert ch.signal.shape == (100,) and ch.label == 0 and ch.seq_to_sig_map[0] == 0 and ch.seq_to_sig_map[-1] == 100


def test_remora_dataset_validation_matches_reference(torch_cuda, O, tmp_path):
    """Three prepared datasets (two different modified bases -> label conversion) loaded the way `remora validate
    from_remora_dataset` loads them - extra arrays dropped, the model's smaller k-mer and chunk contexts, which
    runs the trimming kernel - : every batch equals the reference's, and ValidationLogger.run_validation through
    the fused kernels gives the reference's metrics for a 3-class model and for a 2-class model with an
    un-modelled label."""
    from golden_util import materialise_dataset
    from remora_amd.data_chunks import CoreRemoraDataset, RemoraDataset
    from remora_amd.encoded_kmers import compute_encoded_kmer_batch
    from remora_amd.model_util import model_from_state
    from remora_amd.validate import ValidationLogger

    g, gp = golden("remora_dataset.npz"), golden("prepared_datasets.npz")
    paths = [materialise_dataset(gp, n, str(tmp_path / n)) for n in ("can_ctrl", "mod_m", "mod_h")]
    over = {"extra_arrays": {}, "kmer_context_bases": (2, 3), "chunk_context": (45, 40)}
    ds = RemoraDataset([CoreRemoraDataset(p, override_metadata=dict(over), infinite_iter=False, do_check_super_batches=True)
                        for p in paths], g["mix2_props"], list(g["mix2_hashes"]), batch_size=64)
    assert ds.metadata.kmer_context_bases == (2, 3) and ds.metadata.chunk_context == (45, 40)
    codes, sigs, labs, sizes = [], [], [], []
    for enc, sig, lab in ds:  # the reference's default return arrays; enc_kmers from the encode kernel
        enc = enc.numpy()
        n, K4, L = enc.shape
        e4 = enc.reshape(n, K4 // 4, 4, L)
        codes.append(np.where(e4.sum(2) > 0, e4.argmax(2), -1).astype(np.int8))
        sigs.append(sig.numpy()); labs.append(lab.numpy()); sizes.append(n)
    np.testing.assert_array_equal(sizes, g["mix3_bsizes"])
    np.testing.assert_array_equal(np.concatenate(labs), g["mix3_labels"])
    assert np.array_equal(np.concatenate(sigs).view(np.uint32), g["mix3_signal"].view(np.uint32))
    np.testing.assert_array_equal(np.concatenate(codes), g["mix3_enc_code"])
    ds.load_all_batches()
    np.testing.assert_array_equal(ds.get_label_counts(), g["mix3_label_counts_loaded"])
    ds = RemoraDataset([CoreRemoraDataset(p, override_metadata=dict(over), infinite_iter=False) for p in paths],
                       g["mix2_props"], list(g["mix2_hashes"]), batch_size=64)
    val = ValidationLogger(open(os.devnull, "w"))
    for tag, prefix, mods in (("hm", "mix3_w__", ["h", "m"]), ("m_only", "mix3b_w__", ["m"])):
        state = O.state_from_npz(g, prefix)
        model = model_from_state(state, dict(chunk_context=(45, 40), kmer_context_bases=(2, 3)), device=0)
        ms = val.run_validation(model, mods, None, ds, 0.1)
        loss, acc, ncalls, ff, facc, thr = g[f"mix3_{tag}_metrics"]
        assert ms.num_calls == int(ncalls)
        np.testing.assert_array_equal(ms.conf_mat, g[f"mix3_{tag}_conf"])
        np.testing.assert_array_equal(ms.filt_conf_mat, g[f"mix3_{tag}_filt_conf"])
        assert ms.acc == acc and ms.filt_frac == ff and ms.filt_acc == facc
        assert abs(ms.loss - loss) <= 1e-5 * max(1.0, abs(loss)) and abs(ms.filt_thresh - thr) <= 1e-5


def test_dataset_cli_prepare_then_validate(torch_cuda, O, tmp_path):
    """`python -m remora_amd dataset prepare` twice (control + modified sample), a two-dataset config (the file format
    of the reference's `chunks` fixture, tests/conftest.py), then `validate from_remora_dataset` on the config: the
    reference's flow for a two-sample dataset end to end on the GPU path."""
    import subprocess
    import sys

    from remora_amd.data_chunks import CoreRemoraDataset
    from remora_amd.validate import ValidationLogger

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = os.path.join(root, "tests", "golden", "data")
    run = lambda *a: subprocess.run([sys.executable, "-m", "remora_amd", *a], cwd=root, capture_output=True, text=True, timeout=600)
    can, mod, cfg = str(tmp_path / "can_chunks"), str(tmp_path / "mod_chunks"), str(tmp_path / "chunks.cfg")
    r = run("dataset", "prepare", os.path.join(data, "can_reads.pod5"), os.path.join(data, "can_mappings.bam"), "--output-path", can,
            "--mod-base-control", "--motif", "CG", "0", "--chunk-context", "50", "50")
    assert r.returncode == 0, r.stderr[-2000:]
    r = run("dataset", "prepare", os.path.join(data, "mod_reads.pod5"), os.path.join(data, "mod_mappings.bam"), "--output-path", mod,
            "--mod-base", "m", "5mC", "--motif", "CG", "0", "--chunk-context", "50", "50")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Label distribution: control:0; 5mC:" in r.stdout
    r = run("dataset", "prepare", os.path.join(data, "mod_reads.pod5"), os.path.join(data, "mod_mappings.bam"), "--output-path", mod,
            "--mod-base", "m", "5mC", "--motif", "CG", "0")
    assert r.returncode == 1 and "Refusing to overwrite" in r.stderr
    n_can, n_mod = CoreRemoraDataset(can).size, CoreRemoraDataset(mod).size  # 14 reads x <= 15 random focus bases each
    assert 150 < n_can <= 210 and 150 < n_mod <= 210
    json.dump([[can, n_can / (n_can + n_mod)], [mod, n_mod / (n_can + n_mod)]], open(cfg, "w"))
    pt = _mint_pt(tmp_path, golden("call_read_mods_cg_5mc.npz"), O)
    r = run("validate", "from_remora_dataset", cfg, "--model", pt, "--batch-size", "100")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[0] == ValidationLogger.HEADER
    f = lines[1].split("\t")
    conf_mat = np.array(json.loads(f[4]))
    # batches of 100 split 49 / 51 until the smaller dataset runs out: 4 full batches + the 9 / 6 remainder
    assert int(f[6]) == conf_mat.sum() and conf_mat.shape == (2, 2) and int(f[6]) >= 300
    assert conf_mat[0].sum() > 100 and conf_mat[1].sum() > 100  # both samples' labels are present


def test_reference_etl_tests_known_sizes(torch_cuda, tmp_path):
    """The reference's own ETL tests, same command lines and the same expected numbers (tests/conftest.py:251-403,
    tests/test_main.py:23-136): `dataset prepare` on the canonical sample gives 205 chunks all labelled 0, on the
    modified sample 210 chunks all labelled 1 (also with ChEBI codes as short names); the two-dataset config has
    label counts [205, 210]; a config over the four datasets has four labels with 205 / 210 / 210 / 210; the expanded
    config (`RemoraDataset.get_config`, what the reference's `dataset inspect --out-path` writes) carries the hashes."""
    import subprocess
    import sys

    from remora_amd.data_chunks import CoreRemoraDataset, RemoraDataset

    EXPECTED_CAN_SIZE, EXPECTED_MOD_SIZE = 205, 210
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = os.path.join(root, "tests", "golden", "data")

    def remora(*a):
        r = subprocess.run([sys.executable, "-m", "remora_amd", *a], cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout

    def prepare(name, sample, *label_args):
        out = str(tmp_path / name)
        remora("dataset", "prepare", os.path.join(data, f"{sample}_reads.pod5"), os.path.join(data, f"{sample}_mappings.bam"),
               "--output-path", out, *label_args, "--motif", "CG", "0")
        return out

    can_chunks = prepare("can_chunks", "can", "--mod-base-control")
    mod_chunks = prepare("mod_chunks", "mod", "--mod-base", "m", "5mC")
    mod_chebi_chunks = prepare("mod_chebi_chunks", "mod", "--mod-base", "27551", "5-methylcytosine")
    mod_chebi2_chunks = prepare("mod_chebi2_chunks", "mod", "--mod-base", "76792", "5-hydroxymethylcytosine")
    ds = CoreRemoraDataset(can_chunks, batch_size=10)  # test_prep_can
    assert ds.size == EXPECTED_CAN_SIZE and ds.get_label_counts()[0] == EXPECTED_CAN_SIZE
    assert ds.metadata.chunk_context == (200, 200) and ds.metadata.max_seq_len == 80
    for path in (mod_chunks, mod_chebi_chunks):  # test_prep_mod, test_prep_mod_chebi
        ds = CoreRemoraDataset(path, batch_size=10)
        assert ds.size == EXPECTED_MOD_SIZE and ds.get_label_counts()[1] == EXPECTED_MOD_SIZE
    chunks = str(tmp_path / "chunks.cfg")  # the `chunks` fixture
    json.dump([[can_chunks, 0.5], [mod_chunks, 0.5]], open(chunks, "w"))
    dataset = RemoraDataset.from_config(chunks, batch_size=10)  # test_remora_dataset
    counts = dataset.get_label_counts()
    assert counts.size == 2 and dataset.size == EXPECTED_CAN_SIZE + EXPECTED_MOD_SIZE
    assert counts[0] == EXPECTED_CAN_SIZE and counts[1] == EXPECTED_MOD_SIZE
    chebi_chunks = str(tmp_path / "chebi.cfg")  # the `chebi_chunks` fixture
    four = [can_chunks, mod_chunks, mod_chebi_chunks, mod_chebi2_chunks]  # weights = sizes, as the reference's make_config defaults
    sizes = np.array([CoreRemoraDataset(p).size for p in four], float)
    json.dump([[p, w] for p, w in zip(four, (sizes / sizes.sum()).tolist())], open(chebi_chunks, "w"))
    dataset = RemoraDataset.from_config(chebi_chunks, batch_size=10)  # test_remora_dataset_chebi
    counts = dataset.get_label_counts()
    assert counts.size == 4 and dataset.size == EXPECTED_CAN_SIZE + 3 * EXPECTED_MOD_SIZE
    assert list(counts) == [EXPECTED_CAN_SIZE] + [EXPECTED_MOD_SIZE] * 3
    assert dataset.metadata.mod_bases == ["27551", "76792", "m"]
    batch = next(iter(dataset))  # every dataset contributes to every batch of 10
    assert batch[0].shape == (10, 36, 400) and batch[1].shape == (10, 1, 400) and batch[2].shape == (10,)
    for cfg in (chunks, chebi_chunks):  # the expanded config: one [path, weight, sha256] entry per core dataset
        conf = RemoraDataset.from_config(cfg, batch_size=10).get_config()
        assert len(conf) == (2 if cfg == chunks else 4) and all(len(c[2]) == 64 for c in conf)


def test_host_buffer_path_pipelined_upload_matches_device_path(torch_cuda):
    """rmr_infer_chunks with HOST buffers larger than the upload sub-batch (pinned double-buffered copies on the
    aux stream under the kernels of the previous sub-batch, ragged last sub-batch): logits bit-identical to the
    device-resident call, label counts equal; also with a tiny sub-batch so that many slots are recycled."""
    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    torch = torch_cuda
    n = 300_001
    data = synth.synth_chunks(n, 100, 20, (4, 4), seed=77)
    model = model_from_state(synth.synth_state("conv_lstm", 64, 9, 2, seed=3), dict(chunk_context=(50, 50), kmer_context_bases=(4, 4)), device=0)
    host = [data[k] for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
    dev = [torch.from_numpy(a).cuda() for a in host]
    dc = torch.zeros(2, dtype=torch.int64, device="cuda")
    want = model.infer_chunks(*dev, (4, 4), label_counts=dc).cpu().numpy()
    for sub in (None, "4096"):
        if sub is not None:
            os.environ["RMR_HOST_SUBBATCH"] = sub
        try:
            hc = np.zeros(2, np.int64)
            got = model.infer_chunks(*host, (4, 4), label_counts=hc)
        finally:
            os.environ.pop("RMR_HOST_SUBBATCH", None)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert np.array_equal(hc, dc.cpu().numpy()) and hc.sum() == n


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "f16"])
def test_streamed_read_batches_equal_batched_calls(torch_cuda, dtype):
    """iter_call_reads_mods (staging of the next batch in a worker thread on its own stream) returns, batch by
    batch, exactly what call_reads_mods returns - without a refiner and with one (rough re-scale + banded DP) - with the
    fp32 model and with the 16-bit models (whose streamed rate bench.py reports)."""
    sys_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    import sys

    sys.path.insert(0, sys_path)
    import bench_refine
    from remora_amd import synth
    from remora_amd.data_chunks import RemoraRead
    from remora_amd.inference import call_reads_mods, iter_call_reads_mods
    from remora_amd.model_util import model_from_state
    from remora_amd.refine_signal_map import SigMapRefiner

    md = dict(chunk_context=(50, 50), kmer_context_bases=(4, 4), motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"],
              can_base="C", base_start_justify=False, offset=0, sig_map_refiner=None)
    model = model_from_state(synth.synth_state(seed=4), md, device=0, dtype=dtype)
    table, center, base = bench_refine.synth_reads(40, 1500, seed=9)
    refiner = SigMapRefiner(_levels_array=table, center_idx=center, do_rough_rescale=True, scale_iters=0)

    def fresh():
        return [RemoraRead(dacs=b[0], shift=400.0, scale=60.0, seq_to_sig_map=b[1].copy(), int_seq=b[2], read_id=f"r{i}")
                for i, b in enumerate(base)]

    for mdx in (md, dict(md, sig_map_refiner=refiner)):
        want_reads = fresh()
        sizes = [7, 1, 0, 20, 12]
        cuts = np.cumsum([0] + sizes)
        want = [call_reads_mods(want_reads[a:b], model, mdx) for a, b in zip(cuts[:-1], cuts[1:])]
        got_reads = fresh()
        got = list(iter_call_reads_mods([got_reads[a:b] for a, b in zip(cuts[:-1], cuts[1:])], model, mdx))
        assert len(got) == len(want)
        for (reads, res), exp, (a, b) in zip(got, want, zip(cuts[:-1], cuts[1:])):
            assert [r.read_id for r in reads] == [r.read_id for r in got_reads[a:b]] and len(res) == len(exp)
            for (o1, l1, p1), (o2, l2, p2) in zip(res, exp):
                assert np.array_equal(p1, p2) and np.array_equal(np.asarray(o1).view(np.uint32), np.asarray(o2).view(np.uint32))
        for r1, r2 in zip(got_reads, want_reads):
            assert np.array_equal(r1.seq_to_sig_map, r2.seq_to_sig_map) and (r1.shift, r1.scale) == (r2.shift, r2.scale)
    assert list(iter_call_reads_mods([], model, md)) == []


def test_infer_pipeline_repeated_calls_and_large_move_batches(torch_cuda, O, tmp_path):
    """infer_from_pod5_and_bam several times in one process (a fresh ingest thread per call) gives the same output
    file every time; and rmr_parse_moves_batch with far more tables than one ingest batch (staging sized per read)."""
    from remora_amd.inference import infer_from_pod5_and_bam
    from remora_amd.io import parse_move_tags
    from remora_amd.model_util import load_model

    data = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")
    model, md = load_model(_mint_pt(tmp_path, golden("real_reads_can.npz"), O), device=0)
    outs = []
    for k in range(4):
        out = str(tmp_path / f"o{k}.bam")
        stats = infer_from_pod5_and_bam(os.path.join(data, "can_reads.pod5"), os.path.join(data, "can_mappings.bam"), model, md,
                                        out, reads_per_batch=5)
        assert stats[None] == 14
        outs.append(open(out, "rb").read())
    assert all(o == outs[0] for o in outs[1:])
    rng = np.random.default_rng(2)
    tags, sls, qls = [], [], []
    for _ in range(3000):
        n = int(rng.integers(1, 40))
        mv = (rng.random(n) < 0.5).astype(np.int8)
        mv[0] = 1
        tags.append(np.concatenate([[5], mv]).astype(np.int8)); sls.append(5 * n + 2); qls.append(int(mv.sum()))
    for t, sl, ql, r in zip(tags, sls, qls, parse_move_tags(tags, sls, qls)):
        assert np.array_equal(r[0], O.parse_move_tag(t, sl, seq_len=ql)[0])


def test_library_can_be_touched_before_torch(torch_cuda):
    """A fresh process that reads a BAM file through the native reader (which loads libremora_hip.so) before
    anything imported torch can still create an engine afterwards: the loader brings PyTorch's HIP runtime in
    first (two HIP runtimes in the wrong order end in "no ROCm-capable device")."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from remora_amd import io as rio\n"
            "n = sum(1 for _ in rio.iter_bam_records(%r))\n"
            "assert 'torch' in sys.modules\n"
            "from remora_amd.engine import get_engine\n"
            "get_engine().synchronize(); print('ok', n)\n") % (root, os.path.join(root, "tests", "golden", "data", "can_mappings.bam"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok 14", r.stderr[-1500:]


def test_device_reads_mapping_crosses_pcie_as_int32_and_arrives_as_int64(torch_cuda, monkeypatch):
    """DeviceReads ships the mapping as int32 and widens it on the device (rmr_pack_reads_narrow + a cast behind the copy): the
    resident arrays equal the reads' own, synchronous and asynchronous upload; a batch with a mapping value beyond int32 is
    gathered again as int64 (also from the asynchronous path, whose staging slot it hands back); RMR_READS_NARROW_MAPS=0
    ships int64 from the start."""
    from remora_amd.data_chunks import DeviceReads, RemoraRead

    rng = np.random.default_rng(3)

    def batch(big=None):
        reads = []
        for i, n in enumerate((300, 1, 2, 77, 1200)):
            m = np.concatenate([[0], np.cumsum(rng.integers(1, 9, n))]).astype(np.int64)
            if big is not None and i == 3:
                m[-1] = big
            reads.append(RemoraRead(dacs=rng.integers(-900, 900, 20 + int(min(max(m[-1], 0), 20000))).astype(np.int16), shift=1.5, scale=2.0,
                                    seq_to_sig_map=m, int_seq=rng.integers(0, 4, n).astype(np.int64)))
        return reads

    def check(dr, reads):
        dr.wait_ready()
        torch_cuda.cuda.synchronize()
        assert dr.s2s.dtype == torch_cuda.int64
        assert np.array_equal(dr.s2s.cpu().numpy(), np.concatenate([r.seq_to_sig_map for r in reads]))
        assert np.array_equal(dr.dacs.cpu().numpy(), np.concatenate([r.dacs for r in reads]))
        assert np.array_equal(dr.iseq.cpu().numpy(), np.concatenate([r.int_seq for r in reads]).astype(np.int8))
        assert np.array_equal(dr.d_seq_off.cpu().numpy(), dr.seq_off)

    for big in (None, (1 << 31) + 5, -(1 << 31) - 1):
        reads = batch(big)
        check(DeviceReads(reads), reads)
        for _ in range(3):  # the two staging slots take turns; a retry must not leave one of them lost
            check(DeviceReads(reads, async_upload=True), reads)
    monkeypatch.setenv("RMR_READS_NARROW_MAPS", "0")
    reads = batch()
    check(DeviceReads(reads), reads)
