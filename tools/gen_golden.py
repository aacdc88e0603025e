#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING the reference (nanoporetech/remora
v3.2.0, read-only at /root/reference) in the build container.

This script is test infrastructure. It is the only place the reference is ever
imported; nothing here travels to the GPU box except the small `.npz` files it
writes under tests/golden/ (data only: inputs and the reference's outputs).

Recipe: the three Cython modules are built out-of-tree in
/tmp, absent third-party deps (pysam, pod5, polars, ...) are replaced by permissive
stub modules, then `remora.*` is imported from the scratch copy and driven.

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import types

import numpy as np

REF = "/root/reference"
SCRATCH = "/tmp/remora_ref_build"


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit(f"reference not found at {REF}; run in the build container")
    so_ok = os.path.isdir(SCRATCH) and any(
        f.startswith("encoded_kmers") and f.endswith(".so")
        for f in os.listdir(os.path.join(SCRATCH, "src", "remora"))
    )
    if not so_ok:
        shutil.rmtree(SCRATCH, ignore_errors=True)
        os.makedirs(SCRATCH)
        for name in ("src", "setup.py", "setup.cfg", "README.rst"):
            src = os.path.join(REF, name)
            dst = os.path.join(SCRATCH, name)
            if os.path.isdir(src):
                shutil.copytree(src, dst)
            else:
                shutil.copy(src, dst)
        subprocess.check_call(["chmod", "-R", "u+w", SCRATCH])
        subprocess.check_call(
            [sys.executable, "setup.py", "build_ext", "--inplace"],
            cwd=SCRATCH,
            stdout=subprocess.DEVNULL,
        )
    sys.path.insert(0, os.path.join(SCRATCH, "src"))

    class _Any:
        def __init__(s, *a, **k):
            pass

        def __call__(s, *a, **k):
            return _Any()

        def __getattr__(s, n):
            return _Any()

        def __add__(s, o):
            return s

    class _Stub(types.ModuleType):
        def __getattr__(s, n):
            if n.startswith("__"):
                raise AttributeError(n)
            return _Any()

    for m in ("toml", "polars", "pysam", "pod5", "plotnine", "parasail", "thop"):
        sys.modules.setdefault(m, _Stub(m))
    import remora  # noqa
    from remora import (  # noqa
        data_chunks,
        data_chunks_core,
        encoded_kmers,
        inference,
        io,
        model_util,
        util,
        validate,
    )

    return types.SimpleNamespace(
        remora=remora,
        data_chunks=data_chunks,
        data_chunks_core=data_chunks_core,
        encoded_kmers=encoded_kmers,
        inference=inference,
        io=io,
        model_util=model_util,
        util=util,
        validate=validate,
    )


# --------------------------------------------------------------------------------------
# helpers shared by several fixtures
# --------------------------------------------------------------------------------------


def synth_read(rng, nbases, dwell_lo=5, dwell_hi=15, with_n=False, zero_dwell=False):
    """SURVEY §8(d) synthetic read: uniform bases, uniform dwell, uniform dacs."""
    int_seq = rng.integers(0, 4, nbases).astype(np.int64)
    if with_n:
        int_seq[rng.choice(nbases, max(1, nbases // 25), replace=False)] = -1
    dwells = rng.integers(dwell_lo, dwell_hi + 1, nbases)
    if zero_dwell:
        dwells[rng.choice(nbases, max(1, nbases // 10), replace=False)] = 0
        dwells[0] = max(dwells[0], 1)
    seq_to_sig = np.concatenate([[0], np.cumsum(dwells)]).astype(np.int64)
    dacs = rng.integers(300, 701, seq_to_sig[-1]).astype(np.int16)
    return dacs, seq_to_sig, int_seq


def randomise_bn(net, gen):
    """Non-trivial BatchNorm running stats so that folding errors are visible."""
    import torch

    for name, mod in net.named_modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            n = mod.num_features
            mod.running_mean.copy_(torch.randn(n, generator=gen))
            mod.running_var.copy_(torch.rand(n, generator=gen) * 1.5 + 0.5)
            mod.weight.data.copy_(1.0 + 0.2 * torch.randn(n, generator=gen))
            mod.bias.data.copy_(0.2 * torch.randn(n, generator=gen))


def make_net(R, arch, size, kmer_len, num_out, seed):
    import torch

    torch.manual_seed(seed)
    net = R.model_util._load_python_model(
        f"{REF}/models/{arch}.py", size=size, kmer_len=kmer_len, num_out=num_out
    )
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        randomise_bn(net, gen)
        # default init gives logits in a ~0.05-wide band; widen the dynamic range so the
        # 1e-4 logit tolerance is a discriminating test (still the reference's forward)
        for pname, p in net.named_parameters():
            if pname.startswith("lstm") and "weight" in pname:
                p.mul_(2.5)
            if pname.startswith("fc."):
                p.mul_(12.0 if hasattr(net, "lstm1") else 1.5)  # keep logits within a few units
            if pname.startswith("lstm") and pname.endswith("bias_ih_l0"):
                n4 = p.shape[0] // 4
                p[n4 : 2 * n4] += 2.0  # forget-gate bias: longer memory, as in trained LSTMs
            if "conv" in pname and pname.endswith("weight"):
                p.mul_(2.45)  # He-scale: keeps the logits sensitive to the inputs
    net.eval()
    return net


def state_to_np(net, prefix="w__"):
    return {
        prefix + k.replace(".", "__"): v.detach().cpu().numpy()
        for k, v in net.state_dict().items()
    }


def synth_chunks(rng, n, chunk_len, max_seq_len, kb, ka, minus_one_frac=0.0, zero_dwell=False):
    """Random chunk arrays in CoreRemoraDataset layout (data_chunks.py:786-816)."""
    seq_w = max_seq_len + kb + ka
    seqs = np.full((n, seq_w), -1, np.int8)
    maps = np.zeros((n, max_seq_len + 1), np.int16)
    lens = np.zeros(n, np.int16)
    for c in range(n):
        sl = int(rng.integers(max(1, max_seq_len // 4), max_seq_len + 1))
        if zero_dwell and c % 3 == 0:
            cuts = np.sort(rng.integers(0, chunk_len + 1, sl - 1))
        else:
            sl = min(sl, chunk_len)
            cuts = np.sort(rng.choice(np.arange(1, chunk_len), sl - 1, replace=False))
        maps[c, : sl + 1] = np.concatenate([[0], cuts, [chunk_len]])
        # padded columns are garbage in the reference (np.empty): emulate
        maps[c, sl + 1 :] = rng.integers(-5, chunk_len + 5, max_seq_len - sl)
        seqs[c, : sl + kb + ka] = rng.integers(0, 4, sl + kb + ka)
        seqs[c, sl + kb + ka :] = rng.integers(-1, 4, seq_w - sl - kb - ka)
        if minus_one_frac > 0:
            m = rng.random(sl + kb + ka) < minus_one_frac
            seqs[c, : sl + kb + ka][m] = -1
        lens[c] = sl
    return seqs, maps, lens


# --------------------------------------------------------------------------------------
# fixtures
# --------------------------------------------------------------------------------------


def gen_parse_move_tag(R, out):
    """io.parse_move_tag (src/remora/io.py:394-407)."""
    rng = np.random.default_rng(11)
    cases = {}
    idx = 0

    def add(mv_tag, sig_len, seq_len, reverse, check=True):
        nonlocal idx
        try:
            q2s, mv, stride = R.io.parse_move_tag(
                list(mv_tag), sig_len, seq_len=seq_len, check=check, reverse_signal=reverse
            )
            err = ""
        except R.remora.RemoraError as e:
            q2s, err = np.zeros(0, np.int64), str(e)
        cases[f"c{idx}_mv"] = np.asarray(mv_tag, np.int8)
        cases[f"c{idx}_args"] = np.asarray(
            [sig_len, -1 if seq_len is None else seq_len, int(reverse), int(check)], np.int64
        )
        cases[f"c{idx}_q2s"] = np.asarray(q2s, np.int64)
        cases[f"c{idx}_err"] = np.asarray(err)
        idx += 1

    add([5, 1, 0, 0, 1, 1, 0, 1, 0], 40, 4, False)
    add([5, 1, 0, 0, 1, 1, 0, 1, 0], 40, 4, True)
    add([5, 1, 0, 0, 1, 1, 0, 1, 0], 43, 4, False)  # sig_len not multiple of stride
    add([5, 1, 0, 0, 1, 1, 0, 1, 0], 40, 5, False)  # discordant with basecalls
    add([5, 1, 0, 0, 1, 1, 0, 1], 40, 4, False)  # discordant with signal
    add([5, 1, 0, 0, 1, 1, 0, 1], 40, 4, False, check=False)
    add([6, 1], 6, 1, False)
    for stride in (5, 6):
        nmv = int(rng.integers(200, 2000))
        mv = (rng.random(nmv) < 0.45).astype(np.int8)
        mv[0] = 1
        sig_len = nmv * stride + int(rng.integers(0, stride))
        add([stride, *mv], sig_len, int(mv.sum()), False)
        add([stride, *mv], sig_len, int(mv.sum()), True)
        add([stride, *mv], sig_len, None, False)
    cases["num_cases"] = np.asarray(idx)
    np.savez_compressed(os.path.join(out, "parse_move_tag.npz"), **cases)
    print("parse_move_tag:", idx, "cases")


def gen_seq_motif(R, out):
    """util.seq_to_int (util.py:131-142), Motif.findall (:281-297),
    find_focus_bases_in_int_sequence (:413-426; python-set iteration order)."""
    rng = np.random.default_rng(12)
    d = {}
    seqs = [
        "ACGTACGTNNACGCGCGTTTCGA",
        "CGCGCGCG",
        "A",
        "".join(rng.choice(list("ACGT"), 3000)),
        "".join(rng.choice(list("ACGTN"), 500, p=[0.24, 0.24, 0.24, 0.24, 0.04])),
    ]
    motif_sets = [
        [("CG", 0)],
        [("C", 0)],
        [("CG", 0), ("GC", 1)],
        [("CHH", 0), ("CHG", 0), ("CG", 0)],
        [("NCGN", 1)],
        [("DRACH", 2)],
        [("A", 0), ("C", 0), ("G", 0), ("T", 0)],
    ]
    d["num_seqs"] = np.asarray(len(seqs))
    d["num_motif_sets"] = np.asarray(len(motif_sets))
    for si, s in enumerate(seqs):
        int_seq = R.util.seq_to_int(s)
        d[f"s{si}_str"] = np.asarray(s)
        d[f"s{si}_int"] = int_seq.astype(np.int64)
        for mi, ms in enumerate(motif_sets):
            motifs = [R.util.Motif(*m) for m in ms]
            if any(len(m.raw_motif) > len(s) for m in motifs):
                fb = np.zeros(0, np.int64)
                skipped = 1
            else:
                fb = R.util.find_focus_bases_in_int_sequence(int_seq, motifs)
                skipped = 0
            d[f"s{si}_m{mi}_focus"] = np.asarray(fb, np.int64)
            d[f"s{si}_m{mi}_skipped"] = np.asarray(skipped)
    for mi, ms in enumerate(motif_sets):
        d[f"m{mi}_seqs"] = np.asarray([m[0] for m in ms])
        d[f"m{mi}_offs"] = np.asarray([m[1] for m in ms], np.int64)
        # Motif normalisation (N-clipping) results
        mots = [R.util.Motif(*m) for m in ms]
        d[f"m{mi}_norm_seqs"] = np.asarray([m.raw_motif for m in mots])
        d[f"m{mi}_norm_offs"] = np.asarray([m.focus_pos for m in mots], np.int64)
    np.savez_compressed(os.path.join(out, "seq_motif.npz"), **d)
    print("seq_motif: done")


def _chunks_to_arrays(chunks):
    """Ragged Chunk list -> padded arrays (+ lengths)."""
    n = len(chunks)
    L = chunks[0].signal.size if n else 0
    max_sw = max((c.seq_w_context.size for c in chunks), default=0)
    max_mw = max((c.seq_to_sig_map.size for c in chunks), default=0)
    sig = np.zeros((n, L), np.float32)
    seq = np.full((n, max_sw), -9, np.int64)
    smap = np.full((n, max_mw), -9999, np.int64)
    seq_len = np.zeros(n, np.int64)
    misc = np.zeros((n, 4), np.int64)
    for i, c in enumerate(chunks):
        sig[i] = c.signal
        seq[i, : c.seq_w_context.size] = c.seq_w_context
        smap[i, : c.seq_to_sig_map.size] = c.seq_to_sig_map
        seq_len[i] = c.seq_len
        misc[i] = (c.chunk_sig_focus_idx, c.chunk_focus_base, c.read_focus_base, c.label)
    return dict(signal=sig, seq_w_context=seq, seq_to_sig_map=smap, seq_len=seq_len, misc=misc)


def gen_extract_chunks(R, out):
    """RemoraRead.sig (data_chunks.py:191-197), iter_chunks (:425-466),
    extract_chunk (:331-423) incl. both padding branches and -1 sequence fill."""
    rng = np.random.default_rng(13)
    d = {}
    reads = [
        ("long", synth_read(rng, 400), 500.0, 80.0),
        ("short", synth_read(rng, 7), 511.25, 77.5),  # shorter than one chunk
        ("with_n", synth_read(rng, 150, with_n=True), 480.0, 91.3),
        ("zero_dwell", synth_read(rng, 120, dwell_lo=1, dwell_hi=6, zero_dwell=True), 500.0, 80.0),
        ("fast", synth_read(rng, 300, dwell_lo=1, dwell_hi=3), 500.0, 80.0),
    ]
    configs = [
        ((50, 50), (4, 4), False, 0),
        ((100, 100), (4, 4), False, 0),
        ((50, 50), (2, 3), True, 0),
        ((30, 25), (4, 4), False, 1),
        ((50, 50), (4, 4), False, -2),
        ((200, 200), (1, 10), True, 0),
    ]
    d["read_names"] = np.asarray([r[0] for r in reads])
    d["configs"] = np.asarray([[*c[0], *c[1], int(c[2]), c[3]] for c in configs], np.int64)
    for rname, (dacs, s2s, int_seq), shift, scale in reads:
        read = R.data_chunks.RemoraRead(
            dacs=dacs, shift=shift, scale=scale, seq_to_sig_map=s2s, int_seq=int_seq, read_id=rname
        )
        read.check()
        d[f"{rname}_dacs"] = dacs
        d[f"{rname}_map"] = s2s
        d[f"{rname}_int_seq"] = int_seq
        d[f"{rname}_shift_scale"] = np.asarray([shift, scale], np.float64)
        d[f"{rname}_sig"] = read.sig
        for motif in (("CG", 0), ("C", 0)):
            mname = motif[0]
            read.set_motif_focus_bases([R.util.Motif(*motif)])
            d[f"{rname}_{mname}_focus"] = np.asarray(read.focus_bases, np.int64)
            for ci, (cc, kcb, bsj, off) in enumerate(configs):
                chunks = list(read.iter_chunks(cc, kcb, bsj, off))
                assert len(chunks) == read.focus_bases.size
                arrs = _chunks_to_arrays(chunks)
                for k, v in arrs.items():
                    d[f"{rname}_{mname}_c{ci}_{k}"] = v
    np.savez_compressed(os.path.join(out, "extract_chunks.npz"), **d)
    print("extract_chunks: done", sum(v.nbytes for v in d.values()) // 1024, "KiB raw")


def gen_extract_chunk_padding(R, out):
    """RemoraRead.extract_chunk(..., signal_padding=True) (data_chunks.py:331-368): the zero padding replaced by the mirrored
    signal next to the read's ends.  No caller of the reference passes it; pinned for completeness of the method.  A read
    shorter than the padding it would have to mirror makes numpy refuse the assignment: recorded as the error text."""
    rng = np.random.default_rng(113)
    d = {}
    reads = [("long", synth_read(rng, 300), 500.0, 80.0), ("short", synth_read(rng, 7), 511.25, 77.5),
             ("mid", synth_read(rng, 30), 480.0, 91.3)]
    d["read_names"] = np.asarray([r[0] for r in reads])
    cases = []
    for rname, (dacs, s2s, int_seq), shift, scale in reads:
        read = R.data_chunks.RemoraRead(dacs=dacs, shift=shift, scale=scale, seq_to_sig_map=s2s, int_seq=int_seq, read_id=rname)
        d[f"{rname}_dacs"], d[f"{rname}_map"], d[f"{rname}_int_seq"] = dacs, s2s, int_seq
        d[f"{rname}_shift_scale"] = np.asarray([shift, scale], np.float64)
        n = dacs.size
        for f, cc in ((3, (50, 50)), (n - 2, (50, 50)), (n // 2, (50, 50)), (10, (30, 25)), (n - 1, (100, 100)), (0, (20, 20)),
                      (n // 2, (200, 200)), (n, (10, 40))):
            key = f"{rname}_f{f}_c{cc[0]}_{cc[1]}"
            try:
                ch = read.extract_chunk(f, cc, (4, 4), label=0, read_focus_base=5, signal_padding=True)
                d[key + "_signal"] = ch.signal.astype(np.float32)
                d[key + "_map"] = ch.seq_to_sig_map.astype(np.int32)
                d[key + "_seq"] = np.asarray(ch.seq_w_context, np.int8)
                err = ""
            except ValueError as e:
                err = str(e)
            d[key + "_err"] = np.asarray(err)
            cases.append(key)
    d["cases"] = np.asarray(cases)
    np.savez_compressed(os.path.join(out, "extract_chunk_padding.npz"), **d)
    print("extract_chunk_padding:", len(cases), "cases,", sum(1 for c in cases if str(d[c + "_err"])), "refused by numpy")


def gen_encode_kmers(R, out):
    """encoded_kmers.compute_encoded_kmer_batch (src/remora/encoded_kmers.pyx:13-45)."""
    rng = np.random.default_rng(14)
    d = {}
    cases = [
        # n, chunk_len, max_seq_len, kb, ka, minus_one_frac, zero_dwell
        (40, 100, 20, 4, 4, 0.0, False),
        (33, 100, 20, 4, 4, 0.1, True),
        (17, 200, 40, 4, 4, 0.05, True),
        (21, 100, 20, 2, 3, 0.05, False),
        (9, 55, 30, 1, 10, 0.0, True),
        (5, 100, 100, 4, 4, 0.02, True),  # one base per sample possible
        (1, 100, 20, 4, 4, 0.0, False),
        (3, 400, 80, 0, 0, 0.0, False),
    ]
    d["num_cases"] = np.asarray(len(cases))
    for i, (n, L, msl, kb, ka, mof, zd) in enumerate(cases):
        seqs, maps, lens = synth_chunks(rng, n, L, msl, kb, ka, mof, zd)
        enc = R.encoded_kmers.compute_encoded_kmer_batch(kb, ka, seqs, maps, lens)
        assert enc.shape == (n, 4 * (kb + ka + 1), L), enc.shape
        d[f"c{i}_args"] = np.asarray([kb, ka, L], np.int64)
        d[f"c{i}_seqs"] = seqs
        d[f"c{i}_maps"] = maps
        d[f"c{i}_lens"] = lens
        # one-hot stored compactly: base code per (chunk, kmer_pos, sig_pos), -1 = all zero
        code = np.where(
            enc.reshape(n, kb + ka + 1, 4, L).sum(2) > 0,
            enc.reshape(n, kb + ka + 1, 4, L).argmax(2),
            -1,
        ).astype(np.int8)
        assert np.array_equal(
            (code[:, :, None, :] == np.arange(4)[None, None, :, None]).astype(np.float32).reshape(enc.shape),
            enc,
        )
        d[f"c{i}_enc_code"] = code
    np.savez_compressed(os.path.join(out, "encode_kmers.npz"), **d)
    print("encode_kmers: done")


def gen_trim(R, out):
    """data_chunks_core.trim_sb_chunk_context_core (src/remora/data_chunks_core.pyx:10-45),
    called as in CoreRemoraDataset.trim_sb_chunk_context (data_chunks.py:1536-1576)."""
    rng = np.random.default_rng(15)
    d = {}
    cases = [
        # stored cc, new cc, kb, ka, max_seq_len
        ((50, 50), (30, 25), 4, 4, 20),
        ((50, 50), (50, 25), 4, 4, 20),
        ((50, 50), (20, 50), 2, 3, 20),
        ((100, 100), (50, 50), 4, 4, 40),
        ((50, 50), (50, 50), 4, 4, 20),
    ]
    d["num_cases"] = np.asarray(len(cases))
    for i, (scc, cc, kb, ka, msl) in enumerate(cases):
        n = 37
        seqs, maps, lens = synth_chunks(rng, n, sum(scc), msl, kb, ka, 0.03, i % 2 == 1)
        # reference zero-fills nothing, but garbage maps beyond seq_len must not
        # terminate the `while` scans early in a data-dependent way: keep them
        d[f"c{i}_args"] = np.asarray([*scc, *cc, kb + ka], np.int64)
        d[f"c{i}_in_seqs"] = seqs.copy()
        d[f"c{i}_in_maps"] = maps.copy()
        d[f"c{i}_in_lens"] = lens.copy()
        st_diff = scc[0] - cc[0]
        m2 = (maps - st_diff).astype(np.int16)
        s2 = seqs.copy()
        l2 = lens.copy()
        R.data_chunks_core.trim_sb_chunk_context_core(*scc, *cc, kb + ka, s2, m2, l2)
        d[f"c{i}_out_seqs"] = s2
        d[f"c{i}_out_maps"] = m2
        d[f"c{i}_out_lens"] = l2
        # and the encode of the trimmed arrays (what the model finally sees)
        enc = R.encoded_kmers.compute_encoded_kmer_batch(kb, ka, s2, m2, l2)
        K = kb + ka + 1
        code = np.where(
            enc.reshape(n, K, 4, -1).sum(2) > 0, enc.reshape(n, K, 4, -1).argmax(2), -1
        ).astype(np.int8)
        d[f"c{i}_out_enc_code"] = code
    np.savez_compressed(os.path.join(out, "trim_chunk_context.npz"), **d)
    print("trim: done")


def gen_model_logits(R, out):
    """models/ConvLSTM_w_ref.py:39-58 and models/Conv_w_ref.py:44-62 forward in eval mode,
    non-trivial BN running stats. Weights + inputs + reference logits."""
    import torch

    rng = np.random.default_rng(16)
    specs = [
        # name, arch, size, (kb,ka), chunk_len, num_out, nchunks
        ("convlstm_s64_l100_o2", "ConvLSTM_w_ref", 64, (4, 4), 100, 2, 48),
        ("convlstm_s64_l200_o3", "ConvLSTM_w_ref", 64, (4, 4), 200, 3, 24),
        ("convlstm_s16_l100_o2", "ConvLSTM_w_ref", 16, (4, 4), 100, 2, 32),
        ("convlstm_s64_l100_k23", "ConvLSTM_w_ref", 64, (2, 3), 100, 4, 16),
        ("conv_s64_l100_o2", "Conv_w_ref", 64, (4, 4), 100, 2, 48),
        ("conv_s64_l100_o3", "Conv_w_ref", 64, (4, 4), 100, 3, 16),
        # round 6: `--size` is any int in the reference (src/remora/parsers.py:858-862); sizes outside {16, 32, 64} -
        # multiples of 16 above 64 (streamed-weight kernels) and sizes that are padded with zero channels (40 -> 64, 24 -> 32)
        ("convlstm_s96_l100_o2", "ConvLSTM_w_ref", 96, (4, 4), 100, 2, 24),
        ("convlstm_s128_l100_o2", "ConvLSTM_w_ref", 128, (4, 4), 100, 2, 24),
        ("conv_s96_l100_o2", "Conv_w_ref", 96, (4, 4), 100, 2, 24),
        ("convlstm_s40_l100_o2", "ConvLSTM_w_ref", 40, (4, 4), 100, 2, 24),
        ("conv_s24_l100_o3", "Conv_w_ref", 24, (4, 4), 100, 3, 16),
    ]
    for si, (name, arch, size, (kb, ka), L, num_out, n) in enumerate(specs):
        K = kb + ka + 1
        net = make_net(R, arch, size, K, num_out, seed=100 + si)
        msl = L // 5
        seqs, maps, lens = synth_chunks(rng, n, L, msl, kb, ka, 0.03, False)
        sigs = rng.standard_normal((n, 1, L)).astype(np.float32)
        enc = R.encoded_kmers.compute_encoded_kmer_batch(kb, ka, seqs, maps, lens)
        with torch.no_grad():
            logits = net(torch.from_numpy(sigs), torch.from_numpy(enc)).numpy()
            jit_logits = torch.jit.script(net)(torch.from_numpy(sigs), torch.from_numpy(enc)).numpy()
        assert np.abs(logits - jit_logits).max() < 1e-5
        # a dense (non one-hot) seqs input pins the general forward(sigs, seqs) contract
        dense = rng.standard_normal((8, 4 * K, L)).astype(np.float32)
        with torch.no_grad():
            dense_logits = net(torch.from_numpy(sigs[:8]), torch.from_numpy(dense)).numpy()
        d = state_to_np(net)
        d.update(
            arch=np.asarray(arch),
            params=np.asarray([size, kb, ka, L, num_out], np.int64),
            sigs=sigs,
            seqs=seqs,
            maps=maps,
            lens=lens,
            logits=logits,
            dense_seqs=dense,
            dense_logits=dense_logits,
        )
        np.savez_compressed(os.path.join(out, f"model_{name}.npz"), **d)
        print("model", name, "logit range", float(logits.min()), float(logits.max()))


def _ckpt(kcb, cc, mod_bases, mod_long_names, motifs, size, kmer_len, num_out, bsj=False, off=0):
    return dict(
        kmer_context_bases=kcb,
        chunk_context=cc,
        modified_base_labels=True,
        mod_bases=mod_bases,
        mod_long_names=mod_long_names,
        reverse_signal=False,
        refine_kmer_center_idx=-1,
        refine_do_rough_rescale=False,
        refine_scale_iters=-1,
        refine_algo="dwell_penalty",
        refine_half_bandwidth=5,
        base_start_justify=bsj,
        offset=off,
        pa_scaling=None,
        model_params=dict(size=size, kmer_len=kmer_len, num_out=num_out),
        motifs=motifs,
        refine_kmer_levels=None,
        refine_sd_arr=None,
        model_version=3,
    )


def gen_call_read_mods(R, out):
    """model_util.export_model_torchscript / load_model (model_util.py:115-176, 566-699)
    and inference.call_read_mods (inference.py:661-712) end to end on synthetic reads."""
    import json
    import tempfile

    import torch

    rng = np.random.default_rng(17)
    specs = [
        ("cg_5mc", "ConvLSTM_w_ref", (4, 4), (50, 50), ["m"], ["5mC"], [("CG", 0)], 2),
        ("allc_5hmc_5mc", "ConvLSTM_w_ref", (4, 4), (100, 100), ["h", "m"], ["5hmC", "5mC"], [("C", 0)], 3),
        ("conv_cg", "Conv_w_ref", (4, 4), (50, 50), ["m"], ["5mC"], [("CG", 0)], 2),
    ]
    reads = [
        ("r_long", synth_read(rng, 600), 500.0, 80.0),
        ("r_short", synth_read(rng, 9), 505.0, 75.0),
        ("r_n", synth_read(rng, 200, with_n=True), 495.0, 83.0),
        ("r_none", (np.full(40, 500, np.int16), np.arange(0, 41, 10, dtype=np.int64), np.array([0, 0, 3, 3])), 500.0, 80.0),
    ]
    # a model that carries a k-mer level table (like the released ONT models): RemoraRead.refine_signal_mapping
    # (rough re-scale + one dwell-penalty DP pass) runs inside call_read_mods; reads follow the table
    lrng = np.random.default_rng(91)
    ref_levels = lrng.normal(0, 1, 4**5).astype(np.float32)
    specs.append(("cg_5mc_refine", "ConvLSTM_w_ref", (4, 4), (50, 50), ["m"], ["5mC"], [("CG", 0)], 2))
    level_reads = [
        ("l_long", synth_levels_read(lrng, ref_levels, 5, 2, 700, 0.3), 508.0, 84.0),
        ("l_mid", synth_levels_read(lrng, ref_levels, 5, 2, 150, 0.2), 492.0, 77.0),
        ("l_stall", synth_levels_read(lrng, ref_levels, 5, 2, 320, 0.25, 70), 500.0, 80.0),
    ]
    generic_reads = reads
    for si, (name, arch, kcb, cc, mod_bases, mln, motifs, num_out) in enumerate(specs):
        K = sum(kcb) + 1
        net = make_net(R, arch, 64, K, num_out, seed=200 + si)
        ckpt = _ckpt(kcb, cc, mod_bases, mln, motifs, 64, K, num_out)
        reads = generic_reads
        if name.endswith("_refine"):
            from remora.refine_signal_map import SigMapRefiner

            ckpt.update(refine_kmer_levels=ref_levels, refine_kmer_center_idx=2, refine_do_rough_rescale=True,
                        refine_scale_iters=0, refine_sd_arr=np.asarray(SigMapRefiner().sd_arr, np.float32),
                        base_start_justify=True, offset=1)
            reads = level_reads
        with tempfile.TemporaryDirectory() as td:
            pt = os.path.join(td, "m.pt")
            R.model_util.export_model_torchscript(ckpt, net, pt)
            # raw meta.txt as written by the reference (JSON string) - data, not code
            extra = {"meta.txt": ""}
            torch.jit.load(pt, _extra_files=extra, map_location="cpu")
            meta_txt = extra["meta.txt"]
            model, md = R.model_util.load_model(pt, quiet=True, eval_only=True)
        d = state_to_np(net)
        d["arch"] = np.asarray(arch)
        d["meta_txt"] = np.asarray(meta_txt if isinstance(meta_txt, str) else meta_txt.decode())
        md_plain = {
            k: v
            for k, v in md.items()
            if k
            in (
                "motifs", "can_base", "mod_bases", "mod_long_names", "chunk_context", "chunk_len",
                "kmer_context_bases", "kmer_len", "base_start_justify", "offset", "reverse_signal",
                "pa_scaling", "motif", "alphabet_str",
            )
        }
        d["derived_md_json"] = np.asarray(json.dumps(md_plain))
        d["read_names"] = np.asarray([r[0] for r in reads])
        for rname, (dacs, s2s, int_seq), shift, scale in reads:
            def mk():
                return R.data_chunks.RemoraRead(
                    dacs=dacs.copy(), shift=shift, scale=scale, seq_to_sig_map=s2s.copy(),
                    int_seq=int_seq.copy(), read_id=rname,
                )
            d[f"{rname}_dacs"] = dacs
            d[f"{rname}_map"] = s2s
            d[f"{rname}_int_seq"] = int_seq
            d[f"{rname}_shift_scale"] = np.asarray([shift, scale], np.float64)
            nn_out, labels, pos = R.inference.call_read_mods(mk(), model, md)
            d[f"{rname}_nn_out"] = np.asarray(nn_out, np.float32)
            d[f"{rname}_labels"] = np.asarray(labels, np.int64)
            d[f"{rname}_pos"] = np.asarray(pos, np.int64)
            probs, _, pos2 = R.inference.call_read_mods(mk(), model, md, return_mod_probs=True)
            d[f"{rname}_probs"] = np.asarray(probs, np.float64)
            res = R.inference.call_read_mods(mk(), model, md, return_mm_ml_tags=True)
            if len(res) == 2:
                mm, ml = res
                d[f"{rname}_mm"] = np.asarray(mm)
                d[f"{rname}_ml"] = np.asarray(list(ml), np.uint8)
            else:  # no chunks -> 3 empty arrays (inference.py:698-699)
                d[f"{rname}_mm"] = np.asarray("<EMPTY3>")
                d[f"{rname}_ml"] = np.zeros(0, np.uint8)
            if nn_out.size and rname == "r_long":
                fo = int(pos[len(pos) // 2])
                o2, _, p2 = R.inference.call_read_mods(mk(), model, md, focus_offset=fo)
                d[f"{rname}_focus_offset"] = np.asarray(fo)
                d[f"{rname}_focus_nn_out"] = np.asarray(o2, np.float32)
                d[f"{rname}_focus_pos"] = np.asarray(p2, np.int64)
        np.savez_compressed(os.path.join(out, f"call_read_mods_{name}.npz"), **d)
        print("call_read_mods", name, "done")


def gen_post(R, out):
    """util.softmax_axis1 (util.py:182-186), util.format_mm_ml_tags (:485-537) and the
    label tally of validate.compute_metrics (validate.py:42-66) that the multi-GPU count
    reduction reproduces."""
    rng = np.random.default_rng(18)
    d = {}
    x = (rng.standard_normal((257, 3)) * 4).astype(np.float32)
    d["softmax_in"] = x
    d["softmax_out"] = R.util.softmax_axis1(x)
    seq = "".join(rng.choice(list("ACGT"), 400))
    poss = np.array([i for i, b in enumerate(seq) if b == "C"])
    rng.shuffle(poss)
    probs = rng.random((poss.size, 2))
    probs[0] = (1.0, 0.0)
    probs[1] = (0.999999, 1.0)
    mm, ml = R.util.format_mm_ml_tags(seq, poss, probs, ["h", "m"], "C")
    d["tags_seq"] = np.asarray(seq)
    d["tags_poss"] = poss.astype(np.int64)
    d["tags_probs"] = probs
    d["tags_mm"] = np.asarray(mm)
    d["tags_ml"] = np.asarray(list(ml), np.uint8)
    # argmax tally
    logits = (rng.standard_normal((1000, 3))).astype(np.float32)
    logits[5] = (1.0, 1.0, 0.5)  # tie -> first index
    labels = rng.integers(0, 3, 1000)
    pr = R.util.softmax_axis1(logits)
    pred = np.argmax(pr, axis=1)
    d["tally_logits"] = logits
    d["tally_labels"] = labels.astype(np.int64)
    d["tally_pred_counts"] = np.bincount(pred, minlength=3).astype(np.int64)
    conf = np.zeros((3, 3), np.int64)
    np.add.at(conf, (labels, pred), 1)
    d["tally_confusion"] = conf
    np.savez_compressed(os.path.join(out, "post_process.npz"), **d)
    print("post: done")


def gen_dataset_batches(R, out):
    """CoreRemoraDataset in-memory path: write_chunk (data_chunks.py:1376-1418) then
    iteration -> extract_batch (:1652-1676), i.e. RemoraRead.prepare_batches (:468-514)."""
    rng = np.random.default_rng(19)
    dacs, s2s, int_seq = synth_read(rng, 500)
    md = dict(
        sig_map_refiner=R.data_chunks.__dict__.get("SigMapRefiner", None),
    )
    from remora.refine_signal_map import SigMapRefiner

    model_metadata = dict(
        sig_map_refiner=SigMapRefiner(),
        chunk_context=[50, 50],
        kmer_context_bases=[4, 4],
        base_start_justify=False,
        offset=0,
        motifs=[("CG", 0)],
        mod_bases=["m"],
        mod_long_names=["5mC"],
    )
    read = R.data_chunks.RemoraRead(
        dacs=dacs, shift=500.0, scale=80.0, seq_to_sig_map=s2s, int_seq=int_seq, read_id="ds"
    )
    read.set_motif_focus_bases([R.util.Motif("CG", 0)])
    read.prepare_batches(model_metadata, 2048)
    assert len(read.batches) == 1
    sig, enc, labels, rfb = read.batches[0]
    K = 9
    code = np.where(
        enc.reshape(-1, K, 4, 100).sum(2) > 0, enc.reshape(-1, K, 4, 100).argmax(2), -1
    ).astype(np.int8)
    d = dict(
        dacs=dacs, map=s2s, int_seq=int_seq, shift_scale=np.asarray([500.0, 80.0]),
        signal=sig, enc_code=code, labels=labels, read_focus_bases=rfb,
    )
    np.savez_compressed(os.path.join(out, "prepare_batches.npz"), **d)
    print("prepare_batches: done", sig.shape)


def gen_real_reads(R, out):
    """BASELINE configs[0] shape on the reference's own test data (tests/data/can_reads.pod5 +
    can_mappings.bam, copied to tests/golden/data/): the files are parsed by remora_amd.io
    (pysam/pod5 are not installed here), the parsed records are fed to the REFERENCE's
    io.Read.add_alignment (src/remora/io.py:1972-2044, iter_signal calibration convention
    :466-472), into_remora_read (:2123-2177) and inference.call_read_mods (inference.py:661-712)."""
    import json
    import tempfile

    import torch

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from remora_amd import io as rio

    data = os.path.join(out, "data")
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
    from golden_util import pod5_reads_cpu

    for prefix in ("can", "mod"):  # both halves of BASELINE configs[0]; the same CG 5mC model calls both
        _gen_real_reads_one(R, out, data, prefix, rio, pod5_reads_cpu)


def _gen_real_reads_one(R, out, data, prefix, rio, pod5_reads_cpu):
    import tempfile

    import torch

    pods = {p.read_id: p for p in pod5_reads_cpu(os.path.join(data, f"{prefix}_reads.pod5"))}
    recs = list(rio.iter_bam_records(os.path.join(data, f"{prefix}_mappings.bam")))
    net = make_net(R, "ConvLSTM_w_ref", 64, 9, 2, seed=300)
    ckpt = _ckpt((4, 4), (50, 50), ["m"], ["5mC"], [("CG", 0)], 64, 9, 2)
    with tempfile.TemporaryDirectory() as td:
        pt = os.path.join(td, "m.pt")
        R.model_util.export_model_torchscript(ckpt, net, pt)
        extra = {"meta.txt": ""}
        torch.jit.load(pt, _extra_files=extra, map_location="cpu")
        model, md = R.model_util.load_model(pt, quiet=True, eval_only=True)
    # the weights travel once (real_reads_can.npz); the mod file holds the per-read results of the same model
    d = state_to_np(net) if prefix == "can" else {}
    d["meta_txt"] = np.asarray(extra["meta.txt"] if isinstance(extra["meta.txt"], str) else extra["meta.txt"].decode())
    d["num_records"] = np.asarray(len(recs))
    for i, rec in enumerate(recs):
        pod = pods[rec.query_name]
        read = R.io.Read(read_id=pod.read_id, dacs=pod.signal, shift_dacs_to_pa=pod.calibration_offset,
                         scale_dacs_to_pa=pod.calibration_scale)
        read.add_alignment(rec, parse_ref_align=False)
        rr = read.into_remora_read(False)
        d[f"r{i}_name"] = np.asarray(rec.query_name)
        d[f"r{i}_flag"] = np.asarray(rec.flag)
        d[f"r{i}_shift_scale"] = np.asarray([rr.shift, rr.scale], np.float64)
        d[f"r{i}_ndacs"] = np.asarray(rr.dacs.size)
        d[f"r{i}_dacs_crc"] = np.asarray(int(np.bitwise_xor.reduce(rr.dacs.astype(np.int64) * (np.arange(rr.dacs.size) % 251 + 1))))
        d[f"r{i}_map"] = np.asarray(rr.seq_to_sig_map, np.int64)
        d[f"r{i}_seq"] = np.asarray(rr.str_seq)
        nn_out, labels, pos = R.inference.call_read_mods(rr, model, md)
        d[f"r{i}_nn_out"] = np.asarray(nn_out, np.float32)
        d[f"r{i}_pos"] = np.asarray(pos, np.int64)
        mm, ml = R.inference.call_read_mods(read.into_remora_read(False), model, md, return_mm_ml_tags=True)
        d[f"r{i}_mm"] = np.asarray(mm)
        d[f"r{i}_ml"] = np.asarray(list(ml), np.uint8)
        # reference-anchored flavour: add_alignment with parse_ref_align (io.py:2045-2084; the record's
        # get_reference_sequence is remora_amd's MD-tag reconstruction, pinned separately on the NM tags and
        # the CpG ground truth of tests/data/can_gt.bed), into_remora_read(True), call_read_mods
        read = R.io.Read(read_id=pod.read_id, dacs=pod.signal, shift_dacs_to_pa=pod.calibration_offset,
                         scale_dacs_to_pa=pod.calibration_scale)
        read.add_alignment(rec, parse_ref_align=True)
        ra = read.into_remora_read(True)
        d[f"r{i}_ra_ref_seq"] = np.asarray(read.ref_seq)
        d[f"r{i}_ra_ref_to_signal"] = np.asarray(read.ref_to_signal, np.int64)
        d[f"r{i}_ra_region"] = np.asarray([read.ref_reg.ctg, read.ref_reg.strand, str(read.ref_reg.start), str(read.ref_reg.end)])
        d[f"r{i}_ra_map"] = np.asarray(ra.seq_to_sig_map, np.int64)
        d[f"r{i}_ra_ndacs"] = np.asarray(ra.dacs.size)
        nn_out, labels, pos = R.inference.call_read_mods(ra, model, md)
        d[f"r{i}_ra_nn_out"] = np.asarray(nn_out, np.float32)
        d[f"r{i}_ra_pos"] = np.asarray(pos, np.int64)
        mm, ml = R.inference.call_read_mods(read.into_remora_read(True), model, md, return_mm_ml_tags=True)
        d[f"r{i}_ra_mm"] = np.asarray(mm)
    np.savez_compressed(os.path.join(out, f"real_reads_{prefix}.npz"), **d)
    print(f"real_reads {prefix}:", len(recs), "records,", sum(int(d[f"r{i}_pos"].size) for i in range(len(recs))), "chunks,",
          sum(int(d[f"r{i}_ra_pos"].size) for i in range(len(recs))), "reference-anchored chunks")


def gen_real_read_branches(R, out):
    """The branches of io.Read.add_alignment / into_remora_read that the reference's test files never take
    (src/remora/io.py:1995-2044 reverse_signal, 1851-1856 median / MAD without sm / sd, 2001-2020 split reads with sp / pi,
    442-461 + 2159-2167 pa_scaling), driven on the same parsed records as gen_real_reads through edited tags
    (tests/golden_util.py: real_read_branch) -> real_read_branches.npz: per file, record and branch the composed shift /
    scale, the trimmed signal's checksum, the query-to-signal map and the logits of the CG 5mC model of real_reads_can.npz."""
    import tempfile

    import torch

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
    from golden_util import REAL_READ_BRANCHES, RecordWithTags, pod5_reads_cpu, real_read_branch
    from remora_amd import io as rio

    data = os.path.join(out, "data")
    net = make_net(R, "ConvLSTM_w_ref", 64, 9, 2, seed=300)  # the model of gen_real_reads
    ckpt = _ckpt((4, 4), (50, 50), ["m"], ["5mC"], [("CG", 0)], 64, 9, 2)
    with tempfile.TemporaryDirectory() as td:
        pt = os.path.join(td, "m.pt")
        R.model_util.export_model_torchscript(ckpt, net, pt)
        model, md = R.model_util.load_model(pt, quiet=True, eval_only=True)
    d, n_cases = {}, 0
    for prefix in ("can", "mod"):
        pods = {p.read_id: p for p in pod5_reads_cpu(os.path.join(data, f"{prefix}_reads.pod5"))}
        recs = list(rio.iter_bam_records(os.path.join(data, f"{prefix}_mappings.bam")))
        for i, rec in enumerate(recs):
            pod = pods[rec.query_name]
            for variant in REAL_READ_BRANCHES:
                read_id, dacs, rec_v, kw, _ = real_read_branch(variant, pod, rec, 1000 + i)
                init_kw = {}
                if "pa_scaling" in kw:  # iter_signal hands it to the constructor as well (io.py:457-472)
                    init_kw = dict(shift_pa_to_zc_pa=kw["pa_scaling"][0], scale_pa_to_zc_pa=kw["pa_scaling"][1])
                read = R.io.Read(read_id=read_id, dacs=dacs, shift_dacs_to_pa=pod.calibration_offset,
                                 scale_dacs_to_pa=pod.calibration_scale, **init_kw)
                read.add_alignment(rec_v, parse_ref_align=False, **kw)
                rr = read.into_remora_read(False)
                key = f"{prefix}_r{i}_{variant}"
                d[f"{key}_shift_scale"] = np.asarray([rr.shift, rr.scale], np.float64)
                d[f"{key}_ndacs"] = np.asarray(rr.dacs.size)
                d[f"{key}_dacs_crc"] = np.asarray(int(np.bitwise_xor.reduce(rr.dacs.astype(np.int64) * (np.arange(rr.dacs.size) % 251 + 1))))
                d[f"{key}_map"] = np.asarray(rr.seq_to_sig_map, np.int64)
                d[f"{key}_read_ids"] = np.asarray([read.read_id, read.child_read_id])
                nn_out, labels, pos = R.inference.call_read_mods(rr, model, md)
                d[f"{key}_nn_out"] = np.asarray(nn_out, np.float32)
                d[f"{key}_pos"] = np.asarray(pos, np.int64)
                n_cases += 1
        # the error of a split read whose parent is another read (io.py:2016-2018), once per file
        pod, rec = pods[recs[0].query_name], recs[0]
        read = R.io.Read(read_id=pod.read_id, dacs=pod.signal, shift_dacs_to_pa=pod.calibration_offset,
                         scale_dacs_to_pa=pod.calibration_scale)
        try:
            read.add_alignment(RecordWithTags(rec, add={"pi": "somebody-else"}), parse_ref_align=False)
            raise AssertionError("expected an error")
        except R.remora.RemoraError as e:
            d[f"{prefix}_split_mismatch_error"] = np.asarray(str(e))
    np.savez_compressed(os.path.join(out, "real_read_branches.npz"), **d)
    print("real_read_branches:", n_cases, "cases")


def gen_core_dataset(R, out):
    """On-disk CoreRemoraDataset (src/remora/data_chunks.py:926-1702) written by the reference from
    two synthetic labelled reads (chunk_context (50,50), k-mer (4,4)), then read back by the
    reference as stored AND with the dynamic overrides chunk_context (30,25) / k-mer (2,3), which
    drive trim_sb_kmer_context_bases (:1512-1534), trim_sb_chunk_context (:1536-1576) and
    extract_batch (:1652-1676).  The dataset directory itself is committed as fixture data."""
    import shutil

    from remora.refine_signal_map import SigMapRefiner

    rng = np.random.default_rng(21)
    ddir = os.path.join(out, "data", "core_dataset")
    shutil.rmtree(ddir, ignore_errors=True)
    os.makedirs(ddir)
    chunks, src_reads = [], []
    for ri in range(2):
        dacs, s2s, int_seq = synth_read(rng, 900 + 300 * ri)
        labels = rng.integers(0, 2, int_seq.size).astype(np.int64)
        read = R.data_chunks.RemoraRead(dacs=dacs, shift=500.0, scale=80.0, seq_to_sig_map=s2s, int_seq=int_seq,
                                        read_id=f"ds{ri}", labels=labels)
        read.set_motif_focus_bases([R.util.Motif("CG", 0)])
        chunks += list(read.iter_chunks((50, 50), (4, 4), False, 0))
        src_reads.append((dacs, s2s, int_seq, labels))
    md = R.data_chunks.DatasetMetadata(
        allocate_size=len(chunks) + 7, max_seq_len=20, mod_bases=["m"], mod_long_names=["5mC"],
        motif_sequences=["CG"], motif_offsets=[0], chunk_context=(50, 50), kmer_context_bases=(4, 4),
        sig_map_refiner=SigMapRefiner())
    ds = R.data_chunks.CoreRemoraDataset(data_path=ddir, mode="w", metadata=md)
    kept = 0
    for c in chunks:
        if c.seq_len <= 20:
            ds.write_chunk(c)
            kept += 1
    ds.write_metadata()
    ds.flush()
    ds.close_memmaps()
    d = {"num_chunks": np.asarray(kept)}
    for ri, (dacs, s2s, int_seq, labels) in enumerate(src_reads):  # the reads the dataset was cut from
        d[f"ds{ri}_dacs"], d[f"ds{ri}_map"], d[f"ds{ri}_int_seq"], d[f"ds{ri}_labels"] = dacs, s2s, int_seq, labels
    d["metadata_jsn"] = np.asarray(open(os.path.join(ddir, "metadata.jsn")).read())
    for tag, override in (("stored", None),
                          ("trim", {"chunk_context": (30, 25), "kmer_context_bases": (2, 3)}),
                          ("trimcc", {"chunk_context": (40, 50)}),
                          ("trimk", {"kmer_context_bases": (4, 1)})):
        rd = R.data_chunks.CoreRemoraDataset(data_path=ddir, mode="r", override_metadata=override, batch_size=64,
                                             super_batch_size=160, infinite_iter=False)
        sigs, codes, labs = [], [], []
        for batch in rd:
            enc = batch["enc_kmers"]
            K = enc.shape[1] // 4
            n, L = enc.shape[0], enc.shape[2]
            e4 = enc.reshape(n, K, 4, L)
            codes.append(np.where(e4.sum(2) > 0, e4.argmax(2), -1).astype(np.int8))
            sigs.append(batch["signal"])
            labs.append(batch["labels"])
        d[f"{tag}_signal"] = np.concatenate(sigs)
        d[f"{tag}_enc_code"] = np.concatenate(codes)
        d[f"{tag}_labels"] = np.concatenate(labs)
        d[f"{tag}_override"] = np.asarray(json.dumps(override))
        print("core_dataset", tag, d[f"{tag}_signal"].shape, d[f"{tag}_enc_code"].shape)
    np.savez_compressed(os.path.join(out, "core_dataset.npz"), **d)
    print(open(os.path.join(ddir, "metadata.jsn")).read()[:900])
    print(sorted(os.listdir(ddir)))


def synth_levels_read(rng, levels_arr, kmer_len, center_idx, nbases, noise=0.25, stall_every=0):
    """A read whose signal follows a k-mer level table: per base level + N(0, noise), dwell
    U[4,16] (optionally a long stall), DAC = 500 + 80 * norm, then a deliberately wrong
    (shift, scale) and a jittered mapping so that refinement has work to do."""
    int_seq = rng.integers(0, 4, nbases).astype(np.int64)
    lv = np.zeros(nbases, np.float32)
    for pos in range(nbases - kmer_len + 1):
        idx = 0
        for b in int_seq[pos : pos + kmer_len]:
            idx = idx * 4 + int(b)
        lv[pos + center_idx] = levels_arr[idx]
    dwell = rng.integers(4, 17, nbases)
    if stall_every:
        dwell[stall_every::stall_every] = rng.integers(150, 400, dwell[stall_every::stall_every].size)
    true_map = np.concatenate([[0], np.cumsum(dwell)]).astype(np.int64)
    norm = np.repeat(lv, dwell) + noise * rng.standard_normal(true_map[-1])
    dacs = np.round(500 + 80 * norm).astype(np.int16)
    jit = true_map.copy()
    jit[1:-1] += rng.integers(-3, 4, nbases - 1)
    jit = np.maximum.accumulate(np.clip(jit, 0, true_map[-1]))
    jit[0], jit[-1] = 0, true_map[-1]
    return dacs, jit, int_seq


def gen_refine(R, out):
    """Signal-mapping refinement: refine_signal_map_core.pyx (adjust_seq_band :31-74, extract_levels
    :87-101, seq_banded_dp :403-473) and SigMapRefiner.rough_rescale / refine_sig_map
    (refine_signal_map.py:366-408, :472-497) as driven by RemoraRead.refine_signal_mapping
    (data_chunks.py:267-308)."""
    from remora.refine_signal_map import (SigMapRefiner, compute_sig_band, convert_to_seq_band,
                                          refine_signal_mapping)
    from remora.refine_signal_map_core import adjust_seq_band, extract_levels, seq_banded_dp

    rng = np.random.default_rng(23)
    kmer_len, center = 5, 2
    levels_arr = rng.normal(0, 1, 4**kmer_len).astype(np.float32)
    d = {"kmer_levels": levels_arr, "center_idx": np.asarray(center)}
    reads = [("a", 400, 0.25, 0), ("b", 1500, 0.4, 0), ("c", 260, 0.15, 60), ("d", 40, 0.2, 0)]
    d["read_names"] = np.asarray([r[0] for r in reads])
    # function level: one DP per read and algorithm
    for name, nb, noise, stall in reads:
        dacs, s2s, int_seq = synth_levels_read(rng, levels_arr, kmer_len, center, nb, noise, stall)
        d[f"{name}_dacs"], d[f"{name}_map"], d[f"{name}_int_seq"] = dacs, s2s, int_seq
        lv = extract_levels(int_seq.astype(np.int32), levels_arr, kmer_len, center)
        d[f"{name}_levels"] = lv
        sig = ((dacs - 505.0) / 83.0).astype(np.float32)
        for hbw in (5, 2):
            band = convert_to_seq_band(compute_sig_band(s2s, lv, bhw=hbw))
            d[f"{name}_hbw{hbw}_band_raw"] = band.copy()
            adjust_seq_band(band, min_step=2)
            d[f"{name}_hbw{hbw}_band"] = band
            for algo in ("Viterbi", "dwell_penalty"):
                sdp = SigMapRefiner().sd_arr
                scores, path, tb, offs = seq_banded_dp(sig, lv, band, sdp, algo)
                pre = f"{name}_hbw{hbw}_{algo}_"
                d[pre + "path"] = np.asarray(path)
                if name != "b":  # keep the fixture small: full score/traceback bands for the short reads only
                    d[pre + "scores"], d[pre + "tb"] = np.asarray(scores), np.asarray(tb)
        d["sd_arr"] = np.asarray(SigMapRefiner().sd_arr, np.float32)
    # object level: RemoraRead.refine_signal_mapping with several refiner settings
    settings = [
        dict(do_rough_rescale=True, scale_iters=0, algo="dwell_penalty", half_bandwidth=5, rough_rescale_method="least_squares"),
        dict(do_rough_rescale=True, scale_iters=-1, algo="dwell_penalty", half_bandwidth=5, rough_rescale_method="theil_sen"),
        dict(do_rough_rescale=False, scale_iters=0, algo="Viterbi", half_bandwidth=3, rough_rescale_method="least_squares"),
        dict(do_rough_rescale=True, scale_iters=2, algo="dwell_penalty", half_bandwidth=5, rough_rescale_method="least_squares"),
    ]
    d["settings_json"] = np.asarray(json.dumps(settings))
    for si, st in enumerate(settings):
        ref = SigMapRefiner(_levels_array=levels_arr, center_idx=center, **st)
        for name, nb, noise, stall in reads:
            np.random.seed(1000 + si)
            read = R.data_chunks.RemoraRead(dacs=d[f"{name}_dacs"], shift=505.0, scale=83.0, seq_to_sig_map=d[f"{name}_map"].copy(),
                                            int_seq=d[f"{name}_int_seq"], read_id=name)
            read.refine_signal_mapping(ref)
            d[f"s{si}_{name}_map"] = np.asarray(read.seq_to_sig_map, np.int64)
            d[f"s{si}_{name}_shift_scale"] = np.asarray([read.shift, read.scale], np.float64)
    # k-mer table file -> SigMapRefiner (load_kmer_table, determine_dominant_pos, fix_gauge; :226-349)
    from itertools import product as _product

    trng = np.random.default_rng(77)
    base_lv = {"A": -1.2, "C": -0.3, "G": 0.5, "T": 1.4}
    os.makedirs(os.path.join(out, "data"), exist_ok=True)
    table_path = os.path.join(out, "data", "levels_4mer.txt")
    with open(table_path, "w") as fh:
        for kmer in _product("ACGT", repeat=4):
            # position 2 dominates, position 1 contributes a little, plus noise; one NaN entry (reads as 0)
            lvl = 80 + 12 * base_lv[kmer[2]] + 3 * base_lv[kmer[1]] + trng.normal(0, 0.8)
            fh.write("".join(kmer).lower() + "\t" + ("nan" if "".join(kmer) == "ACGT" else f"{lvl:.4f}") + "\n")
    for fix in (False, True):
        ref = SigMapRefiner(kmer_model_filename=table_path, do_rough_rescale=True, do_fix_guage=fix)
        tag = "fix" if fix else "raw"
        d[f"table_{tag}_levels"] = np.asarray(ref.levels_array, np.float64)
        d[f"table_{tag}_center"] = np.asarray(int(ref.center_idx))
        d[f"table_{tag}_stats"] = np.asarray(ref.kmer_idx_stats, np.float64)
    np.savez_compressed(os.path.join(out, "refine_signal_map.npz"), **d)
    print("refine: done;", "moved", int((d["s0_b_map"] != d["b_map"]).sum()), "of", d["b_map"].size, "breakpoints in read b;",
          "shift/scale", d["s0_b_shift_scale"])


def gen_batching(R, out):
    """inference.batch_reads (inference.py:171-262), unbatch_reads (:331-367) and unbatch (:370-415): reads
    whose chunks straddle fixed-size batches, error reads in between, two canonical-base models."""
    import queue
    from types import SimpleNamespace

    rng = np.random.default_rng(31)
    L, kb, ka, W = 20, 1, 1, 8
    mds = [dict(can_base="C", chunk_len=L, kmer_len=kb + ka + 1), dict(can_base="A", chunk_len=L, kmer_len=kb + ka + 1)]
    counts = {"C": [5, 13, None, 2, 9, 1], "A": [3, 7, None, 6, 4, 2]}  # None: error read
    d = {"batch_size": np.asarray(4), "chunk_len": np.asarray(L), "kmer_context_bases": np.asarray([kb, ka])}
    reads, prepped = [], []
    for ri in range(6):
        io_read = SimpleNamespace(read_id=f"read{ri}")
        reads.append(io_read)
        if counts["C"][ri] is None:
            prepped.append([(io_read, None, "Read prep error: spoofed")])
            d[f"r{ri}_err"] = np.asarray("Read prep error: spoofed")
            continue
        bases_chunks = {}
        for cb in "CA":
            n = counts[cb][ri]
            if n == 0:
                continue
            lens = rng.integers(2, W - kb - ka, n).astype(np.int16)
            seqs = np.full((n, W), -1, np.int8)
            maps = np.zeros((n, W - kb - ka + 1), np.int16)
            for c in range(n):
                seqs[c, : lens[c] + kb + ka] = rng.integers(0, 4, lens[c] + kb + ka)
                cuts = np.sort(rng.choice(np.arange(1, L), lens[c] - 1, replace=False))
                maps[c, : lens[c] + 1] = np.concatenate([[0], cuts, [L]])
            sig = rng.standard_normal((n, 1, L)).astype(np.float32)
            enc = R.encoded_kmers.compute_encoded_kmer_batch(kb, ka, seqs, maps, lens)
            rfb = rng.integers(0, 5000, n).astype(np.int64)
            bases_chunks[cb] = {"signal": sig, "enc_kmers": enc, "read_focus_bases": rfb}
            for k, v in (("signal", sig), ("sequence", seqs), ("mapping", maps), ("lengths", lens), ("rfb", rfb)):
                d[f"r{ri}_{cb}_{k}"] = v
        prepped.append([(io_read, bases_chunks, None)])
    prepped.append([])  # a read without valid mappings contributes nothing (prep_nn_input turns [] into an error)
    bq = queue.Queue()
    R.inference.batch_reads(iter(prepped), bq, 4, mds)
    batches = []
    while True:
        it = bq.get()
        if it is StopIteration:
            break
        batches.append(it)
    d["num_batches"] = np.asarray(len(batches))
    called = queue.Queue()
    for bi, (cb, b_sigs, b_enc, b_pos, b_reads) in enumerate(batches):
        d[f"b{bi}_can_base"] = np.asarray(cb)
        d[f"b{bi}_sigs"] = np.asarray(b_sigs)
        d[f"b{bi}_enc"] = np.asarray(b_enc).astype(np.uint8)
        d[f"b{bi}_pos"] = np.asarray(b_pos, np.int64)
        d[f"b{bi}_spans"] = np.asarray(json.dumps([[r.read_id, st, en, err] for r, st, en, err in b_reads]))
        # a stand-in network output that depends on the chunk: (mean signal, read position)
        nn_out = np.stack([b_sigs.mean(axis=(1, 2)), b_pos.astype(np.float32)], axis=1).astype(np.float32)
        d[f"b{bi}_nn_out"] = nn_out
        called.put((cb, SimpleNamespace(cpu=lambda a=nn_out: SimpleNamespace(numpy=lambda a=a: a)), b_pos, b_reads))
    called.put(StopIteration)
    rq = queue.Queue()
    R.inference.unbatch(called, rq, mds)
    done = []
    while True:
        it = rq.get()
        if it is StopIteration:
            break
        done.append(it)
    d["num_done"] = np.asarray(len(done))
    for di, (io_read, mod_calls, err) in enumerate(done):
        d[f"d{di}_read_id"] = np.asarray(io_read.read_id)
        d[f"d{di}_err"] = np.asarray("" if err is None else err)
        d[f"d{di}_bases"] = np.asarray([cb for cb, _, _ in mod_calls])
        for cb, nn_out, pos in mod_calls:
            d[f"d{di}_{cb}_nn_out"] = np.asarray(nn_out)
            d[f"d{di}_{cb}_pos"] = np.asarray(pos)
    np.savez_compressed(os.path.join(out, "batching.npz"), **d)
    print("batching:", len(batches), "batches;", len(done), "reads returned:", [str(x[0].read_id) + ("!" if x[2] else "") for x in done])


PREP_CONFIGS = {
    # name: (data, mod_base | None (= control), kwargs of extract_chunks / the dataset metadata)
    "can_ctrl": ("can", None, dict(chunk_context=(50, 50))),
    "mod_m": ("mod", ("m", "5mC"), dict(chunk_context=(50, 50))),
    "mod_h": ("mod", ("h", "5hmC"), dict(chunk_context=(50, 50), max_chunks_per_read=6)),
    "can_bc_bed": ("can", None, dict(chunk_context=(50, 50), basecall_anchor=True, bed="can_gt.bed", kmer_context_bases=(2, 3))),
    "can_bc": ("can", None, dict(chunk_context=(40, 60), basecall_anchor=True, max_chunks_per_read=9)),
    "can_ref_bed": ("can", None, dict(chunk_context=(50, 50), bed="can_gt.bed", max_chunks_per_read=200)),
    "can_refine": ("can", None, dict(chunk_context=(50, 50), refine=True, max_chunks_per_read=20)),
    "can_default": ("can", ("m", "5mC"), dict(offset=1, base_start_justify=True, motifs=[("CG", 0), ("CH", 0)])),
}


def _prep_defaults(kw):
    full = dict(chunk_context=(200, 200), kmer_context_bases=(4, 4), max_chunks_per_read=15, basecall_anchor=False,
                bed=None, refine=False, offset=0, base_start_justify=False, motifs=[("CG", 0)], min_samps_per_base=5)
    full.update(kw)
    return full


def gen_prepare(R, out):
    """`remora dataset prepare` (src/remora/prepare_train_data.py): the reference's own extract_chunks (:33-118)
    on io.Read objects built by the reference's add_alignment from the test POD5 + BAM files (parsed by
    remora_amd.io, pysam/pod5 being absent), in BAM order under np.random.seed(11); the chunks go through the
    reference's CoreRemoraDataset.write_chunk with the metadata extract_chunk_dataset builds (:165-191), then
    write_metadata + shuffle (:268-276).  Stored: the written rows of every array (padding columns masked) and
    the metadata.jsn text; plus, per read, the untouched order before the shuffle."""
    import shutil
    import tempfile

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
    from golden_util import dataset_rows
    from remora import prepare_train_data as rprep
    from remora.refine_signal_map import SigMapRefiner
    from remora_amd import io as rio

    data = os.path.join(out, "data")
    for fn in ("mod_reads.pod5", "mod_mappings.bam", "mod_gt.bed"):
        if not os.path.exists(os.path.join(data, fn)):
            shutil.copy(os.path.join(REF, "tests", "data", fn), os.path.join(data, fn))
            os.chmod(os.path.join(data, fn), 0o644)
    d = {"configs_json": np.asarray(json.dumps({k: [v[0], v[1], _prep_defaults(v[2])] for k, v in PREP_CONFIGS.items()}))}
    for name, (which, mod_base, kw) in PREP_CONFIGS.items():
        kw = _prep_defaults(kw)
        from golden_util import pod5_reads_cpu

        pods = {p.read_id: p for p in pod5_reads_cpu(os.path.join(data, f"{which}_reads.pod5"))}
        recs = [r for r in rio.iter_bam_records(os.path.join(data, f"{which}_mappings.bam"))
                if not (r.is_secondary or r.is_supplementary)]
        motifs = [R.util.Motif(*m) for m in kw["motifs"]]
        focus_ref_pos = None if kw["bed"] is None else R.io.parse_bed(os.path.join(data, kw["bed"]))
        refiner = SigMapRefiner()
        if kw["refine"]:
            refiner = SigMapRefiner(kmer_model_filename=os.path.join(data, "levels_4mer.txt"), do_rough_rescale=True,
                                    scale_iters=0, do_fix_guage=True)
        max_seq_len = sum(kw["chunk_context"]) // kw["min_samps_per_base"]
        td = tempfile.mkdtemp()
        os.makedirs(os.path.join(td, "ds"))
        ds = R.data_chunks.CoreRemoraDataset(
            data_path=os.path.join(td, "ds"), mode="w",
            metadata=R.data_chunks.DatasetMetadata(
                allocate_size=kw["max_chunks_per_read"] * len(recs), max_seq_len=max_seq_len,
                mod_bases=[] if mod_base is None else [mod_base[0]],
                mod_long_names=[] if mod_base is None else [mod_base[1]],
                motif_sequences=[m.raw_motif for m in motifs], motif_offsets=[m.focus_pos for m in motifs],
                extra_arrays={"read_ids": ("<U36", "Read identifier"),
                              "read_focus_bases": ("int64", "Position within read training sequence")},
                chunk_context=kw["chunk_context"], kmer_context_bases=kw["kmer_context_bases"], reverse_signal=False,
                pa_scaling=None, sig_map_refiner=refiner, base_start_justify=kw["base_start_justify"], offset=kw["offset"]))
        np.random.seed(11)
        errs = {}
        for rec in recs:
            pod = pods[rec.query_name]
            read = R.io.Read(read_id=pod.read_id, dacs=pod.signal, shift_dacs_to_pa=pod.calibration_offset,
                             scale_dacs_to_pa=pod.calibration_scale)
            read.add_alignment(rec)
            for chunks, err in rprep.extract_chunks(
                    [(read, None)], 0 if mod_base is None else 1, motifs, focus_ref_pos, refiner, kw["max_chunks_per_read"],
                    kw["chunk_context"], kw["kmer_context_bases"], kw["base_start_justify"], kw["offset"], kw["basecall_anchor"]):
                if chunks is None:
                    errs[err] = errs.get(err, 0) + 1
                    continue
                for ch in chunks:
                    if ch.seq_len > max_seq_len:
                        errs["Sequence too long"] = errs.get("Sequence too long", 0) + 1
                        continue
                    ds.write_chunk(ch)
        ds.write_metadata()
        ds.flush()
        _, pre = dataset_rows(os.path.join(td, "ds"))
        d[f"{name}_preshuffle__read_ids"] = pre["read_ids"]
        d[f"{name}_preshuffle__read_focus_bases"] = pre["read_focus_bases"]
        ds.shuffle()
        ds.flush()
        md, rows = dataset_rows(os.path.join(td, "ds"))
        for k, v in rows.items():
            d[f"{name}__{k}"] = v
        d[f"{name}__metadata_jsn"] = np.asarray(open(os.path.join(td, "ds", "metadata.jsn")).read())
        if os.path.exists(os.path.join(td, "ds", "kmer_table.npy")):
            d[f"{name}__kmer_table"] = np.load(os.path.join(td, "ds", "kmer_table.npy"))
        d[f"{name}__errs_json"] = np.asarray(json.dumps(errs))
        print("prepare", name, "chunks", rows["labels"].size, "of alloc", md["allocate_size"], "errs", errs,
              "maxlen", int(rows["sequence_lengths"].max()))
        shutil.rmtree(td)
    np.savez_compressed(os.path.join(out, "prepared_datasets.npz"), **d)


def gen_remora_dataset(R, out):
    """RemoraDataset (src/remora/data_chunks.py:1806-2276) over the prepared datasets of gen_prepare: merged
    metadata, batch split, the batches the reference yields (finite and wrapping infinite iteration, label
    conversion when the datasets carry different modified bases, context overrides as `validate
    from_remora_dataset` applies them), label counts, config + hashes, head / train_test_split, epoch summary;
    and validate.ValidationLogger.run_validation (validate.py:190-259) + compute_metrics (:42-66) on them."""
    import shutil
    import tempfile

    import torch

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
    from golden_util import materialise_dataset

    g = np.load(os.path.join(out, "prepared_datasets.npz"))
    td = tempfile.mkdtemp()
    dirs = {n: materialise_dataset(g, n, os.path.join(td, n)) for n in ("can_ctrl", "mod_m", "mod_h", "can_bc_bed")}
    DC = R.data_chunks
    d = {}

    def batches_to(d, tag, it, limit=None):
        codes, sigs, labs = [], [], []
        for bi, (enc, sig, lab) in enumerate(it):
            enc = enc.numpy()
            n, K4, L = enc.shape
            e4 = enc.reshape(n, K4 // 4, 4, L)
            codes.append(np.where(e4.sum(2) > 0, e4.argmax(2), -1).astype(np.int8))
            sigs.append(sig.numpy())
            labs.append(lab.numpy())
            if limit is not None and bi + 1 >= limit:
                break
        d[f"{tag}_nbatch"] = np.asarray(len(codes))
        d[f"{tag}_bsizes"] = np.asarray([c.shape[0] for c in codes])
        d[f"{tag}_enc_code"], d[f"{tag}_signal"], d[f"{tag}_labels"] = map(np.concatenate, (codes, sigs, labs))

    for name, path in dirs.items():
        d[f"hash_{name}"] = np.asarray(DC.CoreRemoraDataset.hash(path))
    # mix1: the reference's tests/conftest.py `chunks` fixture shape (two datasets, 0.5 / 0.5), finite
    cfg1 = os.path.join(td, "mix1.cfg")
    json.dump([[dirs["can_ctrl"], 0.5], [dirs["mod_m"], 0.5]], open(cfg1, "w"))
    ds = DC.RemoraDataset.from_config(cfg1, ds_kwargs={"infinite_iter": False}, batch_size=32)
    d["mix1_batch_sizes"] = np.asarray(ds.batch_sizes)
    d["mix1_props"] = np.asarray(ds.props)
    d["mix1_label_counts"] = np.asarray(ds.get_label_counts())
    d["mix1_label_summary"] = np.asarray(ds.label_summary)
    d["mix1_mod_bases"] = np.asarray(json.dumps([ds.metadata.mod_bases, ds.metadata.mod_long_names, list(map(list, ds.metadata.motifs))]))
    d["mix1_epoch_summary"] = np.asarray(ds.epoch_summary(10).replace(td, "<TD>"))
    batches_to(d, "mix1", iter(ds))
    hd = ds.head(40)
    d["mix1_head_sizes"] = np.asarray([s.size for s in hd.datasets])
    batches_to(d, "mix1_head", iter(hd))
    trn, tst = ds.train_test_split(25)
    d["mix1_split_sizes"] = np.asarray([[s.size for s in trn.datasets], [s.size for s in tst.datasets]])
    batches_to(d, "mix1_test", iter(tst))
    batches_to(d, "mix1_train", iter(trn), limit=5)
    # mix2: three datasets, two modified bases -> label conversion; infinite iteration wrapping around small
    # super batches; nested config with weights
    sub = os.path.join(td, "sub.cfg")
    json.dump([[dirs["mod_m"], 3], [dirs["mod_h"], 1]], open(sub, "w"))
    cfg2 = os.path.join(td, "mix2.cfg")
    json.dump([[dirs["can_ctrl"], 2], [sub, 3]], open(cfg2, "w"))
    paths, props, hashes = DC.parse_dataset_config(cfg2)
    d["mix2_props"] = np.asarray(props)
    d["mix2_paths"] = np.asarray([os.path.basename(p) for p in paths])
    d["mix2_hashes"] = np.asarray(hashes)
    ds = DC.RemoraDataset([DC.CoreRemoraDataset(p) for p in paths], props, hashes, batch_size=50, super_batch_size=70)
    d["mix2_batch_sizes"] = np.asarray(ds.batch_sizes)
    d["mix2_mod_bases"] = np.asarray(json.dumps([ds.metadata.mod_bases, ds.metadata.mod_long_names]))
    d["mix2_label_conv"] = np.asarray(json.dumps([None if s.label_conv is None else s.label_conv.tolist() for s in ds.datasets]))
    d["mix2_label_counts"] = np.asarray(ds.get_label_counts())
    d["mix2_config_json"] = np.asarray(json.dumps([[os.path.basename(c[0])] + list(c[1:]) for c in ds.get_config()]))
    batches_to(d, "mix2", iter(ds), limit=9)
    # mix3: as `remora validate from_remora_dataset` loads it (parsers.py:1918-1941): extra arrays dropped and
    # the model's (smaller) contexts applied; then run_validation with a ConvLSTM of that shape
    over = {"extra_arrays": {}, "kmer_context_bases": (2, 3), "chunk_context": (45, 40)}
    ds = DC.RemoraDataset([DC.CoreRemoraDataset(p, override_metadata=dict(over), infinite_iter=False, do_check_super_batches=True)
                           for p in paths], props, hashes, batch_size=64)
    batches_to(d, "mix3", iter(ds))
    ds.load_all_batches()
    d["mix3_label_counts_loaded"] = np.asarray(ds.get_label_counts())
    net = make_net(R, "ConvLSTM_w_ref", 32, 6, 3, seed=811)
    for k, v in state_to_np(net, prefix="mix3_w__").items():
        d[k] = v
    val = R.validate.ValidationLogger(open(os.devnull, "w"))
    for tag, model_mods in (("hm", ["h", "m"]), ("m_only", ["m"])):
        if tag == "m_only":
            net2 = make_net(R, "ConvLSTM_w_ref", 32, 6, 2, seed=812)
            for k, v in state_to_np(net2, prefix="mix3b_w__").items():
                d[k] = v
        ms = val.run_validation(net if tag == "hm" else net2, model_mods, torch.nn.CrossEntropyLoss(), ds, 0.1, disable_pbar=True)
        d[f"mix3_{tag}_metrics"] = np.asarray([ms.loss, ms.acc, ms.num_calls, ms.filt_frac, ms.filt_acc, ms.filt_thresh], np.float64)
        d[f"mix3_{tag}_conf"], d[f"mix3_{tag}_filt_conf"] = np.asarray(ms.conf_mat), np.asarray(ms.filt_conf_mat)
    # compute_metrics / add_unmodeled_labels on random inputs
    rng = np.random.default_rng(5)
    logits = rng.normal(0, 2, (500, 3))
    probs = R.util.softmax_axis1(logits)
    labels = rng.integers(0, 3, 500)
    d["cm_probs"], d["cm_labels"] = probs, labels
    for fi, frac in enumerate((0.1, 0.0, 0.5)):
        acc, conf, ff, facc, fconf, thr = R.validate.compute_metrics(probs, labels, frac)
        d[f"cm{fi}_scalars"] = np.asarray([frac, acc, ff, facc, thr], np.float64)
        d[f"cm{fi}_conf"], d[f"cm{fi}_filt_conf"] = np.asarray(conf), np.asarray(fconf)
    d["aul_in"] = logits[:7, :2].astype(np.float32)
    d["aul_out_1"] = R.validate.add_unmodeled_labels(d["aul_in"], np.array([1]))
    d["aul_out_2"] = R.validate.add_unmodeled_labels(d["aul_in"], np.array([2]))
    d["aul_out_13"] = R.validate.add_unmodeled_labels(d["aul_in"], np.array([1, 3]))
    for fi, (tot, pr) in enumerate(((32, [0.5, 0.5]), (50, [0.4, 0.45, 0.15]), (3, [0.98, 0.01, 0.01]), (2048, [0.7, 0.2, 0.05, 0.05]))):
        d[f"split{fi}"] = np.asarray(DC.compute_best_split(tot, np.asarray(pr)))
        d[f"split{fi}_in"] = np.asarray([tot] + pr, np.float64)
    np.savez_compressed(os.path.join(out, "remora_dataset.npz"), **d)
    print("remora_dataset: mix1", d["mix1_nbatch"], d["mix1_batch_sizes"], d["mix1_label_counts"], "| mix2", d["mix2_batch_sizes"],
          str(d["mix2_mod_bases"]), str(d["mix2_label_conv"]), d["mix2_label_counts"], "| mix3", d["mix3_nbatch"],
          d["mix3_hm_metrics"], d["mix3_m_only_metrics"])
    shutil.rmtree(td)


def gen_batch_params(R, out):
    """CoreRemoraDataset.adjust_batch_params (src/remora/data_chunks.py:1471-1510) over a grid of dataset /
    batch / super-batch sizes and sample fractions, and one seeded iteration with super_batch_sample_frac
    (np.random.choice inside load_super_batch, :1618-1626) over a prepared dataset."""
    import shutil
    import tempfile

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
    from golden_util import materialise_dataset

    g = np.load(os.path.join(out, "prepared_datasets.npz"))
    td = tempfile.mkdtemp()
    path = materialise_dataset(g, "can_ctrl", os.path.join(td, "can_ctrl"))
    rows = []
    for bs in (1, 7, 16, 64, 300):
        for sbs in (10, 64, 100, 205, 1000):
            for frac in (None, 0.01, 0.1, 0.5, 0.99, 1.0):
                ds = R.data_chunks.CoreRemoraDataset(path, batch_size=bs, super_batch_size=sbs, super_batch_sample_frac=frac,
                                                     infinite_iter=False)
                cps, sel = ds.adjust_batch_params()
                rows.append([bs, sbs, -1.0 if frac is None else frac, cps, -1 if sel is None else sel, ds.batch_size,
                             ds.super_batch_size])
    d = {"grid": np.asarray(rows, np.float64)}
    for tag, inf in (("finite", False), ("infinite", True)):
        ds = R.data_chunks.CoreRemoraDataset(path, batch_size=16, super_batch_size=64, super_batch_sample_frac=0.5,
                                             infinite_iter=inf)
        np.random.seed(5)
        labs, fbs, ids, sizes = [], [], [], []
        for bi, b in enumerate(ds):
            labs.append(b["labels"]); fbs.append(b["read_focus_bases"]); ids.append(b["read_ids"]); sizes.append(b["labels"].size)
            if bi >= 9:
                break
        d[f"frac_{tag}_sizes"] = np.asarray(sizes)
        d[f"frac_{tag}_read_focus_bases"] = np.concatenate(fbs)
        d[f"frac_{tag}_read_ids"] = np.concatenate(ids)
    np.savez_compressed(os.path.join(out, "dataset_batch_params.npz"), **d)
    print("batch_params:", len(rows), "grid rows; frac iteration sizes", d["frac_finite_sizes"], d["frac_infinite_sizes"])
    shutil.rmtree(td)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    R = import_reference()
    gens = dict(
        parse_move_tag=gen_parse_move_tag,
        seq_motif=gen_seq_motif,
        extract_chunks=gen_extract_chunks,
        extract_chunk_padding=gen_extract_chunk_padding,
        encode_kmers=gen_encode_kmers,
        trim=gen_trim,
        model_logits=gen_model_logits,
        call_read_mods=gen_call_read_mods,
        post=gen_post,
        dataset_batches=gen_dataset_batches,
        real_reads=gen_real_reads,
        real_read_branches=gen_real_read_branches,
        core_dataset=gen_core_dataset,
        refine=gen_refine,
        batching=gen_batching,
        prepare=gen_prepare,
        remora_dataset=gen_remora_dataset,
        batch_params=gen_batch_params,
    )
    for name, fn in gens.items():
        if args.only and name not in args.only.split(","):
            continue
        fn(R, out)


if __name__ == "__main__":
    main()
