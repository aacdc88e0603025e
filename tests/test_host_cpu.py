"""CPU-only tests: host-side logic against the golden vectors, the C-ABI library loads and
exports every symbol include/remora_hip.h declares, the product path fails loudly without a
GPU, the synthetic generators, and the world_size-2 (gloo) count reduction."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden


# ---- C ABI ---------------------------------------------------------------------------------
def _header_functions():
    hdr = open(os.path.join(ROOT, "include", "remora_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rmr_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import ctypes

    from remora_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    names = _header_functions()
    assert len(names) >= 20
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/remora_hip.h but not exported"
    # and the ctypes binding covers exactly the header
    assert sorted(_lib.SIGNATURES) == names
    assert b"gfx950" in _lib.lib().rmr_version()
    # the documents quote the header's count (checked here instead of maintained by hand)
    for doc, pat in (("DESIGN.md", r"exports exactly the (\d+) `extern \"C\"` functions"), ("INTEGRATION.md", r"whose (\d+) entry points")):
        m = re.search(pat, open(os.path.join(ROOT, doc)).read())
        assert m and int(m.group(1)) == len(names), (doc, m and m.group(1), len(names))


def _product_env_names():
    """Every RMR_* / REMORA_* name the product reads: (all names, those only the experiment build `make abl` reads)."""
    every, abl = set(), set()
    pkg = os.path.join(ROOT, "remora_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                txt = open(os.path.join(dp, f)).read()
                every |= set(re.findall(r'"((?:RMR|REMORA)_[A-Z0-9_]+)"', txt))
                abl |= set(re.findall(r'abl_int\("((?:RMR|REMORA)_[A-Z0-9_]+)"', txt))
    abl |= {"RMR_DUMP_CAT", "RMR_FUSED_DUMP_X"}  # getenv under #ifdef RMR_TIMING_ABLATIONS (engine.hip)
    return every, abl


def test_every_environment_switch_is_in_the_design_table_and_there_are_few():
    """Review item: 81 switch names had accumulated, most of them settled A/Bs shipping as untested configurations.  What the
    shipped library and the Python host read now is the table of DESIGN.md section 11, name for name, at most 38 of them; the knobs
    of the experiment build (abl_int / #ifdef RMR_TIMING_ABLATIONS) are listed beside it and are not read by the product."""
    every, abl = _product_env_names()
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    at = design.index("## 11. Environment switches")
    table = design[at : design.index("\n## ", at + 5)] if "\n## " in design[at + 5 :] else design[at:]
    listed = set(re.findall(r"`((?:RMR|REMORA)_[A-Z0-9_]+)`", table))
    assert listed == every, (sorted(every - listed), sorted(listed - every))
    assert len(every - abl) <= 38, sorted(every - abl)
    eng = open(os.path.join(ROOT, "remora_amd", "csrc", "engine.hip")).read()
    for name in ("RMR_DUMP_CAT", "RMR_FUSED_DUMP_X"):  # really behind the experiment-build macro
        before = eng[: eng.index(f'getenv("{name}")')]
        assert before.rfind("#ifdef RMR_TIMING_ABLATIONS") > before.rfind("#endif"), name


def test_library_embeds_gfx950_code_object():
    from remora_amd import _lib

    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"conv_mfma_kernel" in blob and b"lstm_head_kernel" in blob


def test_no_gpu_fails_loudly_not_silently():
    """No CPU fallback: without a GPU every compute entry point raises RemoraError."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from remora_amd import RemoraError
    from remora_amd.encoded_kmers import compute_encoded_kmer_batch
    from remora_amd.io import parse_move_tag
    from remora_amd.model_util import model_from_state
    from remora_amd import synth

    with pytest.raises(RemoraError, match="no GPU|no HIP device"):
        compute_encoded_kmer_batch(4, 4, np.zeros((1, 28), np.int8), np.zeros((1, 21), np.int16), np.ones(1, np.int16))
    with pytest.raises(RemoraError):
        parse_move_tag([5, 1, 0, 1], 15)
    with pytest.raises(RemoraError):
        model_from_state(synth.synth_state(), dict(chunk_context=(50, 50)))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "remora_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "libremora_oracle" not in src, f


# ---- host logic vs golden ----------------------------------------------------------------------
def test_util_seq_motif_golden():
    from remora_amd import util

    g = golden("seq_motif.npz")
    for si in range(int(g["num_seqs"])):
        s = str(g[f"s{si}_str"])
        int_seq = util.seq_to_int(s)
        assert np.array_equal(int_seq, g[f"s{si}_int"])
        assert util.int_to_seq(int_seq) == s.replace("N", "N")
        for mi in range(int(g["num_motif_sets"])):
            if int(g[f"s{si}_m{mi}_skipped"]):
                continue
            motifs = [util.Motif(str(a), int(b)) for a, b in zip(g[f"m{mi}_seqs"], g[f"m{mi}_offs"])]
            fb = util.find_focus_bases_in_int_sequence(int_seq, motifs)
            assert np.array_equal(fb, g[f"s{si}_m{mi}_focus"]), (si, mi)
    for mi in range(int(g["num_motif_sets"])):
        for raw, off, nraw, noff in zip(g[f"m{mi}_seqs"], g[f"m{mi}_offs"], g[f"m{mi}_norm_seqs"], g[f"m{mi}_norm_offs"]):
            m = util.Motif(str(raw), int(off))
            assert m.to_tuple() == (str(nraw), int(noff))


def test_motif_errors_and_match():
    from remora_amd import RemoraError, util

    with pytest.raises(RemoraError):
        util.Motif("CX", 0)
    with pytest.raises(RemoraError):
        util.Motif("CG", 2)
    with pytest.raises(RemoraError):
        util.Motif("CG", "a")
    m = util.Motif("CG", 0)
    seq = util.seq_to_int("ACGTCG")
    assert m.match(seq, 1) and m.match(seq, 4) and not m.match(seq, 0) and not m.match(seq, 5)
    assert list(m.findall(seq)) == [1, 4]
    assert m.focus_base == "C" and m.num_bases_after_focus == 1


def test_motif_match_many_equals_match():
    """The vectorised form used per batch of focus bases agrees with `match` everywhere, motifs hanging over either end
    and motifs whose focus sits in stripped leading Ns included."""
    from remora_amd import RemoraError, util

    rng = np.random.default_rng(5)
    checked = 0
    for _ in range(1500):
        n = int(rng.integers(1, 12))
        seq = rng.integers(-1, 4, n).astype(np.int64)
        raw = "".join(rng.choice(list("ACGTNRYWH"), int(rng.integers(1, 6))))
        try:
            m = util.Motif(raw, int(rng.integers(0, len(raw))))
        except RemoraError:
            continue
        pos = rng.integers(0, n, 6)
        assert np.array_equal(m.match_many(seq, pos), np.array([m.match(seq, int(p)) for p in pos], bool)), (raw, seq, pos)
        checked += 1
    assert checked > 1000
    assert util.Motif("CG", 0).match_many(seq, np.zeros(0, np.int64)).size == 0


def test_post_process_golden():
    from remora_amd import util

    g = golden("post_process.npz")
    assert np.allclose(util.softmax_axis1(g["softmax_in"]), g["softmax_out"], rtol=0, atol=1e-7)
    mm, ml = util.format_mm_ml_tags(str(g["tags_seq"]), g["tags_poss"], g["tags_probs"], ["h", "m"], "C")
    assert mm == str(g["tags_mm"])
    assert np.array_equal(np.asarray(list(ml), np.uint8), g["tags_ml"])
    assert util.format_mm_ml_tags("ACGT", np.array([], int), np.zeros((0, 1)), ["m"], "C")[0] == ""


@pytest.mark.parametrize("name", ["cg_5mc", "allc_5hmc_5mc", "conv_cg"])
def test_add_derived_metadata_golden(name):
    from remora_amd.model_util import add_derived_metadata

    g = golden(f"call_read_mods_{name}.npz")
    md = json.loads(str(g["meta_txt"]))
    add_derived_metadata(md)
    ref = json.loads(str(g["derived_md_json"]))
    for k, v in ref.items():
        assert json.loads(json.dumps(md[k])) == v, k
    assert md["sig_map_refiner"].is_loaded is False
    assert not any(k.startswith("refine_") for k in md)


def test_state_to_blob_layout_matches_engine_count():
    import ctypes

    from remora_amd import _lib as L
    from remora_amd import synth
    from remora_amd.engine import detect_arch, state_to_blob

    for arch, size, K, no in (("conv_lstm", 64, 9, 2), ("conv_lstm", 16, 6, 4), ("conv_only", 64, 9, 3), ("conv_only", 32, 9, 2),
                              ("conv_lstm", 48, 9, 2), ("conv_lstm", 128, 9, 2), ("conv_only", 200, 9, 2), ("conv_lstm", 1, 9, 2)):
        st = synth.synth_state(arch, size, K, no)
        a, s, k, n, blob = state_to_blob(st)
        assert (a, s, k, n) == (arch, size, K, no)
        desc = L.ModelDesc(0 if arch == "conv_lstm" else 1, size, K, no, 100, 0)
        assert L.lib().rmr_model_weight_count(ctypes.byref(desc)) == blob.size
    # 16-bit above 64 channels: bf16 / f16 on the streamed kernels, in steps of 32 channels
    for size, dtype, want in ((96, 1, 96), (80, 1, 96), (80, 0, 80), (130, 4, 160), (256, 1, 256), (40, 4, 64)):
        d16 = L.ModelDesc(0, size, 9, 2, 100, dtype)
        assert L.lib().rmr_model_padded_size(ctypes.byref(d16)) == want and L.lib().rmr_model_weight_count(ctypes.byref(d16)) > 0
    for bad in (L.ModelDesc(0, 257, 9, 2, 100, 0), L.ModelDesc(0, 0, 9, 2, 100, 0), L.ModelDesc(0, 96, 9, 2, 100, 5), L.ModelDesc(0, 260, 9, 2, 100, 1),
                L.ModelDesc(1, 64, 9, 2, 100, 1), L.ModelDesc(0, 64, 9, 17, 100, 0)):  # too wide; empty; split dtype above 64; too wide in 16 bits; 16-bit conv_only
        assert L.lib().rmr_model_weight_count(ctypes.byref(bad)) == 0 and L.lib().rmr_model_padded_size(ctypes.byref(bad)) == 0
    from remora_amd import RemoraError

    with pytest.raises(RemoraError):
        detect_arch({"sig_conv1", "fc"})


def test_remora_read_host_semantics():
    from remora_amd import RemoraError
    from remora_amd.data_chunks import RemoraRead

    r = RemoraRead.test_read()
    assert r.str_seq == "ACGT" * 5 and r.seq_to_sig_map[-1] == 200
    r.check()
    with pytest.raises(RemoraError):
        RemoraRead(np.zeros(10), 0.0, 1.0, np.arange(4))
    bad = RemoraRead(np.zeros(10), 0.0, 1.0, np.array([0, 5, 9]), int_seq=np.array([0, 1]))
    with pytest.raises(RemoraError, match="mapping end"):
        bad.check()
    r2 = r.copy()
    assert r2 is not r and np.array_equal(r2.int_seq, r.int_seq)
    from remora_amd.util import Motif

    r.set_motif_focus_bases([Motif("CG", 0)])
    assert sorted(r.focus_bases) == [1, 5, 9, 13, 17]


# ---- synthetic generators -------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", ["C100", "C200"])
def test_synth_chunks_are_valid_dataset_rows(cfg):
    from oracle import oracle as O
    from remora_amd import synth

    d = synth.synth_chunks_config(cfg, 3000, shard=1)
    cc, kcb, msl, num_out, cg = synth.CONFIGS[cfg]
    L = sum(cc)
    m, l, s = d["sequence_to_signal_mapping"], d["sequence_lengths"], d["sequence"]
    assert m.dtype == np.int16 and s.dtype == np.int8 and l.dtype == np.int16 and d["signal"].dtype == np.float32
    assert m.shape == (3000, msl + 1) and s.shape == (3000, msl + 8) and d["signal"].shape == (3000, 1, L)
    for c in range(3000):
        row = m[c, : l[c] + 1]
        assert row[0] == 0 and row[-1] == L and np.all(np.diff(row) > 0)
        assert np.all(m[c, l[c] + 1:] == 0) and np.all(s[c, l[c] + 8:] == -1) and np.all(s[c, : l[c] + 8] >= 0)
        p = np.searchsorted(row, L // 2, side="right") - 1
        assert s[c, 4 + p] == 1 and (not cg or s[c, 4 + p + 1] == 2)
    enc = O.compute_encoded_kmer_batch(*kcb, s, m, l)
    assert np.all(enc.sum(axis=(1, 2)) == 9 * L)
    d2 = synth.synth_chunks_config(cfg, 3000, shard=1)
    assert all(np.array_equal(d[k], d2[k]) for k in ("signal", "sequence", "sequence_lengths"))
    d3 = synth.synth_chunks_config(cfg, 3000, shard=2)
    assert not np.array_equal(d["signal"], d3["signal"])


def test_synth_state_loads_into_reference_shaped_module():
    from oracle import torch_ref
    from remora_amd import synth

    for arch in ("conv_lstm", "conv_only"):
        net = torch_ref.from_state(synth.synth_state(arch, 64, 9, 3))
        assert sum(p.numel() for p in net.parameters()) > 100000


# ---- multi-GPU layer on CPU (gloo, world size 2) -------------------------------------------------------
def test_shard_range_partitions_exactly():
    from remora_amd.dist import shard_range

    for n in (0, 1, 7, 1000, 1_000_003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch
from remora_amd import dist as rdist
rank, world, local = rdist.init_process_group("gloo")
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "post_process.npz"))
logits = g["tally_logits"]
lo, hi = rdist.shard_range(logits.shape[0], rank, world)
pred = logits[lo:hi].argmax(1)               # np.argmax: first maximum, as validate.py:42-45
counts = np.bincount(pred, minlength=3).astype(np.int64)
total = rdist.allreduce_counts(counts.copy())
t = torch.from_numpy(counts.copy())
total_t = rdist.allreduce_counts(t)
mx = rdist.allreduce_max_float(float(rank + 1))
rows = rdist.allgather_counts(counts)        # what bench.py reports as label_counts_per_rank
if rank == 0:
    print(json.dumps({"total": total.tolist(), "total_t": total_t.tolist(), "max": mx, "n": [int(lo), int(hi)],
                      "rows": rows.tolist(), "mine": counts.tolist()}))
torch.distributed.destroy_process_group()
'''


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_count_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script), ROOT]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    g = golden("post_process.npz")
    assert res["total"] == g["tally_pred_counts"].tolist() == res["total_t"]
    assert res["max"] == 2.0 and res["n"] == [0, 500]
    rows = np.asarray(res["rows"])
    assert rows.shape == (2, 3) and rows[0].tolist() == res["mine"] and rows.sum(0).tolist() == res["total"]


# ---- POD5 / BAM ingest on the reference's own test data (next row N1) ------------------------------
DATA = os.path.join(ROOT, "tests", "golden", "data")


def test_bam_and_pod5_parsers_on_reference_test_data():
    from oracle import oracle as O
    from remora_amd import io as rio

    recs = list(rio.iter_bam_records(os.path.join(DATA, "can_mappings.bam")))
    from golden_util import pod5_reads_cpu

    pods = {p.read_id: p for p in pod5_reads_cpu(os.path.join(DATA, "can_reads.pod5"))}
    g = golden("real_reads_can.npz")
    assert len(recs) == len(pods) == int(g["num_records"]) == 14
    assert sum(r.is_reverse for r in recs) == 4 and all(r.reference_name == "chr13" for r in recs)
    for i, rec in enumerate(recs):
        assert rec.query_name == str(g[f"r{i}_name"]) and rec.flag == int(g[f"r{i}_flag"])
        tags = dict(rec.tags)
        assert {"mv", "ts", "ns", "sm", "sd"} <= set(tags) and set(rec.query_sequence) <= set("ACGTN")
        sig = pods[rec.query_name].signal
        assert sig.dtype == np.int16 and 300 < sig.mean() < 1500
        # move table, trimmed signal and basecalls must be mutually concordant (io.py:403-406)
        n_sig = sig[tags["ts"] : tags["ns"]].size
        q2s, mv, stride = O.parse_move_tag(np.asarray(tags["mv"], np.int8), n_sig, seq_len=len(rec.query_sequence))
        assert q2s[-1] == n_sig and q2s.size == len(rec.query_sequence) + 1
        # the reference's into_remora_read trims to [q2s[0], q2s[-1]) and re-bases the map
        assert np.array_equal(q2s - q2s[0], g[f"r{i}_map"]) and n_sig - q2s[0] == int(g[f"r{i}_ndacs"])
        seq = rio.revcomp(rec.query_sequence) if rec.is_reverse else rec.query_sequence
        assert seq == str(g[f"r{i}_seq"])
    assert rio.revcomp("ACGTN") == "NACGT"


@pytest.mark.parametrize("name", ["can", "mod"])
def test_parsers_agree_with_numbers_other_programs_wrote(name):
    """pysam and pod5 are not in the image, so the BAM / POD5 readers are checked against each other's second implementation
    elsewhere (native vs Python reader, GPU vs numpy VBZ decoder).  What IS independent of this repository are numbers that the
    basecaller (dorado), MinKNOW and the aligner wrote into the two files about the same read.  A reader that mis-parses a
    field, a tag type, the 4-bit sequence, the qualities, the VBZ rows or the calibration breaks one of these:
      ns tag (BAM)  ==  num_samples (POD5 reads table)  ==  sum of the signal rows' `samples`  ==  decoded signal length
      du tag (seconds) x 4 kHz  ==  ns                      mv: stride 5, one entry per stride of [ts, ns), its ones == bases
      qs tag  ==  rounded mean-error q-score of the decoded quality string
      sm / sd tags (pA shift / scale)  ~  median / MAD of the decoded, calibrated signal      NM tag: test_reference_sequence_from_md..."""
    from golden_util import pod5_reads_cpu
    from remora_amd import io as rio

    pod5_path = os.path.join(DATA, f"{name}_reads.pod5")
    f = rio.Pod5File(pod5_path)
    num_samples = dict(zip(f.read_ids, f._reads.column("num_samples").to_pylist()))
    decoded = {p.read_id: p for p in pod5_reads_cpu(pod5_path)}
    n = 0
    for rec in rio.iter_bam_records(os.path.join(DATA, f"{name}_mappings.bam")):
        t, p = dict(rec.tags), decoded[rec.query_name]
        rows = f.signal_rows(rec.query_name)
        assert t["ns"] == num_samples[rec.query_name] == sum(k for _, k in rows) == p.signal.size
        assert abs(t["du"] * 4000.0 - t["ns"]) < 0.5
        mv = np.asarray(t["mv"])
        assert int(mv[0]) == 5 and mv.size - 1 == (t["ns"] - t["ts"]) // 5 and set(mv[1:].tolist()) <= {0, 1}
        assert int(mv[1:].sum()) == len(rec.query_sequence) == len(rec.query_qualities)
        q = np.frombuffer(bytes(rec.query_qualities), dtype=np.uint8).astype(np.float64)
        assert abs(-10.0 * np.log10(np.mean(10.0 ** (-q / 10.0))) - t["qs"]) < 0.75
        off, scale = f.calibration(rec.query_name)
        pa = (p.signal[t["ts"] :].astype(np.float64) + off) * scale
        med = np.median(pa)
        # (dorado's shift / scale are quantile based: the same quantities to a quarter / a fifth of the scale)
        assert abs(t["sm"] - med) < 0.25 * t["sd"] and abs(t["sd"] / (1.4826 * np.median(np.abs(pa - med))) - 1.0) < 0.2
        n += 1
    assert n == len(f.read_ids) >= 10


def test_bam_writer_roundtrip(tmp_path):
    from remora_amd import io as rio

    src = os.path.join(DATA, "can_mappings.bam")
    recs = list(rio.iter_bam_records(src))
    out = str(tmp_path / "rt.bam")
    with rio.BamWriter(out, rio.read_bam_header_bytes(src)) as w:
        for i, r in enumerate(recs):
            w.write(rio.record_with_mod_tags(r, "C+m?,1,2;" if i % 2 == 0 else None, [10, 200] if i % 2 == 0 else None))
    back = list(rio.iter_bam_records(out))
    assert len(back) == len(recs)
    for i, (b, r) in enumerate(zip(back, recs)):
        assert (b.query_name, b.flag, b.reference_start, b.cigartuples, b.query_sequence) == \
               (r.query_name, r.flag, r.reference_start, r.cigartuples, r.query_sequence)
        got = [(k, list(v) if not isinstance(v, (str, int, float)) else v) for k, v in b.tags]
        want = [(k, list(v) if not isinstance(v, (str, int, float)) else v) for k, v in r.tags]
        if i % 2 == 0:
            want += [("MM", "C+m?,1,2;"), ("ML", [10, 200])]
        assert got == want
    # re-tagging an already tagged record replaces, not duplicates
    again = rio.record_with_mod_tags(back[0], "C+m?,5;", [7])
    with rio.BamWriter(out, rio.read_bam_header_bytes(src)) as w:
        w.write(again)
    (only,) = list(rio.iter_bam_records(out))
    assert [k for k, _ in only.tags].count("MM") == 1 and only.get_tag("MM") == "C+m?,5;" and list(only.get_tag("ML")) == [7]


# ---- N2 host side: SigMapRefiner without a GPU ---------------------------------------------------
def test_sig_map_refiner_table_file_matches_reference():
    """load_kmer_table / determine_dominant_pos / fix_gauge against values the reference produced
    for the same table file (tools/gen_golden.py gen_refine)."""
    from remora_amd.refine_signal_map import SigMapRefiner

    g = golden("refine_signal_map.npz")
    path = os.path.join(ROOT, "tests", "golden", "data", "levels_4mer.txt")
    for tag, fix in (("raw", False), ("fix", True)):
        ref = SigMapRefiner(kmer_model_filename=path, do_rough_rescale=True, do_fix_guage=fix)
        assert ref.is_loaded and ref.is_valid and ref.kmer_len == 4
        assert int(ref.center_idx) == int(g[f"table_{tag}_center"])
        np.testing.assert_allclose(ref.kmer_idx_stats, g[f"table_{tag}_stats"], rtol=1e-12)
        np.testing.assert_array_equal(np.asarray(ref.levels_array, np.float64), g[f"table_{tag}_levels"])
        assert ref.bases_before == 2 and ref.bases_after == 1
    assert "4-mer table" in repr(ref)


def test_sig_map_refiner_host_rescale_matches_reference():
    """Settings without a DP pass (scale_iters = -1) are pure host arithmetic: shift/scale equal the
    reference's for both rough re-scale methods; extract_levels equals the Cython function."""
    from remora_amd.data_chunks import RemoraRead
    from remora_amd.refine_signal_map import SigMapRefiner

    g = golden("refine_signal_map.npz")
    settings = json.loads(str(g["settings_json"]))
    si = [i for i, st in enumerate(settings) if st["scale_iters"] < 0][0]
    ref = SigMapRefiner(_levels_array=g["kmer_levels"], center_idx=int(g["center_idx"]), **settings[si])
    for n in "abcd":
        np.testing.assert_array_equal(ref.extract_levels(g[f"{n}_int_seq"]), g[f"{n}_levels"])
        read = RemoraRead(dacs=g[f"{n}_dacs"], shift=505.0, scale=83.0, seq_to_sig_map=g[f"{n}_map"].copy(),
                          int_seq=g[f"{n}_int_seq"], read_id=n)
        read.refine_signal_mapping(ref)
        np.testing.assert_array_equal(read.seq_to_sig_map, g[f"s{si}_{n}_map"])
        np.testing.assert_allclose([read.shift, read.scale], g[f"s{si}_{n}_shift_scale"], rtol=1e-12)
    # least squares flavour against the oracle restatement of the reference
    ref2 = SigMapRefiner(_levels_array=g["kmer_levels"], center_idx=int(g["center_idx"]), do_rough_rescale=True)
    from oracle import oracle as O

    want = O.refiner_rough_rescale(dict(levels=g["kmer_levels"], center_idx=int(g["center_idx"]),
                                        rough_rescale_method="least_squares"), 505.0, 83.0, g["b_map"], g["b_int_seq"],
                                   g["b_dacs"])
    got = ref2.rough_rescale(505.0, 83.0, g["b_map"], g["b_int_seq"], g["b_dacs"])
    assert tuple(got) == tuple(want)


def test_fast_quantile_is_numpy_quantile():
    from remora_amd import refine_signal_map as R

    rng = np.random.default_rng(3)
    q = np.arange(0.05, 1, 0.05)
    for t in range(300):
        n = int(rng.integers(1, 3000))
        a = rng.normal(0, 1, n).astype(np.float32 if t % 2 else np.float64)
        if t % 3 == 0:
            a = np.round(a, 1)  # ties
        got, want = R._quantile(a, q), np.quantile(a, q)
        assert got.dtype == want.dtype
        np.testing.assert_array_equal(got, want)
    assert R._QUANTILE_OK is True  # the replica is in use with this numpy


def test_sig_map_refiner_metadata_roundtrip_and_errors():
    from remora_amd import RemoraError
    from remora_amd.refine_signal_map import SigMapRefiner, compute_dwell_pen_array, index_from_kmer

    g = golden("refine_signal_map.npz")
    np.testing.assert_array_equal(compute_dwell_pen_array(4, 3, 0.5), g["sd_arr"])
    assert index_from_kmer("CAAAAAAAA") == 65536 and index_from_kmer("AAA") == 0
    a = SigMapRefiner(_levels_array=g["kmer_levels"], center_idx=2, do_rough_rescale=True, scale_iters=0)
    b = SigMapRefiner.load_from_metadata(a.asdict())
    assert a == b and a != SigMapRefiner()
    assert SigMapRefiner() == SigMapRefiner() and not SigMapRefiner().is_loaded
    table = {"".join(k): float(i) for i, k in enumerate(__import__("itertools").product("ACGT", repeat=2))}
    c = SigMapRefiner.load_from_dict(table)
    assert c.kmer_len == 2 and c.levels_array[index_from_kmer("CA")] == 4.0
    with pytest.raises(RemoraError):
        SigMapRefiner(scale_iters=0)  # refinement without a table
    with pytest.raises(RemoraError):
        SigMapRefiner(_levels_array=g["kmer_levels"], rough_rescale_method="nope", do_rough_rescale=True)


def test_refinement_fails_loudly_without_gpu():
    import torch
    from remora_amd import RemoraError
    from remora_amd.data_chunks import RemoraRead
    from remora_amd.refine_signal_map import SigMapRefiner

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = golden("refine_signal_map.npz")
    ref = SigMapRefiner(_levels_array=g["kmer_levels"], center_idx=2, scale_iters=0)
    read = RemoraRead(dacs=g["d_dacs"], shift=505.0, scale=83.0, seq_to_sig_map=g["d_map"].copy(),
                      int_seq=g["d_int_seq"], read_id="d")
    with pytest.raises(RemoraError, match="no GPU|GPU"):
        read.refine_signal_mapping(ref)


# ---- B1 / N3: batching bookkeeping of `remora infer` ---------------------------------------------
def _batching_inputs(g):
    from types import SimpleNamespace

    from remora_amd.inference import PackedKmers

    kcb = tuple(int(x) for x in g["kmer_context_bases"])
    reads, prepped = {}, []
    for ri in range(6):
        io_read = SimpleNamespace(read_id=f"read{ri}")
        reads[io_read.read_id] = io_read
        if f"r{ri}_err" in g:
            prepped.append([(io_read, None, str(g[f"r{ri}_err"]))])
            continue
        chunks = {}
        for cb in "CA":
            chunks[cb] = {"signal": g[f"r{ri}_{cb}_signal"],
                          "kmers": PackedKmers(g[f"r{ri}_{cb}_sequence"], g[f"r{ri}_{cb}_mapping"], g[f"r{ri}_{cb}_lengths"], kcb),
                          "read_focus_bases": g[f"r{ri}_{cb}_rfb"]}
        prepped.append([(io_read, chunks, None)])
    prepped.append([])
    L = int(g["chunk_len"])
    mds = [dict(can_base="C", chunk_len=L, kmer_len=sum(kcb) + 1), dict(can_base="A", chunk_len=L, kmer_len=sum(kcb) + 1)]
    return prepped, mds, kcb


def test_batch_reads_and_unbatch_match_reference():
    """Same batches (signals, read positions, [read, b_st, b_en, err] spans, k-mer content) and the same per-read
    re-assembly as the reference's batch_reads / unbatch on reads that straddle batches (tools/gen_golden.py
    gen_batching): 6 reads incl. an error read, 2 canonical-base models, batch size 4."""
    import queue
    from types import SimpleNamespace

    from oracle import oracle as O
    from remora_amd.inference import batch_reads, unbatch

    g = golden("batching.npz")
    prepped, mds, kcb = _batching_inputs(g)
    bq = queue.Queue()
    batch_reads(iter(prepped), bq, int(g["batch_size"]), mds)
    batches = []
    while True:
        it = bq.get()
        if it is StopIteration:
            break
        batches.append(it)
    assert len(batches) == int(g["num_batches"])
    called = queue.Queue()
    for bi, (cb, b_sigs, b_kmers, b_pos, b_reads) in enumerate(batches):
        assert cb == str(g[f"b{bi}_can_base"])
        np.testing.assert_array_equal(b_sigs, g[f"b{bi}_sigs"])
        np.testing.assert_array_equal(b_pos, g[f"b{bi}_pos"])
        assert json.dumps([[r.read_id, st, en, err] for r, st, en, err in b_reads]) == str(g[f"b{bi}_spans"])
        enc = O.compute_encoded_kmer_batch(kcb[0], kcb[1], b_kmers.sequence, b_kmers.mapping, b_kmers.lengths)
        np.testing.assert_array_equal(enc.astype(np.uint8), g[f"b{bi}_enc"])
        nn_out = g[f"b{bi}_nn_out"]
        called.put((cb, SimpleNamespace(cpu=lambda a=nn_out: SimpleNamespace(numpy=lambda a=a: a)), b_pos, b_reads))
    called.put(StopIteration)
    rq = queue.Queue()
    unbatch(called, rq, mds)
    done = []
    while True:
        it = rq.get()
        if it is StopIteration:
            break
        done.append(it)
    assert len(done) == int(g["num_done"])
    for di, (io_read, mod_calls, err) in enumerate(done):
        assert io_read.read_id == str(g[f"d{di}_read_id"])
        assert (err or "") == str(g[f"d{di}_err"])
        assert [cb for cb, _, _ in mod_calls] == [str(x) for x in g[f"d{di}_bases"]]
        for cb, nn_out, pos in mod_calls:
            np.testing.assert_array_equal(nn_out, g[f"d{di}_{cb}_nn_out"])
            np.testing.assert_array_equal(pos, g[f"d{di}_{cb}_pos"])


def test_prep_nn_input_and_unbatch_errors():
    from types import SimpleNamespace

    from remora_amd import RemoraError
    from remora_amd.inference import prep_nn_input, unbatch_reads

    assert prep_nn_input([]) == [(None, None, "No valid mappings")]
    r = SimpleNamespace(read_id="a")
    assert prep_nn_input([(r, None, "boom")]) == [(r, None, "boom")]
    with pytest.raises(RemoraError, match="None read"):
        unbatch_reads(None, np.zeros((2, 2)), np.zeros(2), [[r, None, 1, None]])
    other = (SimpleNamespace(read_id="b"), np.zeros((1, 2)), np.zeros(1), None)
    with pytest.raises(RemoraError, match="mismatching"):
        unbatch_reads(other, np.zeros((2, 2)), np.zeros(2), [[r, None, 1, None]])


# ---- N1: reference-anchored reads (host part) ----------------------------------------------------
def test_reference_sequence_from_md_is_pinned_three_ways():
    """BamRecord.get_reference_sequence (MD tag + CIGAR + query) on the reference's test BAM: (1) mismatches +
    inserted + deleted bases equal the NM tag of every record, (2) every CpG of the reference's ground-truth
    BED (tests/data/can_gt.bed) inside a read's span is C / G in the rebuilt sequence, (3) the strand-aware
    sequence and compute_ref_to_signal equal what the reference's Read.add_alignment derived from the same
    records (tools/gen_golden.py gen_real_reads)."""
    from remora_amd import io as rio
    from remora_amd.data_chunks import compute_ref_to_signal, make_sequence_coordinate_mapping

    data = os.path.join(ROOT, "tests", "golden", "data")
    g = golden("real_reads_can.npz")
    gt = [(ln.split()[0], int(ln.split()[1]), ln.split()[5]) for ln in open(os.path.join(data, "can_gt.bed"))]
    checked = 0
    for i, rec in enumerate(rio.iter_bam_records(os.path.join(data, "can_mappings.bam"))):
        ref = rec.get_reference_sequence()
        tags = dict(rec.tags)
        ins = sum(ln for op, ln in rec.cigartuples if op == 1)
        dele = sum(ln for op, ln in rec.cigartuples if op == 2)
        assert sum(c.islower() for c in ref) + ins + dele == tags["NM"]
        span = sum(ln for op, ln in rec.cigartuples if op in (0, 2, 3, 7, 8))
        assert len(ref) == span
        for ctg, pos, strand in gt:
            if ctg == rec.reference_name and rec.reference_start <= pos < rec.reference_start + span:
                assert ref[pos - rec.reference_start].upper() == ("C" if strand == "+" else "G")
                checked += 1
        seq = rio.revcomp(ref.upper()) if rec.is_reverse else ref.upper()
        assert seq == str(g[f"r{i}_ra_ref_seq"])
        cigar = rec.cigartuples[::-1] if rec.is_reverse else rec.cigartuples
        assert make_sequence_coordinate_mapping(cigar).size == len(seq) + 1
        # the golden stores maps relative to their first entry; an integer offset commutes with the floor
        r2s = compute_ref_to_signal(g[f"r{i}_map"], cigar)
        r2s_full = g[f"r{i}_ra_ref_to_signal"]
        np.testing.assert_array_equal(r2s - r2s[0], r2s_full - r2s_full[0])
    assert i == 13 and checked > 1000


def test_cigar_mapping_errors_and_ref_anchored_record_bytes():
    from remora_amd import RemoraError
    from remora_amd import io as rio
    from remora_amd.data_chunks import make_sequence_coordinate_mapping

    with pytest.raises(RemoraError, match="No match"):
        make_sequence_coordinate_mapping([(4, 10), (1, 3)])
    with pytest.raises(RemoraError, match="Invalid cigar"):
        make_sequence_coordinate_mapping([(0, 5), (9, 1), (0, 2)])
    knots = make_sequence_coordinate_mapping([(4, 2), (0, 3), (1, 2), (0, 2), (2, 1), (0, 1), (4, 7)])
    np.testing.assert_allclose(knots, [2, 3, 4, 7, 8, 8.5, 9, 10])
    # a reference-anchored output record parses back as <len>M + the reference sequence, tags preserved
    data = os.path.join(ROOT, "tests", "golden", "data")
    rec = next(iter(rio.iter_bam_records(os.path.join(data, "can_mappings.bam"))))
    fwd = rec.get_reference_sequence().upper()
    raw = rio.record_with_mod_tags(rec, "C+m?,1,2;", [7, 200], ref_anchored_seq=fwd)
    hdr = rio.read_bam_header_bytes(os.path.join(data, "can_mappings.bam"))
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "o.bam")
        with rio.BamWriter(path, hdr) as w:
            w.write(raw)
        back = next(iter(rio.iter_bam_records(path)))
    assert back.cigartuples == [(0, len(fwd))] and back.query_sequence == fwd
    assert back.reference_start == rec.reference_start and back.flag == rec.flag and back.query_name == rec.query_name
    t = dict(back.tags)
    assert t["MM"] == "C+m?,1,2;" and list(t["ML"]) == [7, 200] and t["NM"] == dict(rec.tags)["NM"]


# ---- N4: dataset writer on the host (no GPU needed for write_batch / write_chunk / shuffle) -------
def test_core_dataset_write_batch_roundtrip_and_shuffle(tmp_path):
    """Rows read from the reference-written dataset, written back through write_batch / write_chunk, give the
    same metadata.jsn text and the same valid bytes; shuffle applies one permutation to every array."""
    from remora_amd import RemoraError
    from remora_amd.data_chunks import Chunk, CoreRemoraDataset, dataset_metadata

    ref_dir = os.path.join(ROOT, "tests", "golden", "data", "core_dataset")
    src = CoreRemoraDataset(ref_dir)
    n = src.size
    md = dataset_metadata(allocate_size=n + 7, max_seq_len=20, mod_bases=["m"], mod_long_names=["5mC"],
                          motif_sequences=["CG"], motif_offsets=[0], chunk_context=(50, 50), kmer_context_bases=(4, 4))
    out = str(tmp_path / "w")
    ds = CoreRemoraDataset(out, mode="w", metadata=md)
    rows = {k: np.array(v[:n]) for k, v in src.arrays.items()}
    half = n // 2
    ds.write_batch({k: v[:half] for k, v in rows.items()})
    for i in range(half, n):  # the rest chunk by chunk, as RemoraRead.iter_chunks would hand them over
        sl = int(rows["sequence_lengths"][i])
        ds.write_chunk(Chunk(signal=rows["signal"][i, 0], seq_w_context=rows["sequence"][i, : sl + 8],
                             seq_to_sig_map=rows["sequence_to_signal_mapping"][i, : sl + 1], kmer_context_bases=(4, 4),
                             chunk_sig_focus_idx=50, chunk_focus_base=0, read_focus_base=0, read_id="x",
                             label=int(rows["labels"][i])))
    ds.flush()
    assert open(os.path.join(out, "metadata.jsn")).read() == open(os.path.join(ref_dir, "metadata.jsn")).read()
    back = CoreRemoraDataset(out)
    for k in ("signal", "sequence_lengths", "labels"):
        np.testing.assert_array_equal(back.arrays[k][:n], rows[k])
    for i in range(n):
        sl = int(rows["sequence_lengths"][i])
        np.testing.assert_array_equal(back.arrays["sequence"][i, : sl + 8], rows["sequence"][i, : sl + 8])
        np.testing.assert_array_equal(back.arrays["sequence_to_signal_mapping"][i, : sl + 1],
                                      rows["sequence_to_signal_mapping"][i, : sl + 1])
    with pytest.raises(RemoraError, match="allocated"):
        ds.write_batch({k: v[:20] for k, v in rows.items()})
    with pytest.raises(RemoraError, match="Missing"):
        ds.write_batch({"signal": rows["signal"][:1]})
    with pytest.raises(RemoraError, match="mode"):
        back.write_batch({k: v[:1] for k, v in rows.items()})
    np.random.seed(5)
    ds.shuffle(batch_size=37)
    sig_after = np.array(ds.arrays["signal"][:n])
    order = [int(np.flatnonzero((rows["signal"] == sig_after[i]).all(axis=(1, 2)))[0]) for i in range(n)]
    assert sorted(order) == list(range(n)) and order != list(range(n))
    np.testing.assert_array_equal(np.array(ds.arrays["labels"][:n]), rows["labels"][order])
    np.testing.assert_array_equal(np.array(ds.arrays["sequence_lengths"][:n]), rows["sequence_lengths"][order])


def test_batch_reads_widens_for_reads_with_longer_chunks():
    """Reads whose widest chunk differs: the batch arrays grow and pad (sequence -1, mapping 0) without touching
    rows already packed."""
    import queue
    from types import SimpleNamespace

    from remora_amd.inference import PackedKmers, batch_reads

    def chunks(n, w, fill):
        return {"C": {"signal": np.full((n, 1, 10), fill, np.float32),
                      "kmers": PackedKmers(np.full((n, w + 2), fill, np.int8), np.full((n, w + 1), fill, np.int16),
                                           np.full(n, w, np.int16), (1, 1)),
                      "read_focus_bases": np.arange(n) + 100 * fill}}

    a, b = SimpleNamespace(read_id="a"), SimpleNamespace(read_id="b")
    q = queue.Queue()
    batch_reads(iter([[(a, chunks(3, 4, 1), None)], [(b, chunks(3, 9, 2), None)]]), q, 4,
                [dict(can_base="C", chunk_len=10, kmer_len=3)])
    first, second = q.get(), q.get()
    assert q.get() is StopIteration
    _, sig, km, pos, spans = first
    assert sig.shape == (4, 1, 10) and km.sequence.shape == (4, 11) and km.mapping.shape == (4, 10)
    assert (km.sequence[:3, :6] == 1).all() and (km.sequence[:3, 6:] == -1).all() and (km.mapping[:3, 5:] == 0).all()
    assert (km.sequence[3] == 2).all() and list(km.lengths) == [4, 4, 4, 9]
    assert [[r.read_id, s, e] for r, s, e, _ in spans] == [["a", 0, 3], ["b", 3, None]]
    assert [[r.read_id, s, e] for r, s, e, _ in second[4]] == [["b", None, 2]] and len(second[2]) == 2


def test_io_edge_cases_host():
    """MD-less records, soft clips / insertions / deletions in get_reference_sequence, truncated BAM, POD5 index."""
    import gzip
    import tempfile

    from remora_amd import RemoraError
    from remora_amd import io as rio

    rec = rio.BamRecord("q", 0, 0, "chr", 10, 60, [(4, 2), (0, 3), (1, 2), (0, 2), (2, 3), (0, 1), (4, 1)], "TTACGGGTCAG"[:11],
                        b"", [("MD", "2T2^ACG1")])
    # columns: M3 = ACG ; I2 skipped (GG) ; M2 = TC ; D3 ; M1 = A  -> MD: 2 match, T mismatch, 2 match, ^ACG deletion, 1 match
    assert rec.get_reference_sequence() == "ACtTCACGA"
    assert rio.BamRecord("q", 0, 0, "chr", 0, 0, [(0, 3)], "ACG", b"", []).is_unmapped is False
    with pytest.raises(ValueError):
        rio.BamRecord("q", 0, 0, "chr", 0, 0, [(0, 3)], "ACG", b"", []).get_reference_sequence()
    with pytest.raises(ValueError):
        rio.BamRecord("q", 0, 0, "chr", 0, 0, [(0, 3)], "ACG", b"", [("MD", "5")]).get_reference_sequence()
    assert rio.revcomp("ACGTN") == "NACGT"
    data = os.path.join(ROOT, "tests", "golden", "data")
    with tempfile.TemporaryDirectory() as td:
        raw = gzip.open(os.path.join(data, "can_mappings.bam"), "rb").read()
        bad = os.path.join(td, "trunc.bam")
        with gzip.open(bad, "wb") as fh:
            fh.write(raw[: len(raw) // 2])
        with pytest.raises(RemoraError, match="truncated"):
            list(rio.iter_bam_records(bad))
        notbam = os.path.join(td, "x.bam")
        with gzip.open(notbam, "wb") as fh:
            fh.write(b"nope")
        with pytest.raises(RemoraError, match="not a BAM"):
            list(rio.iter_bam_records(notbam))
    f = rio.Pod5File(os.path.join(data, "can_reads.pod5"))
    assert len(f) == len(f.read_ids) == 14 and f.read_ids[0] in f and "nope" not in f
    from golden_util import pod5_reads_cpu

    r = pod5_reads_cpu(os.path.join(data, "can_reads.pod5"), read_ids=f.read_ids[3:4])[0]
    assert r.signal.dtype == np.int16 and r.signal.size > 1000 and r.calibration_scale > 0
    assert sum(n for _, n in f.signal_rows(f.read_ids[3])) == r.signal.size and f.calibration(f.read_ids[3])[1] == r.calibration_scale
    with pytest.raises(RemoraError, match="no GPU|GPU"):
        f.get(f.read_ids[3])  # decoding a read is a GPU call: no CPU fallback in the product
    assert rio._pack_seq("ACGTN") == bytes([0x12, 0x48, 0xF0])


# ---- N4: RemoraDataset mixing, dataset configs, validation metrics (host logic; golden from the reference) ----
@pytest.fixture(scope="module")
def O():
    from oracle import oracle

    return oracle


def _materialise(tmp_path, names):
    from golden_util import materialise_dataset

    g = golden("prepared_datasets.npz")
    return {n: materialise_dataset(g, n, str(tmp_path / n)) for n in names}


def _batches(it, O, kcb, limit=None):
    """Concatenated (enc code, signal, labels, batch sizes) of raw-row batches; the one-hot code through the oracle."""
    codes, sigs, labs, sizes = [], [], [], []
    for bi, (sig, seq, mp, ln, lab) in enumerate(it):
        enc = O.compute_encoded_kmer_batch(*kcb, seq.numpy(), mp.numpy(), ln.numpy())
        n, K4, L = enc.shape
        e4 = enc.reshape(n, K4 // 4, 4, L)
        codes.append(np.where(e4.sum(2) > 0, e4.argmax(2), -1).astype(np.int8))
        sigs.append(sig.numpy()); labs.append(lab.numpy()); sizes.append(n)
        if limit is not None and bi + 1 >= limit:
            break
    return np.concatenate(codes), np.concatenate(sigs), np.concatenate(labs), np.asarray(sizes)


RAW = ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths", "labels")


def _check_batches(g, tag, got):
    code, sig, lab, sizes = got
    np.testing.assert_array_equal(sizes, g[f"{tag}_bsizes"])
    np.testing.assert_array_equal(lab, g[f"{tag}_labels"])
    assert np.array_equal(sig.view(np.uint32), g[f"{tag}_signal"].view(np.uint32))
    np.testing.assert_array_equal(code, g[f"{tag}_enc_code"])


def test_remora_dataset_two_way_mix_matches_reference(tmp_path, O):
    """Two prepared datasets at 0.5 / 0.5 from a config file (the reference's own test fixture shape): batch split,
    merged metadata, label counts, every batch of the finite iteration, head() and train_test_split()."""
    from remora_amd.data_chunks import CoreRemoraDataset, RemoraDataset

    g = golden("remora_dataset.npz")
    dirs = _materialise(tmp_path, ["can_ctrl", "mod_m"])
    for n, p in dirs.items():
        assert CoreRemoraDataset.hash(p) == str(g[f"hash_{n}"]) and CoreRemoraDataset.check_dataset_dir(p)
    cfg = str(tmp_path / "mix1.cfg")
    json.dump([[dirs["can_ctrl"], 0.5], [dirs["mod_m"], 0.5]], open(cfg, "w"))
    ds = RemoraDataset.from_config(cfg, ds_kwargs={"infinite_iter": False}, batch_size=32, return_arrays=RAW)
    np.testing.assert_array_equal(ds.batch_sizes, g["mix1_batch_sizes"])
    np.testing.assert_allclose(ds.props, g["mix1_props"])
    np.testing.assert_array_equal(ds.get_label_counts(), g["mix1_label_counts"])
    assert ds.label_summary == str(g["mix1_label_summary"])
    mods, longs, motifs = json.loads(str(g["mix1_mod_bases"]))
    assert ds.metadata.mod_bases == mods and ds.metadata.mod_long_names == longs
    assert [list(m) for m in ds.metadata.motifs] == motifs
    assert ds.epoch_summary(10).replace(str(tmp_path), "<TD>") == str(g["mix1_epoch_summary"])
    kcb = ds.metadata.kmer_context_bases
    _check_batches(g, "mix1", _batches(iter(ds), O, kcb))
    _check_batches(g, "mix1", _batches(iter(ds), O, kcb))  # a finite dataset restarts on every iter()
    hd = ds.head(40)
    np.testing.assert_array_equal([s.size for s in hd.datasets], g["mix1_head_sizes"])
    _check_batches(g, "mix1_head", _batches(iter(hd), O, kcb))
    trn, tst = ds.train_test_split(25)
    np.testing.assert_array_equal([[s.size for s in trn.datasets], [s.size for s in tst.datasets]], g["mix1_split_sizes"])
    _check_batches(g, "mix1_test", _batches(iter(tst), O, kcb))
    _check_batches(g, "mix1_train", _batches(iter(trn), O, kcb, limit=5), )
    ds.load_all_batches()
    np.testing.assert_array_equal(ds.get_label_counts(), np.bincount(g["mix1_labels"], minlength=2))
    _check_batches(g, "mix1", _batches(iter(ds), O, kcb))


def test_remora_dataset_nested_config_label_conversion_and_wraparound(tmp_path, O):
    """Three datasets with two different modified bases through a nested config: proportions, hashes, the
    label conversion table of every dataset, merged (sorted) mod bases, and the first 9 batches of the
    infinite iteration with super batches smaller than the datasets (wrap-around)."""
    from remora_amd import RemoraError
    from remora_amd.data_chunks import CoreRemoraDataset, RemoraDataset, load_dataset, parse_dataset_config

    g = golden("remora_dataset.npz")
    dirs = _materialise(tmp_path, ["can_ctrl", "mod_m", "mod_h"])
    sub, cfg = str(tmp_path / "sub.cfg"), str(tmp_path / "mix2.cfg")
    json.dump([[dirs["mod_m"], 3], [dirs["mod_h"], 1]], open(sub, "w"))
    json.dump([[dirs["can_ctrl"], 2], [sub, 3]], open(cfg, "w"))
    paths, props, hashes = parse_dataset_config(cfg)
    assert [os.path.basename(p) for p in paths] == list(g["mix2_paths"])
    np.testing.assert_allclose(props, g["mix2_props"], rtol=0, atol=1e-15)
    assert hashes == list(g["mix2_hashes"])
    assert load_dataset(dirs["mod_m"])[0] == [dirs["mod_m"]] and load_dataset(cfg)[0] == paths
    ds = RemoraDataset([CoreRemoraDataset(p) for p in paths], props, hashes, batch_size=50, super_batch_size=70,
                       return_arrays=RAW)
    np.testing.assert_array_equal(ds.batch_sizes, g["mix2_batch_sizes"])
    assert [ds.metadata.mod_bases, ds.metadata.mod_long_names] == json.loads(str(g["mix2_mod_bases"]))
    want_conv = json.loads(str(g["mix2_label_conv"]))
    assert [None if s.label_conv is None else s.label_conv.tolist() for s in ds.datasets] == want_conv
    np.testing.assert_array_equal(ds.get_label_counts(), g["mix2_label_counts"])
    assert [[os.path.basename(c[0])] + list(c[1:]) for c in ds.get_config()] == json.loads(str(g["mix2_config_json"]))
    _check_batches(g, "mix2", _batches(iter(ds), O, ds.metadata.kmer_context_bases, limit=9))
    with pytest.raises(RemoraError, match="infinite"):
        ds.load_all_batches()
    # config errors
    bad = str(tmp_path / "bad.cfg")
    json.dump([[dirs["can_ctrl"], 1, "0" * 64]], open(bad, "w"))
    with pytest.raises(RemoraError, match="hash"):
        parse_dataset_config(bad)
    json.dump([[bad, 1]], open(bad, "w"))
    with pytest.raises(RemoraError, match="Circular"):
        parse_dataset_config(bad)
    json.dump([[str(tmp_path / "nope"), 1]], open(bad, "w"))
    with pytest.raises(RemoraError, match="does not exist"):
        parse_dataset_config(bad)
    with pytest.raises(RemoraError, match="proportions"):
        RemoraDataset([CoreRemoraDataset(paths[0])], [1.5])
    with pytest.raises(RemoraError, match="same length"):
        RemoraDataset([CoreRemoraDataset(paths[0])], [0.5, 0.5])


def test_core_dataset_override_errors(tmp_path):
    """Error behaviour of the metadata overrides (src/remora/data_chunks.py:1078-1216) and of mixing datasets
    whose extraction settings differ (:1880-1896)."""
    from remora_amd import RemoraError
    from remora_amd.data_chunks import CoreRemoraDataset, RemoraDataset

    dirs = _materialise(tmp_path, ["can_ctrl", "can_default"])
    path = dirs["can_ctrl"]
    ds = CoreRemoraDataset(path, override_metadata={"extra_arrays": {}, "kmer_context_bases": (2, 3)})
    assert ds.metadata.extra_array_names == [] and ds.metadata.stored_kmer_context_bases == (4, 4)
    assert ds.metadata.kmer_context_bases_adjusted and not ds.metadata.chunk_context_adjusted
    with pytest.raises(RemoraError, match="Cannot expand chunk context"):
        CoreRemoraDataset(path, override_metadata={"chunk_context": (60, 50)})
    with pytest.raises(RemoraError, match="Cannot expand kmer context"):
        CoreRemoraDataset(path, override_metadata={"kmer_context_bases": (5, 4)})
    with pytest.raises(RemoraError, match="Cannot change metadata values"):
        CoreRemoraDataset(path, override_metadata={"offset": 2})
    with pytest.raises(RemoraError, match="missing arrays"):
        CoreRemoraDataset(path, override_metadata={"extra_arrays": {"nope": 1}})
    with pytest.raises(RemoraError, match="empty"):
        CoreRemoraDataset(path, override_metadata={"dataset_start": 205})
    with pytest.raises(RemoraError, match="past loaded end"):
        CoreRemoraDataset(path, override_metadata={"dataset_end": 9999})
    with pytest.raises(RemoraError, match="must have same"):
        RemoraDataset([CoreRemoraDataset(path), CoreRemoraDataset(dirs["can_default"])], [0.5, 0.5])


def test_compute_best_split_and_metrics_match_reference():
    from remora_amd.data_chunks import compute_best_split
    from remora_amd.validate import add_unmodeled_labels, compute_metrics, confusion_matrix, mat_to_str

    g = golden("remora_dataset.npz")
    for fi in range(4):
        spec = g[f"split{fi}_in"]
        np.testing.assert_array_equal(compute_best_split(int(spec[0]), spec[1:]), g[f"split{fi}"])
    probs, labels = g["cm_probs"], g["cm_labels"]
    for fi in range(3):
        frac, acc, ff, facc, thr = g[f"cm{fi}_scalars"]
        got = compute_metrics(probs, labels, frac)
        assert (got[0], got[2], got[3], got[5]) == (acc, ff, facc, thr)
        np.testing.assert_array_equal(got[1], g[f"cm{fi}_conf"])
        np.testing.assert_array_equal(got[4], g[f"cm{fi}_filt_conf"])
    np.testing.assert_array_equal(confusion_matrix([2, 2, 5], [5, 2, 5]), [[1, 1], [0, 1]])  # only classes that occur
    assert mat_to_str(np.array([[1, 2], [3, 4]])) == "[[1,2],[3,4]]"
    for tag, cols in (("1", [1]), ("2", [2]), ("13", [1, 3])):
        np.testing.assert_array_equal(add_unmodeled_labels(g["aul_in"], np.array(cols)), g[f"aul_out_{tag}"])
    assert add_unmodeled_labels(g["aul_in"], np.array([])) is g["aul_in"] or True


def test_check_super_batch_rejects_broken_rows(tmp_path):
    from remora_amd import RemoraError
    from remora_amd.data_chunks import CoreRemoraDataset, check_super_batch

    d = _materialise(tmp_path, ["can_ctrl"])["can_ctrl"]
    ds = CoreRemoraDataset(d, infinite_iter=False)
    sb = ds.load_super_batch(0, 50)
    check_super_batch(sb, 100)
    for name, row, col, val, msg in (("sequence_to_signal_mapping", 3, 1, -4, "negative"),
                                     ("sequence_to_signal_mapping", 3, 0, 120, "beyond"),
                                     ("sequence", 5, 2, 7, "less than 4"), ("sequence", 5, 2, -3, "greater")):
        bad = {k: v.copy() for k, v in sb.items()}
        bad[name][row, col] = val
        with pytest.raises(RemoraError, match=msg):
            check_super_batch(bad, 100)
    bad = {k: v.copy() for k, v in sb.items()}
    ln = int(bad["sequence_lengths"][7])
    bad["sequence_to_signal_mapping"][7, ln] = 99
    with pytest.raises(RemoraError, match="does not end"):
        check_super_batch(bad, 100)
    bad = {k: v.copy() for k, v in sb.items()}
    bad["sequence_to_signal_mapping"][7, 2] = bad["sequence_to_signal_mapping"][7, 1] - 1
    with pytest.raises(RemoraError, match="monotonic"):
        check_super_batch(bad, 100)


def test_map_ref_to_signal_known_answers_of_the_reference_tests():
    """The known-answer vectors the reference's own tests hold for data_chunks.map_ref_to_signal
    (tests/test_duplex.py:57-251; query_to_signal = arange(len(simplex)), knots = the duplex-to-simplex map):
    (simplex length, first knot, number of knots) -> expected signal coordinates."""
    from remora_amd.data_chunks import map_ref_to_signal

    cases = [
        (16, 5, 12, [5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 15]),    # extra simplex sequence at the 5' end
        (11, 0, 12, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10]),         # simplex missing sequence at the 5' end
        (11, 2, 10, [2, 3, 4, 5, 6, 7, 8, 9, 10, 10]),               # ... and starting with unpaired sequence
        (11, 0, 12, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10]),         # missing sequence at the 3' end
        (13, 0, 12, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]),         # ... with unaligned simplex sequence after it
        (22, 5, 12, [5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]),    # ragged ends, simplex longer
        (11, 0, 12, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10]),         # ragged ends, duplex longer
    ]
    for n, first, count, want in cases:
        got = map_ref_to_signal(query_to_signal=np.arange(n), ref_to_query_knots=np.arange(first, first + count))
        np.testing.assert_array_equal(got, want)


def test_dataset_cli_merge_head_copy(tmp_path):
    """`dataset merge`, `dataset head`, `dataset copy` (src/remora/parsers.py:493-727) on three prepared datasets
    with two different modified bases: the merged dataset holds exactly the union of the rows with the labels
    converted to the merged label set, `--max-size` caps it in proportion, head takes the first rows, copy
    reproduces directories + config (hashes verify)."""
    from golden_util import dataset_rows
    from remora_amd.__main__ import main
    from remora_amd.data_chunks import CoreRemoraDataset, RemoraDataset

    dirs = _materialise(tmp_path, ["can_ctrl", "mod_m", "mod_h"])
    cfg = str(tmp_path / "two.cfg")
    json.dump([[dirs["mod_m"], 1], [dirs["mod_h"], 1]], open(cfg, "w"))
    merged = str(tmp_path / "merged")
    np.random.seed(3)
    assert main(["dataset", "merge", merged, dirs["can_ctrl"], cfg]) == 0
    ds = CoreRemoraDataset(merged, infinite_iter=False)
    assert ds.size == 205 + 210 + 84 and ds.metadata.mod_bases == ["h", "m"] and ds.metadata.max_seq_len == 20
    np.testing.assert_array_equal(ds.get_label_counts(), [205, 84, 210])
    _, got = dataset_rows(merged)
    key = lambda r: sorted(zip(r["read_ids"].tolist(), r["read_focus_bases"].tolist(), r["labels"].tolist(),
                               r["sequence_lengths"].tolist(), [s.tobytes() for s in r["signal"]]))
    want = []
    for name, conv in (("can_ctrl", {0: 0}), ("mod_m", {1: 2}), ("mod_h", {1: 1})):
        _, rows = dataset_rows(dirs[name])
        rows["labels"] = np.array([conv[int(x)] for x in rows["labels"]])
        want += key(rows)
    assert key(got) == sorted(want)
    assert main(["dataset", "merge", merged, dirs["can_ctrl"], cfg]) == 1  # exists
    assert main(["dataset", "merge", merged, dirs["can_ctrl"], cfg, "--overwrite", "--max-size", "100"]) == 0
    small = CoreRemoraDataset(merged)
    assert small.size == 100 and list(small.get_label_counts()) == [41, 17, 42]
    head = str(tmp_path / "head")
    assert main(["dataset", "head", head, dirs["mod_m"], "37"]) == 0
    _, hrows = dataset_rows(head)
    _, src = dataset_rows(dirs["mod_m"])
    assert sorted(zip(hrows["read_ids"].tolist(), hrows["read_focus_bases"].tolist())) == \
        sorted(zip(src["read_ids"][:37].tolist(), src["read_focus_bases"][:37].tolist()))
    copied = str(tmp_path / "copied")
    three = str(tmp_path / "three.cfg")
    json.dump([[dirs["can_ctrl"], 2], [cfg, 3]], open(three, "w"))
    assert main(["dataset", "copy", three, copied]) == 0
    assert sorted(os.listdir(copied)) == ["dataset.cfg", "dataset_000", "dataset_001", "dataset_002", "sources.txt"]
    back = RemoraDataset.from_config(os.path.join(copied, "dataset.cfg"), batch_size=50)  # hashes are verified on load
    np.testing.assert_allclose(back.props, [0.4, 0.3, 0.3])
    assert back.size == 499 and list(back.get_label_counts()) == [205, 84, 210]


def test_batch_params_and_seeded_subsampling_match_reference(tmp_path):
    """adjust_batch_params over a grid of batch / super-batch sizes and sample fractions, and the batches of a
    seeded iteration with super_batch_sample_frac (np.random.choice inside every super batch), finite (including
    the reference's short and empty trailing batches) and infinite."""
    from remora_amd.data_chunks import CoreRemoraDataset

    g = golden("dataset_batch_params.npz")
    path = _materialise(tmp_path, ["can_ctrl"])["can_ctrl"]
    for bs, sbs, frac, cps, sel, new_bs, new_sbs in g["grid"]:
        ds = CoreRemoraDataset(path, batch_size=int(bs), super_batch_size=int(sbs),
                               super_batch_sample_frac=None if frac < 0 else float(frac), infinite_iter=False)
        got = ds.adjust_batch_params()
        assert (got[0], -1 if got[1] is None else got[1], ds.batch_size, ds.super_batch_size) == (cps, sel, new_bs, new_sbs), \
            (bs, sbs, frac)
    for tag, inf in (("finite", False), ("infinite", True)):
        ds = CoreRemoraDataset(path, batch_size=16, super_batch_size=64, super_batch_sample_frac=0.5, infinite_iter=inf)
        np.random.seed(5)
        sizes, fbs, ids = [], [], []
        for bi, b in enumerate(ds):
            sizes.append(b["labels"].size); fbs.append(b["read_focus_bases"]); ids.append(b["read_ids"])
            if bi >= 9:
                break
        np.testing.assert_array_equal(sizes, g[f"frac_{tag}_sizes"])
        np.testing.assert_array_equal(np.concatenate(fbs), g[f"frac_{tag}_read_focus_bases"])
        np.testing.assert_array_equal(np.concatenate(ids), g[f"frac_{tag}_read_ids"])


def test_zstd_rows_inflate_from_short_lived_threads():
    """POD5 signal rows are inflated in the ingest thread of every infer call: the zstd layer must survive being
    used from many short-lived threads (pyarrow's stream reader does not) and give the same bytes everywhere."""
    import threading

    from remora_amd import io as rio

    path = os.path.join(ROOT, "tests", "golden", "data", "can_reads.pod5")
    f0 = rio.Pod5File(path)
    rows = f0._sig.column("signal")
    ref = [rio._zstd_decompress(rows[i].as_py()) for i in range(f0._sig.num_rows)]
    assert len(ref) == 17 and all(len(r) > 1000 for r in ref)
    bad = []

    def work():
        f = rio.Pod5File(path)
        sr = f._sig.column("signal")
        for i in range(f._sig.num_rows):
            if rio._zstd_decompress(sr[i].as_py()) != ref[i]:
                bad.append(i)

    for _ in range(12):
        t = threading.Thread(target=work)
        t.start()
        t.join()
    assert not bad
    with pytest.raises(Exception):
        rio._zstd_decompress(b"\x28\xb5\x2f\xfd\x20\x10garbage")


def test_bam_writer_thread_pool_is_deterministic(tmp_path):
    """BGZF blocks are deflated by a thread pool and written in order: the file does not depend on the number of
    threads or on how many blocks may be in flight, and reads back record for record."""
    import struct

    from remora_amd import io as rio

    bam = os.path.join(ROOT, "tests", "golden", "data", "mod_mappings.bam")
    hdr = rio.read_bam_header_bytes(bam)
    recs = list(rio.iter_bam_records(bam))
    outs = []
    for k, (threads, pending) in enumerate(((1, 1), (4, 32), (3, 2))):
        path = str(tmp_path / f"w{k}.bam")
        with rio.BamWriter(path, hdr, threads=threads, max_pending=pending) as w:
            for _ in range(5):
                for r in recs:
                    w.write(struct.pack("<i", len(r.raw)) + r.raw)
        outs.append(open(path, "rb").read())
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 500_000
    back = list(rio.iter_bam_records(str(tmp_path / "w1.bam")))
    assert len(back) == 5 * len(recs) and all(b.raw == recs[i % len(recs)].raw for i, b in enumerate(back))


def test_native_bam_reader_equals_python_reader(tmp_path):
    """rmr_bam_read_batch (C++: parallel BGZF inflate, record split, hot tags, MD reconstruction) against the pure
    Python reader on the reference's BAM files and on a multi-megabyte file written by BamWriter: every field,
    the lazily decoded ones included, the hot tags, the rebuilt reference; batch sizes that split the file at
    awkward places; error behaviour on broken files."""
    import gzip
    import struct

    from remora_amd import RemoraError
    from remora_amd import io as rio

    data = os.path.join(ROOT, "tests", "golden", "data")
    big = str(tmp_path / "big.bam")
    src = os.path.join(data, "mod_mappings.bam")
    recs = list(rio.iter_bam_records(src, native=False))
    with rio.BamWriter(big, rio.read_bam_header_bytes(src)) as w:
        for _ in range(40):
            for r in recs:
                w.write(struct.pack("<i", len(r.raw)) + r.raw)
    for path, batch in ((os.path.join(data, "can_mappings.bam"), 512), (src, 5), (src, 1), (big, 97)):
        nat = list(rio.iter_bam_records(path, want_ref=True, batch=batch))
        py = list(rio.iter_bam_records(path, native=False))
        assert len(nat) == len(py) and len(py) in (14, 560)
        for x, y in zip(nat, py):
            for f in ("query_name", "flag", "reference_id", "reference_name", "reference_start", "mapping_quality",
                      "query_sequence", "raw", "tags_offset", "is_reverse", "is_unmapped"):
                assert getattr(x, f) == getattr(y, f), f
            assert x.get_reference_sequence() == y.get_reference_sequence()
            want = dict(y.tags)
            hot = x.hot_tags()
            assert set(hot) == {k for k in want if k in ("mv", "ts", "ns", "sp", "sm", "sd", "pi")}
            assert np.array_equal(hot["mv"], np.asarray(want["mv"], np.int8)) and hot["ts"] == want["ts"]
            assert abs(hot["sm"] - want["sm"]) < 1e-6 and abs(hot["sd"] - want["sd"]) < 1e-6
        for x, y in zip(nat[:14], py[:14]):  # lazily decoded members
            assert x.cigartuples == y.cigartuples and x.query_qualities == y.query_qualities
            assert x.tag_spans == y.tag_spans and x.to_dict() == y.to_dict() and x.get_tag("NM") == y.get_tag("NM")
            assert [(k, list(v) if hasattr(v, "typecode") else v) for k, v in x.tags] == \
                [(k, list(v) if hasattr(v, "typecode") else v) for k, v in y.tags]
    no_ref = next(rio.iter_bam_records(src, want_ref=False))
    assert no_ref.get_reference_sequence() == py[0].get_reference_sequence()  # falls back to the record's own MD tag
    raw = open(src, "rb").read()
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(RemoraError, match="truncated|BGZF"):
        list(rio.iter_bam_records(cut))
    flipped = bytearray(raw)
    flipped[40000] ^= 0x55
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(flipped))
    with pytest.raises(RemoraError, match="corrupt|BGZF|BAM"):
        list(rio.iter_bam_records(bad))
    notbam = str(tmp_path / "x.bam")
    with gzip.open(notbam, "wb") as fh:
        fh.write(b"nope")
    with pytest.raises(RemoraError, match="BGZF|not a BAM"):
        list(rio.iter_bam_records(notbam))
    with pytest.raises(RemoraError, match="cannot open"):
        list(rio.iter_bam_records(str(tmp_path / "missing.bam")))


def test_native_zstd_rows_equal_libzstd_one_shot():
    """rmr_zstd_frame_sizes / rmr_zstd_rows (native threads, one zstd context each) on the signal rows of the
    reference's POD5 files: the same bytes as the one-shot decompression, for 1 and 8 threads; corrupt frames and
    a wrong output span are refused."""
    import ctypes

    from remora_amd import RemoraError
    from remora_amd import _lib as L
    from remora_amd import io as rio

    blobs = []
    for which in ("can", "mod"):
        f = rio.Pod5File(os.path.join(ROOT, "tests", "golden", "data", f"{which}_reads.pod5"))
        rows = f._sig.column("signal")
        blobs += [rows[i].as_py() for i in range(f._sig.num_rows)]
    ref = [rio._zstd_decompress(b) for b in blobs]
    n = len(blobs)
    src = (ctypes.c_char_p * n)(*blobs)
    src_len = np.asarray([len(b) for b in blobs], np.int64)
    sizes = np.zeros(n, np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    L.check(L.lib().rmr_zstd_frame_sizes(src, p(src_len), n, p(sizes)))
    assert sizes.tolist() == [len(r) for r in ref]
    off = np.zeros(n + 1, np.int64)
    np.cumsum(sizes, out=off[1:])
    for threads in (1, 8):
        buf = np.zeros(int(off[-1]), np.uint8)
        L.check(L.lib().rmr_zstd_rows(src, p(src_len), n, p(buf), p(off), threads))
        assert all(bytes(buf[off[i] : off[i + 1]]) == ref[i] for i in range(n))
    bad = [bytearray(b) for b in blobs[:3]]
    bad[1][len(bad[1]) // 2] ^= 0xFF
    bsrc = (ctypes.c_char_p * 3)(*[bytes(b) for b in bad])
    with pytest.raises(RemoraError, match="corrupt zstd frame"):
        L.check(L.lib().rmr_zstd_rows(bsrc, p(src_len), 3, p(np.zeros(int(off[3]), np.uint8)), p(off), 2))
    short = off.copy()
    short[1:] -= 1
    with pytest.raises(RemoraError, match="corrupt zstd frame"):
        L.check(L.lib().rmr_zstd_rows(src, p(src_len), 2, p(np.zeros(int(off[2]), np.uint8)), p(short), 1))
    with pytest.raises(RemoraError, match="not a zstd frame"):
        junk = (ctypes.c_char_p * 1)(b"not zstd at all")
        L.check(L.lib().rmr_zstd_frame_sizes(junk, p(np.asarray([15], np.int64)), 1, p(np.zeros(1, np.int64))))


def test_native_bam_reader_fuzz_all_tag_types(tmp_path):
    """Random BAM records (every tag value type incl. all B sub-types, hot tags with different integer widths,
    unmapped / reverse / secondary flags, records that straddle BGZF blocks, a missing MD tag) written with
    BamWriter and read back by the native and the pure-Python reader: identical records."""
    import struct

    from remora_amd import io as rio

    rng = np.random.default_rng(17)
    text = b"@HD\tVN:1.6\n@SQ\tSN:chrA\tLN:100000\n@SQ\tSN:chrB\tLN:5000\n"
    hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", 2)
    for name, ln in ((b"chrA", 100000), (b"chrB", 5000)):
        hdr += struct.pack("<i", len(name) + 1) + name + b"\x00" + struct.pack("<i", ln)

    def tag(name, ty, val):
        out = name.encode() + ty.encode()
        if ty == "A":
            return out + val.encode()
        if ty in "cCsSiIf":
            return out + struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[ty], val)
        if ty in "ZH":
            return out + val.encode() + b"\x00"
        sub, arr = val
        fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
        return out + sub.encode() + struct.pack("<i", len(arr)) + struct.pack(f"<{len(arr)}{fmt}", *arr)

    records = []
    for k in range(150):
        n = int(rng.integers(1, 400)) if k % 10 else int(rng.integers(20000, 70000))  # some records larger than a block
        seq = "".join(rng.choice(list("ACGTN"), n))
        flag = int(rng.choice([0, 16, 4, 256, 2048, 16 + 2048]))
        ref_id = -1 if flag & 4 else int(rng.integers(0, 2))
        cigar = [] if flag & 4 else [(4, 1), (0, n - 2), (4, 1)] if n > 3 else [(0, n)]
        rname = f"read-{k:05d}".encode()
        body = struct.pack("<iiBBHHHiiii", ref_id, int(rng.integers(0, 4000)), len(rname) + 1, int(rng.integers(0, 61)), 4680,
                           len(cigar), flag, n, -1, -1, 0) + rname + b"\x00"
        body += b"".join(struct.pack("<I", (ln << 4) | op) for op, ln in cigar)
        body += rio._pack_seq(seq) + bytes(rng.integers(0, 42, n).astype(np.uint8))
        mv = [int(rng.integers(1, 13))] + rng.integers(0, 2, int(rng.integers(1, 3 * n + 2))).tolist()
        tags = [tag("NM", "i", int(rng.integers(0, 50))), tag("mv", "B", ("c", mv)),
                tag("ts", rng.choice(list("cCsSiI")), int(rng.integers(0, 100))),
                tag("ns", rng.choice(list("SiI")), int(rng.integers(1000, 60000))),
                tag("sm", "f", float(rng.normal(90, 5))), tag("sd", "f", float(rng.uniform(10, 30))),
                tag("XA", "A", "q"), tag("XZ", "Z", "free text, with spaces"), tag("XH", "H", "1AE301"),
                tag("Xc", "B", ("c", [-5, 6])), tag("XC", "B", ("C", [5, 250])), tag("Xs", "B", ("s", [-300, 7])),
                tag("XS", "B", ("S", [65535])), tag("Xi", "B", ("i", [-70000, 1])), tag("XI", "B", ("I", [4000000000])),
                tag("Xf", "B", ("f", [0.5, -2.25])), tag("Xe", "B", ("C", []))]
        if k % 3 == 0:
            tags.append(tag("sp", "i", int(rng.integers(0, 50))))
        if k % 4 == 0:
            tags.append(tag("pi", "Z", f"parent-{k}"))
        if not flag & 4 and k % 7:
            m = sum(ln for op, ln in cigar if op == 0)
            cut = int(rng.integers(0, m))
            tags.append(tag("MD", "Z", f"{cut}T{m - cut - 1}" if m > 1 else "1"))
        rng.shuffle(tags)
        records.append(body + b"".join(tags))
    path = str(tmp_path / "fuzz.bam")
    with rio.BamWriter(path, hdr, threads=2) as w:
        for r in records:
            w.write(struct.pack("<i", len(r)) + r)
    assert rio.read_bam_header_bytes(path) == hdr
    nat = list(rio.iter_bam_records(path, want_ref=True, batch=37))
    py = list(rio.iter_bam_records(path, native=False))
    assert len(nat) == len(py) == len(records)
    for x, y, raw in zip(nat, py, records):
        assert x.raw == y.raw == raw
        for f in ("query_name", "flag", "reference_id", "reference_name", "reference_start", "mapping_quality",
                  "query_sequence", "tags_offset", "cigartuples", "query_qualities", "tag_spans"):
            assert getattr(x, f) == getattr(y, f), f
        assert [(k, list(v) if hasattr(v, "typecode") else v) for k, v in x.tags] == \
            [(k, list(v) if hasattr(v, "typecode") else v) for k, v in y.tags]
        want = {k: v for k, v in dict(y.tags).items() if k in ("mv", "ts", "ns", "sp", "sm", "sd", "pi")}
        hot = dict(x.hot_tags())
        assert np.array_equal(hot.pop("mv"), np.asarray(want.pop("mv"), np.int8)) and hot == want
        for rec in (x, y):
            try:
                rec._r = rec.get_reference_sequence()
            except ValueError as e:
                rec._r = "ValueError"
        assert x._r == y._r


def test_read_indexed_bam(tmp_path):
    """ReadIndexedBam on the reference's BAM files and on a file with secondary / supplementary / split-read
    records: counts, skip reasons, read ids, random access through BGZF virtual offsets (records equal the streamed
    ones), filters, get_read_ids."""
    import struct

    from remora_amd import RemoraError
    from remora_amd import io as rio

    bam = os.path.join(ROOT, "tests", "golden", "data", "can_mappings.bam")
    streamed = list(rio.iter_bam_records(bam, want_ref=True))
    idx = rio.ReadIndexedBam(bam)
    assert idx.num_records == idx.num_reads == 14 and dict(idx.skip_reasons) == {}
    assert idx.read_ids == [r.query_name for r in streamed] and streamed[3].query_name in idx and "nope" not in idx
    for k in (13, 0, 7, 7, 2):  # any order, repeated
        got = list(idx.get_alignments(streamed[k].query_name))
        assert len(got) == 1 and got[0].raw == streamed[k].raw and got[0].voffset == streamed[k].voffset == idx[streamed[k].query_name][0]
        assert got[0].get_reference_sequence() == streamed[k].get_reference_sequence()
    assert idx.get_first_alignment(streamed[5].query_name).query_name == streamed[5].query_name
    with pytest.raises(RemoraError, match="Could not find"):
        list(idx.get_alignments("nope"))
    assert sum(1 for _ in idx) == 14
    # a file with non-primary records and split reads (pi tag): written from modified copies of real records
    recs = list(rio.iter_bam_records(bam, native=False))
    hdr = rio.read_bam_header_bytes(bam)
    path = str(tmp_path / "mix.bam")

    def with_flag(r, flag, extra=b""):
        raw = bytearray(r.raw)
        raw[14:16] = struct.pack("<H", flag)
        return bytes(raw) + extra

    out = []
    for k, r in enumerate(recs * 30):  # many blocks
        flag = r.flag | (256 if k % 5 == 1 else 0) | (2048 if k % 7 == 3 else 0)
        extra = b"piZ" + f"parent-{k % 11}".encode() + b"\x00" if k % 3 == 0 else b""
        out.append(with_flag(r, flag, extra))
    with rio.BamWriter(path, hdr) as w:
        for raw in out:
            w.write(struct.pack("<i", len(raw)) + raw)
    idx = rio.ReadIndexedBam(path)
    primary = [k for k in range(len(out)) if not (k % 5 == 1 or k % 7 == 3)]
    assert idx.num_records == len(primary) and idx.skip_reasons["Non-primary alignment"] == len(out) - len(primary)
    want = {}
    for k in primary:
        want.setdefault(f"parent-{k % 11}" if k % 3 == 0 else recs[k % 14].query_name, []).append(out[k])
    assert set(idx.read_ids) == set(want) and idx.num_reads == len(want)
    for rid in list(want)[::3]:
        assert [a.raw for a in idx.get_alignments(rid, want_ref=False)] == want[rid]
    everything = rio.ReadIndexedBam(path, skip_non_primary=False)
    assert everything.num_records == len(out)
    some = rio.ReadIndexedBam(path, parent_read_id_subset={"parent-3", recs[1].query_name}, req_tags={"mv", "MD"})
    assert set(some.read_ids) <= {"parent-3", recs[1].query_name} and some.skip_reasons["Parent read ID filtered"] > 0
    assert rio.ReadIndexedBam(path, req_tags={"zz"}).num_records == 0

    class P5:
        read_ids = [recs[0].query_name, recs[1].query_name, "only-in-pod5"]

    both, n = rio.get_read_ids(idx, P5, None)
    assert set(both) == {recs[0].query_name, recs[1].query_name} & set(idx.read_ids) and n == len(both)
    both, n = rio.get_read_ids(idx, P5, 1, return_num_bam_reads=True)
    assert n == 1


def test_pod5_tables_are_located_through_the_footer(tmp_path):
    """Pod5File finds the embedded Arrow tables from the container's footer (flatbuffer at the end of the file)
    and only scans for the Arrow magic when the footer is unusable: both ways give the same tables."""
    from remora_amd import RemoraError
    from remora_amd import io as rio

    for which, n_rows in (("can", 17), ("mod", None)):
        path = os.path.join(ROOT, "tests", "golden", "data", f"{which}_reads.pod5")
        raw = open(path, "rb").read()
        files = rio._pod5_embedded_files(raw)
        assert len(files) == 3 and all(raw[o : o + 6] == b"ARROW1" and raw[o + ln - 6 : o + ln] == b"ARROW1" for o, ln in files)
        good = rio.Pod5File(path)
        assert len(good) == 14 and (n_rows is None or good._sig.num_rows == n_rows)
        broken = bytearray(raw)
        broken[-40:-32] = b"\\x00" * 8  # inside the footer flatbuffer's padding / tail
        broken[-32:-24] = (10**9).to_bytes(8, "little")  # absurd footer length
        bad = str(tmp_path / f"{which}_badfooter.pod5")
        open(bad, "wb").write(bytes(broken))
        with pytest.raises(RemoraError):
            rio._pod5_embedded_files(bytes(broken))
        scanned = rio.Pod5File(bad)  # falls back to the magic scan
        assert scanned.read_ids == good.read_ids and scanned._sig.equals(good._sig)
        assert scanned._reads.schema.equals(good._reads.schema)
        for col in ("read_id", "signal", "calibration_offset", "calibration_scale"):  # (other columns hold NaNs)
            assert scanned._reads.column(col).equals(good._reads.column(col)), col
    with pytest.raises(RemoraError, match="not a POD5"):
        junk = str(tmp_path / "junk.pod5")
        open(junk, "wb").write(b"x" * 200)
        rio.Pod5File(junk)


def _tiny_bam(path, records):
    """records: [(name, flag, seq, cigar [(op, len)], [tag bytes])] -> BAM file with one reference."""
    import struct

    from remora_amd import io as rio

    text = b"@HD\tVN:1.6\n@SQ\tSN:chrA\tLN:1000000\n"
    hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1)
    hdr += struct.pack("<i", 5) + b"chrA\x00" + struct.pack("<i", 1000000)
    with rio.BamWriter(str(path), hdr, threads=1) as w:
        for name, flag, seq, cigar, tags in records:
            rname = name.encode()
            body = struct.pack("<iiBBHHHiiii", 0, 100, len(rname) + 1, 60, 4680, len(cigar), flag, len(seq), -1, -1, 0)
            body += rname + b"\x00" + b"".join(struct.pack("<I", (ln << 4) | op) for op, ln in cigar)
            body += rio._pack_seq(seq) + bytes(len(seq)) + b"".join(tags)
            w.write(struct.pack("<i", len(body)) + body)
    return hdr


def test_native_bam_reader_seq_less_secondary_with_md(tmp_path):
    """A mapped secondary record with SEQ '*' (l_seq = 0, as minimap2 / dorado write them) whose CIGAR and MD still
    describe 200 matched bases: the MD reconstruction must refuse it (ref_seq None -> the python path raises
    ValueError for that record only) instead of reading past the sequence arena; the records around it are
    unaffected.  Same for a CIGAR that consumes more query bases than SEQ holds."""
    from remora_amd import io as rio

    md = lambda s: b"MDZ" + s.encode() + b"\x00"  # noqa: E731
    recs = [("prim", 0, "ACGT" * 50, [(0, 200)], [md("200")]),
            ("prim", 256, "", [(0, 200)], [md("200")]),
            ("long", 0, "ACGTACGT", [(4, 2), (0, 20), (1, 3)], [md("20")]),
            ("next", 16, "TTGCA", [(0, 5)], [md("2a2")])]
    path = tmp_path / "secondary.bam"
    _tiny_bam(path, recs)
    nat = list(rio.iter_bam_records(str(path), want_ref=True, batch=8))
    py = list(rio.iter_bam_records(str(path), native=False))
    assert [r.query_name for r in nat] == ["prim", "prim", "long", "next"]
    got = []
    for x, y in zip(nat, py):
        outs = []
        for rec in (x, y):
            try:
                outs.append(rec.get_reference_sequence())
            except ValueError:
                outs.append(None)
        assert outs[0] == outs[1]
        got.append(outs[0])
    assert got == ["ACGT" * 50, None, None, "TTaCA"]
    idx = rio.ReadIndexedBam(str(path), skip_non_primary=True)
    assert idx.num_records == 3 and [r.query_name for r in idx.get_alignments("next")] == ["next"]
    assert idx.get_first_alignment("prim").get_reference_sequence() == "ACGT" * 50
    idx.close()


def test_bam_long_cigar_in_cg_tag(tmp_path):
    """More than 65535 CIGAR operations: the record holds <l_seq>S<ref_len>N and the real CIGAR sits in CG:B,I
    (SAM spec 4.2.2; htslib/pysam resolve it on read).  Both readers hand out the CG operations, and the MD
    reconstruction uses them."""
    import struct

    from remora_amd import io as rio

    real = [(0, 3), (2, 1), (0, 2), (1, 1), (0, 2)]  # 3M1D2M1I2M: query 8, reference 8
    cg = b"CGBI" + struct.pack("<i", len(real)) + b"".join(struct.pack("<I", (ln << 4) | op) for op, ln in real)
    md = b"MDZ3^G4\x00"
    _tiny_bam(tmp_path / "cg.bam", [("ultra", 0, "ACGTTACA", [(4, 8), (3, 8)], [md, cg]),
                                    ("plain", 0, "ACGT", [(4, 4), (3, 9)], []),
                                    ("odd", 0, "ACGTTACA", [(4, 8), (0, 1), (3, 7)], [md, cg])])
    for native in (True, False):
        a, b, c = list(rio.iter_bam_records(str(tmp_path / "cg.bam"), want_ref=True, native=native))
        assert a.cigartuples == real, native
        assert a.get_reference_sequence() == "ACGGTTCA", native  # I base (index 5, 'A') dropped, deleted G inserted
        assert b.cigartuples == [(4, 4), (3, 9)]  # no CG tag: the record's own CIGAR stands
        assert len(a.query_qualities) == 8
        # htslib (bam_tag2cigar) only asks for a first operation of <l_seq>S on a mapped record, whatever follows
        assert c.cigartuples == real, native
        # ... and deletes CG once it has used it: pysam callers never see the tag; the stored bytes still hold it
        assert "CG" not in dict(a.tags) and "CG" not in dict(c.tags) and "MD" in dict(a.tags), native
        assert b"CGBI" in bytes(a.raw)


def test_device_reads_refuse_dacs_that_int16_cannot_hold():
    """rmr_reads.dacs is int16: float / wide-integer dacs go through only when every value is exactly
    representable (RemoraRead.test_read's float zeros); otherwise RemoraError instead of silent truncation."""
    from remora_amd import RemoraError
    from remora_amd.data_chunks import RemoraRead

    mk = lambda d: RemoraRead(dacs=d, shift=0.0, scale=1.0, seq_to_sig_map=np.array([0, 2, 4]),  # noqa: E731
                              int_seq=np.array([1, 2]), read_id="r")
    for bad in (np.array([0.5, 1.0, 2.0, 3.0]), np.array([0, 1, 2, 40000]), np.array([0.0, np.nan, 1.0, 2.0])):
        with pytest.raises(RemoraError, match="int16"):
            _check_dacs_only([mk(bad)])
    from remora_amd import data_chunks as dc

    assert dc._validated_int16_dacs(mk(np.zeros(4))).dtype == np.int16  # float zeros: exact
    assert np.array_equal(dc._validated_int16_dacs(mk(np.array([-32768, 5, 6, 32767], np.int64))), [-32768, 5, 6, 32767])


def _check_dacs_only(reads):
    """The validation DeviceReads applies, without a GPU: mirrors its staging loop on a plain buffer."""
    from remora_amd import data_chunks as dc

    dc._validated_int16_dacs(reads[0])


def test_bam_writer_output_is_valid_bgzf_and_bam_by_independent_readers(tmp_path):
    """The BGZF/BAM writer is otherwise only read back by this repo's own readers; here its output is held to the SAM/BAM
    specification with Python's zlib / gzip as the independent implementation (pysam / htslib are not installed):
      * every BGZF member: gzip magic, CM 8, FLG.FEXTRA, XLEN 6, the 'BC' subfield with SLEN 2 and BSIZE = member size - 1,
        a raw-deflate payload that zlib inflates to ISIZE bytes (<= 65536) whose CRC32 is the trailer's;
      * the file ends with the 28-byte BGZF EOF marker of the specification;
      * gzip.open reads the concatenated members as one stream: 'BAM\\1', l_text, text, n_ref, references, then records
        whose block_size fields walk exactly to the end of the stream; l_read_name / n_cigar_op / l_seq are consistent
        with each record's size, every tag parses, and the records equal what was written."""
    import gzip
    import struct
    import zlib

    from remora_amd import io as rio

    rng = np.random.default_rng(23)
    recs = []
    for k in range(60):
        n = int(rng.integers(1, 300)) if k % 9 else 40000  # some records span several 64 KiB blocks
        seq = "".join(rng.choice(list("ACGT"), n))
        mv = [5] + rng.integers(0, 2, 2 * n).tolist()
        tags = [b"mvBc" + struct.pack("<i", len(mv)) + struct.pack(f"<{len(mv)}b", *mv), b"NMi" + struct.pack("<i", k),
                b"MMZ" + b"C+m,1,2;" + b"\x00"]
        recs.append((f"r{k}", 16 if k % 2 else 0, seq, [(0, n)], tags))
    path = tmp_path / "w.bam"
    hdr = _tiny_bam(path, recs)
    raw = open(path, "rb").read()
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    assert raw.endswith(eof)
    p, payload, members = 0, bytearray(), 0
    while p < len(raw):
        magic, cm, flg, _mtime, _xfl, _os, xlen = struct.unpack_from("<HBBIBBH", raw, p)
        assert (magic, cm, flg, xlen) == (0x8B1F, 8, 4, 6)
        si1, si2, slen, bsize = struct.unpack_from("<BBHH", raw, p + 12)
        assert (si1, si2, slen) == (66, 67, 2)
        end = p + bsize + 1
        crc, isize = struct.unpack_from("<II", raw, end - 8)
        data = zlib.decompress(raw[p + 18 : end - 8], wbits=-15)
        assert len(data) == isize <= 65536 and zlib.crc32(data) == crc
        payload += data
        p = end
        members += 1
    assert p == len(raw) and members >= 3
    stream = gzip.open(path, "rb").read()
    assert stream == bytes(payload) and stream.startswith(hdr)
    q, out = len(hdr), []
    while q < len(stream):
        (bs,) = struct.unpack_from("<i", stream, q)
        rec = stream[q + 4 : q + 4 + bs]
        assert len(rec) == bs
        _ref, _pos, l_name, _mq, _bin, n_cig, _flag, l_seq = struct.unpack_from("<iiBBHHHi", rec, 0)
        body = 32 + l_name + 4 * n_cig + (l_seq + 1) // 2 + l_seq
        assert body <= bs and rec[32 + l_name - 1] == 0
        tags = rio._parse_tags(rec[body:])  # every tag of the region parses to its end
        assert [t for t, _ in tags] == ["mv", "NM", "MM"]
        out.append(rec)
        q += 4 + bs
    assert q == len(stream) and len(out) == len(recs)
    back = list(rio.iter_bam_records(str(path)))
    assert [r.query_name for r in back] == [r[0] for r in recs] and [r.query_sequence for r in back] == [r[2] for r in recs]


def test_pack_reads_native_gather_equals_numpy():
    """rmr_pack_reads (host-only C++: the per-read arrays of a batch gathered into the concatenated rmr_reads layout by
    native threads) against numpy concatenation: every int_seq itemsize, empty reads, any thread count."""
    import ctypes

    from remora_amd import _lib as L

    lib = L.lib()
    rng = np.random.default_rng(5)
    nr = 37
    dts = [np.int8, np.int16, np.int32, np.int64, np.uint8]
    reads = []
    for i in range(nr):
        nb = 0 if i == 11 else int(rng.integers(1, 400))
        ns = 0 if nb == 0 else int(rng.integers(nb, 12 * nb))
        reads.append((rng.integers(-3000, 3000, ns).astype(np.int16), np.sort(rng.integers(0, ns + 1, nb + 1)).astype(np.int64),
                      rng.integers(-1, 4, nb).astype(dts[i % len(dts)] if dts[i % len(dts)] != np.uint8 else np.int8)))
    sig_n = np.asarray([r[0].size for r in reads], np.int64)
    seq_n = np.asarray([r[2].size for r in reads], np.int64)
    isz = np.asarray([r[2].dtype.itemsize for r in reads], np.int32)
    vp = lambda k: (ctypes.c_void_p * nr)(*[r[k].ctypes.data for r in reads])  # noqa: E731
    for threads in (1, 3, 8):
        dacs = np.full(int(sig_n.sum()) + 1, 7777, np.int16)
        maps = np.full(int(seq_n.sum()) + nr + 1, -5, np.int64)
        seq = np.full(int(seq_n.sum()) + 1, 99, np.int8)
        so, qo = np.empty(nr + 1, np.int64), np.empty(nr + 1, np.int64)
        L.check(lib.rmr_pack_reads(nr, vp(0), sig_n.ctypes.data, vp(1), vp(2), seq_n.ctypes.data, isz.ctypes.data, dacs.ctypes.data,
                                   maps.ctypes.data, seq.ctypes.data, so.ctypes.data, qo.ctypes.data, threads))
        assert np.array_equal(so, np.concatenate([[0], np.cumsum(sig_n)])) and np.array_equal(qo, np.concatenate([[0], np.cumsum(seq_n)]))
        assert np.array_equal(dacs[:-1], np.concatenate([r[0] for r in reads])) and dacs[-1] == 7777
        assert np.array_equal(maps[:-1], np.concatenate([r[1] for r in reads])) and maps[-1] == -5
        assert np.array_equal(seq[:-1], np.concatenate([r[2].astype(np.int8) for r in reads])) and seq[-1] == 99
    bad = isz.copy()
    bad[3] = 3
    assert lib.rmr_pack_reads(nr, vp(0), sig_n.ctypes.data, vp(1), vp(2), seq_n.ctypes.data, bad.ctypes.data, dacs.ctypes.data,
                              maps.ctypes.data, seq.ctypes.data, so.ctypes.data, qo.ctypes.data, 2) != 0
    # the narrow form: the mapping as int32 at the same offsets, and the flag when a value does not fit
    for threads in (1, 5):
        maps32 = np.full(int(seq_n.sum()) + nr + 1, -5, np.int32)
        fit = ctypes.c_int(-1)
        L.check(lib.rmr_pack_reads_narrow(nr, vp(0), sig_n.ctypes.data, vp(1), vp(2), seq_n.ctypes.data, isz.ctypes.data, dacs.ctypes.data,
                                          maps32.ctypes.data, seq.ctypes.data, so.ctypes.data, qo.ctypes.data, threads, ctypes.byref(fit)))
        assert fit.value == 1 and np.array_equal(maps32[:-1], np.concatenate([r[1] for r in reads])) and maps32[-1] == -5
        assert np.array_equal(dacs[:-1], np.concatenate([r[0] for r in reads]))
    for where, value in ((0, 1 << 31), (-1, -(1 << 31) - 1), (2, 1 << 40)):
        keep = reads[20][1][where]
        reads[20][1][where] = value
        L.check(lib.rmr_pack_reads_narrow(nr, vp(0), sig_n.ctypes.data, vp(1), vp(2), seq_n.ctypes.data, isz.ctypes.data, dacs.ctypes.data,
                                          maps32.ctypes.data, seq.ctypes.data, so.ctypes.data, qo.ctypes.data, 3, ctypes.byref(fit)))
        assert fit.value == 0, (where, value)
        reads[20][1][where] = keep
    reads[20][1][0] = -(1 << 31)  # the smallest value that fits
    L.check(lib.rmr_pack_reads_narrow(nr, vp(0), sig_n.ctypes.data, vp(1), vp(2), seq_n.ctypes.data, isz.ctypes.data, dacs.ctypes.data,
                                      maps32.ctypes.data, seq.ctypes.data, so.ctypes.data, qo.ctypes.data, 3, ctypes.byref(fit)))
    assert fit.value == 1 and maps32[qo[20] + 20] == -(1 << 31)
    # int64 bases far outside int8 never narrow to a valid base code (2^32 + 1 kept only its low dword once): clamped to the
    # int8 range on the vector path (16 at a time) and in the scalar tail alike, so RemoraRead.check's range test sees them
    wide = np.array([0, 1, (1 << 32) + 1, -(1 << 32), 3, 1 << 40, 300, -300, 2, (1 << 31), -(1 << 31) - 1, 1, 0, 3, 2, -1,
                     (1 << 32) + 2, -1, 127, -128], np.int64)
    one = lambda a: (ctypes.c_void_p * 1)(a.ctypes.data)  # noqa: E731
    d1, m1 = np.zeros(4, np.int16), np.arange(wide.size + 1, dtype=np.int64)
    out8 = np.zeros(wide.size, np.int8)
    n_sig, n_seq, w8 = np.array([4], np.int64), np.array([wide.size], np.int64), np.array([8], np.int32)
    o_d, o_m, o_so, o_qo = np.zeros(4, np.int16), np.zeros(wide.size + 1, np.int64), np.zeros(2, np.int64), np.zeros(2, np.int64)
    L.check(lib.rmr_pack_reads(1, one(d1), n_sig.ctypes.data, one(m1), one(wide), n_seq.ctypes.data, w8.ctypes.data, o_d.ctypes.data,
                               o_m.ctypes.data, out8.ctypes.data, o_so.ctypes.data, o_qo.ctypes.data, 1))
    assert np.array_equal(out8, np.clip(wide, -128, 127).astype(np.int8)), out8


def test_format_mm_ml_tags_equals_oracle_on_random_reads(O):
    """The MM gap computation (search in the canonical-base positions) against the oracle's running-count formulation
    (src/remora/util.py:485-537) on random sequences: unsorted positions, first / last base called, several mods."""
    from remora_amd import util

    rng = np.random.default_rng(42)
    for trial in range(40):
        n = int(rng.integers(1, 400))
        seq = "".join(rng.choice(list("ACGTN"), n))
        can = "ACGT"[trial % 4]
        sites = np.flatnonzero(np.frombuffer(seq.encode(), np.uint8) == ord(can))
        if sites.size == 0:
            continue
        poss = rng.permutation(sites[rng.random(sites.size) < 0.6]) if trial % 3 else sites.copy()
        nm = 1 + trial % 2
        probs = rng.random((poss.size, nm)) / nm
        if poss.size:
            probs[0, 0] = 1.0  # floor(p * 256) == 256 is clipped to 255
        mods = ["m", "h"][:nm]
        want = O.format_mm_ml_tags(seq, poss, probs, mods, can)
        got = util.format_mm_ml_tags(seq, poss, probs, mods, can)
        assert got[0] == want[0], (trial, seq, poss)
        assert list(got[1]) == list(want[1])


def test_batched_line_fits_equal_per_read_lstsq():
    """refine_signal_map._fit_lines (numpy's LAPACK gufunc on the stacked systems) must return, bit for bit, what
    np.linalg.lstsq returns row by row - including rank-deficient rows (all x equal)."""
    from remora_amd import refine_signal_map as R

    rng = np.random.default_rng(7)
    X = np.sort(rng.normal(0, 1, (50, 19)), axis=1)
    Y = 0.8 * X + 0.1 + 0.05 * rng.normal(0, 1, X.shape)
    X[3] = 0.25  # degenerate: lstsq's minimum-norm solution
    Y[4] = 0.0
    got = R._fit_lines(X, Y)
    want = np.asarray([R._fit_line(X[i], Y[i]) for i in range(X.shape[0])])
    assert got.shape == (50, 2)
    np.testing.assert_array_equal(got, want)
    assert R._fit_lines(np.zeros((0, 19)), np.zeros((0, 19))).shape == (0, 2)


def test_bam_writer_thread_count_and_level_do_not_change_the_payload(tmp_path):
    """More deflate threads (the default is now up to 16) or another level change the compressed bytes at most, never
    the records: every variant inflates to the same stream."""
    import gzip

    from remora_amd import io as rio

    rng = np.random.default_rng(0)
    header = b"BAM\x01" + (0).to_bytes(4, "little") + (0).to_bytes(4, "little")
    recs = [bytes(rng.integers(0, 256, int(rng.integers(100, 40000)), dtype=np.uint8)) for _ in range(60)]
    payloads = []
    for threads in (1, 3, 16):
        p = tmp_path / f"t{threads}.bam"
        with rio.BamWriter(str(p), header, threads=threads) as w:
            for r in recs:
                w.write(len(r).to_bytes(4, "little") + r)
        payloads.append(gzip.open(str(p)).read())
    assert payloads[0] == payloads[1] == payloads[2]
    assert payloads[0] == header + b"".join(len(r).to_bytes(4, "little") + r for r in recs)


def test_c_abi_is_plain_c99_and_fails_cleanly_without_a_gpu(tmp_path):
    """include/remora_hip.h compiles as pedantic C99, libremora_hip.so links from C, and a C caller gets an error code
    plus a message (never an abort) when there is no GPU - or a working engine when there is one (tests/c/abi_from_c.c)."""
    import shutil
    import subprocess

    from remora_amd import _lib

    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    exe = tmp_path / "abi_from_c"
    libdir = os.path.dirname(_lib.LIB_PATH)
    cc = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                         os.path.join(ROOT, "tests", "c", "abi_from_c.c"), "-o", str(exe), "-L", libdir, "-lremora_hip",
                         f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, (run.stdout, run.stderr)
    out = run.stdout
    assert "version=remora_hip" in out and "gfx950" in out
    assert "weights=" in out and int(out.split("weights=")[1].split()[0]) > 100000
    assert "engine ok rc=0" in out or ("engine_create rc=-" in out and "error=" in out and len(out.split("error=")[1].strip()) > 5)


def test_bench_result_line_is_small_and_complete():
    """The driver keeps a bounded tail of bench.py's stdout: the ONE result line is built by `bench.result_line` from the full
    report and must stay below 4 KB with the contract's keys; everything else lives in bench_details.json.  Canned report:
    the full round-2 output (20 KB, the line the driver could not parse)."""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_final.json")))
    assert len(json.dumps(full)) > 16000
    full["details_file"] = "bench_details.json"
    full["label_counts_match_logits_argmax"] = True
    line = bench.result_line(full)
    txt = json.dumps(line)
    assert len(txt) < bench.LINE_LIMIT == 4096, len(txt)
    assert json.loads(txt) == line
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline", "parity"}
    assert need <= set(line), sorted(need - set(line))
    assert set(line["config"]) == {"workload", "name", "baseline_config", "chunks_per_step_all_gpus"}
    assert {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "chunks_per_launch",
            "avg_launch_ms"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["value"] == full["value"]
    assert "other_configs" not in line and "kernels" not in line and "forms" not in line["cpu_baseline"]
    assert line["parity"]["label_counts_exact"] is True and line["parity"]["fp32_max_abs"] < 1e-4


def test_c_inference_program_builds_and_reports_errors_without_a_gpu(tmp_path):
    """tests/c/infer_from_c.c (the C caller that runs rmr_model_create + rmr_infer_chunks, -m gpu: tests/test_gpu_c_abi.py)
    compiles as pedantic C99 against the header, reads the exported fixture, and on a box without a GPU stops at
    rmr_engine_create with the library's message and exit code 2 — never a crash."""
    import shutil
    import subprocess

    from remora_amd import _lib

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import export_c_fixture

    fx = str(tmp_path / "fixture.bin")
    n, nw = export_c_fixture.export(os.path.join(ROOT, "tests", "golden", "model_convlstm_s64_l100_o2.npz"), fx)
    assert (n, nw) == (48, 134538)
    exe = str(tmp_path / "infer_from_c")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cc = subprocess.run([shutil.which("gcc"), "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                         os.path.join(ROOT, "tests", "c", "infer_from_c.c"), "-o", exe, "-L", libdir, "-lremora_hip", "-lm",
                         f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe, fx], capture_output=True, text=True, timeout=120)
    import torch

    if torch.cuda.is_available():
        assert run.returncode == 0 and "OK" in run.stdout, (run.stdout, run.stderr)
    else:
        assert run.returncode == 2 and "rmr_engine_create" in run.stderr and "rc -2" in run.stderr, (run.stdout, run.stderr)


# ---- multi-GPU product pipeline: the sharding plumbing on CPU (the kernels' side: tests/test_gpu_shard.py) --------
def test_bam_byte_shares_partition_the_records_and_verify_their_ends(tmp_path):
    """io.bam_byte_shard (rmr_bam_guess_start): for any number of workers the shares by byte range are contiguous, in rank
    order and together exactly the records of the file - found without a pass over it; rmr_bam_guess_start agrees with
    the exact record list at every probed offset; a share whose end mark is not a record start is refused."""
    import random
    import struct

    from remora_amd import RemoraError
    from remora_amd import io as rio

    big = str(tmp_path / "big.bam")
    recs = list(rio.iter_bam_records(os.path.join(DATA, "can_mappings.bam"))) + list(rio.iter_bam_records(os.path.join(DATA, "mod_mappings.bam")))
    with rio.BamWriter(big, rio.read_bam_header_bytes(os.path.join(DATA, "can_mappings.bam")), level=1) as w:
        for k in range(12):
            for r in recs:
                raw = bytes(r.raw)
                w.write(struct.pack("<i", len(raw)) + raw)
    for path in (big, os.path.join(DATA, "can_mappings.bam"), os.path.join(DATA, "mod_mappings.bam")):
        whole = [(r.query_name, r.flag, r.voffset) for r in rio.iter_bam_records(path)]
        vos = [w[2] for w in whole]
        size = os.path.getsize(path)
        rng = random.Random(7)
        for off in [0, 1, size // 3, size // 2, size - 29, size - 28, size - 1, size, size + 9] + [rng.randrange(size) for _ in range(60)]:
            want = next((v for v in vos if (v >> 16) >= off), None) if off > (vos[0] >> 16) else vos[0]
            assert rio.bam_guess_start(path, off) == want, (path, off)
        for world in (1, 2, 3, 5, 8, 16, 33):
            got = []
            for rank in range(world):
                part = [(r.query_name, r.flag, r.voffset) for r in rio.iter_bam_records(path, shard=(rank, world))]
                st, en = rio.bam_byte_shard(path, rank, world)
                assert (part[0][2] == st if part else True) and (st is None or en is None or en > st)
                got += part
            assert got == whole, (path, world)
    # a file without records, and one with a single record: empty shares, nothing lost
    header = rio.read_bam_header_bytes(os.path.join(DATA, "can_mappings.bam"))
    none, single = str(tmp_path / "none.bam"), str(tmp_path / "single.bam")
    with rio.BamWriter(none, header):
        pass
    with rio.BamWriter(single, header) as w:
        raw = bytes(recs[0].raw)
        w.write(struct.pack("<i", len(raw)) + raw)
    for world in (1, 2, 5):
        assert [len(list(rio.iter_bam_records(none, shard=(r, world)))) for r in range(world)] == [0] * world
        assert sum(len(list(rio.iter_bam_records(single, shard=(r, world)))) for r in range(world)) == 1
    assert rio.bam_guess_start(none, 0) is None and rio.bam_byte_shard(none, 0, 1) == (None, None)
    # an end mark that is no record start: the share in front runs past it and says so
    whole = [(r.query_name, r.flag, r.voffset) for r in rio.iter_bam_records(big)]
    vos, size = [w[2] for w in whole], os.path.getsize(big)
    with pytest.raises(RemoraError, match="guessed wrong"):
        list(rio._iter_bam_records_native(big, False, 64, start_voffset=vos[0], end_voffset=whole[5][2] + 1))
    with pytest.raises(RemoraError, match="end of file before"):
        list(rio._iter_bam_records_native(big, False, 64, start_voffset=vos[0], end_voffset=(size + 5) << 16))


def test_a_wrong_share_boundary_is_refused_before_any_work(tmp_path, monkeypatch):
    """bam_byte_shard checks both of its marks against an independent chain of records before the rank starts (round 3
    found a wrong guess only at the END of the share in front, behind all of its GPU work): a guess that is no record start
    raises at once, naming the scan mode; the true marks pass, with or without the check."""
    import struct

    from remora_amd import RemoraError
    from remora_amd import io as rio

    big = str(tmp_path / "big.bam")
    recs = list(rio.iter_bam_records(os.path.join(DATA, "can_mappings.bam")))
    with rio.BamWriter(big, rio.read_bam_header_bytes(os.path.join(DATA, "can_mappings.bam")), level=1) as w:
        for _ in range(40):
            for r in recs:
                raw = bytes(r.raw)
                w.write(struct.pack("<i", len(raw)) + raw)
    honest = [rio.bam_byte_shard(big, r, 4) for r in range(4)]
    assert all(honest[r][1] == honest[r + 1][0] for r in range(3)) and honest[3][1] is None
    monkeypatch.setenv("REMORA_AMD_BAM_SHARD_VERIFY", "0")
    assert [rio.bam_byte_shard(big, r, 4) for r in range(4)] == honest
    monkeypatch.delenv("REMORA_AMD_BAM_SHARD_VERIFY")
    real, size = rio.bam_guess_start, os.path.getsize(big)

    def off_by_some(path, file_offset):  # the mark of rank 2 lands inside a record
        v = real(path, file_offset)
        return v + 7 if file_offset == size * 2 // 4 else v

    monkeypatch.setattr(rio, "bam_guess_start", off_by_some)
    for rank in (1, 2):  # rank 1's end and rank 2's start are that mark
        with pytest.raises(RemoraError, match="not a record start.*REMORA_AMD_BAM_SHARD=scan"):
            rio.bam_byte_shard(big, rank, 4)
    assert rio.bam_byte_shard(big, 0, 4) == honest[0] and rio.bam_byte_shard(big, 3, 4) == honest[3]


def test_raw_bam_batches_are_the_records_of_iter_bam_records(tmp_path, monkeypatch):
    """io.iter_bam_raw_batches (flat arrays per native batch: what the batch ingest of `infer` runs on) hands out exactly
    the records iter_bam_records does - whole file, shares by byte range and by record count, any batch size - and a share
    whose end mark is not a record start is refused the same way; RawBamBatch.head cuts a batch without copying its blobs."""
    import struct

    from remora_amd import RemoraError
    from remora_amd import io as rio

    big = str(tmp_path / "big.bam")
    recs = list(rio.iter_bam_records(os.path.join(DATA, "can_mappings.bam"))) + list(rio.iter_bam_records(os.path.join(DATA, "mod_mappings.bam")))
    with rio.BamWriter(big, rio.read_bam_header_bytes(os.path.join(DATA, "can_mappings.bam")), level=1) as w:
        for _ in range(6):
            for r in recs:
                raw = bytes(r.raw)
                w.write(struct.pack("<i", len(raw)) + raw)

    def key(r):
        return (r.query_name, r.flag, r.voffset, bytes(r.raw), r.tags_offset, r.query_sequence, tuple(sorted((k, np.asarray(v).tobytes() if k == "mv" else v)
                                                                                                            for k, v in r.hot_tags().items())))

    def from_raw(path, batch, shard=None):
        out = []
        for rb, records in rio.iter_bam_raw_batches(path, batch=batch, shard=shard):
            assert rb.n == rb.flag.size == rb.voffset.size == rb.raw_off.size - 1 and int(rb.raw_off[-1]) <= len(rb.raw)
            got = records(rb)
            assert len(got) == rb.n
            # the flat arrays say what the record objects say
            for i, r in enumerate(got):
                assert rb.raw[rb.raw_off[i] : rb.raw_off[i + 1]] == bytes(r.raw) and rb.seq[rb.seq_off[i] : rb.seq_off[i + 1]].decode() == r.query_sequence
                assert int(rb.tags_off[i]) == r.tags_offset and int(rb.flag[i]) == r.flag
                assert np.array_equal(rb.mv[rb.mv_off[i] : rb.mv_off[i + 1]], r.hot_tags().get("mv", np.zeros(0, np.int8)))
            out += [key(r) for r in got]
        return out

    whole = [key(r) for r in rio.iter_bam_records(big)]
    for batch in (1, 7, 64, 512):
        assert from_raw(big, batch) == whole
    for mode in ("bytes", "scan"):
        monkeypatch.setenv("REMORA_AMD_BAM_SHARD", mode)
        for world in (2, 3, 5):
            parts = [from_raw(big, 16, shard=(rank, world)) for rank in range(world)]
            assert [k for p in parts for k in p] == whole, (mode, world)
            assert parts == [[key(r) for r in rio.iter_bam_records(big, shard=(rank, world))] for rank in range(world)]
    monkeypatch.delenv("REMORA_AMD_BAM_SHARD")
    rb, records = next(iter(rio.iter_bam_raw_batches(big, batch=9)))
    cut = rb.head(4)
    assert cut.n == 4 and cut.raw is rb.raw and cut.raw_off.size == 5 and [key(r) for r in records(cut)] == whole[:4]

    class Share:  # a byte-range share whose end is one byte behind a record start
        def result(self):
            return ("bytes", whole[0][2], whole[20][2] + 1)

    with pytest.raises(RemoraError, match="guessed wrong"):
        list(rio.iter_bam_raw_batches(big, batch=8, shard=Share()))


@pytest.mark.parametrize("name", ["can", "mod"])
def test_reference_anchored_host_path_from_cigar_arrays(name):
    """Reference-anchored reads on the host: Read.add_alignment feeds the native reader's CIGAR arrays to
    make_sequence_coordinate_mapping (src/remora/data_chunks.py:77-115) - the knots, ref_to_signal and the training reads of
    `dataset prepare` (prepare_train_data.py:53-91) equal those of the tuple form, which the reference-generated goldens pin
    on the GPU side; a random CIGAR maps the same through every accepted form, errors included."""
    from golden_util import pod5_reads_cpu
    from oracle import oracle as O
    from remora_amd import RemoraError, util
    from remora_amd import io as rio
    from remora_amd.data_chunks import compute_ref_to_signal, make_sequence_coordinate_mapping
    from remora_amd.prepare_train_data import _training_read

    motifs = [util.Motif("CG", 0)]
    pods = {p.read_id: p for p in pod5_reads_cpu(os.path.join(DATA, f"{name}_reads.pod5"))}
    for rec in rio.iter_bam_records(os.path.join(DATA, f"{name}_mappings.bam"), want_ref=True):
        t, sig = rec.hot_tags(), pods[rec.query_name].signal
        sl = len(range(sig.size)[t.get("sp", 0) :][t.get("ts", 0) : t.get("ns", None)])
        mv = O.parse_move_tag(np.asarray(t["mv"], np.int8), sl, seq_len=len(rec.query_sequence))
        r = rio.Read.from_pod5(pods[rec.query_name])
        r.add_alignment(rec, parse_ref_align=True, parsed_moves=mv)
        assert np.array_equal(r.ref_to_signal, compute_ref_to_signal(r.query_to_signal, r.cigar))  # arrays == tuples
        again = rio.Read.from_pod5(pods[rec.query_name])
        again.add_alignment(rec, parse_ref_align=True, parsed_moves=mv)
        again.ref_to_signal = None  # what prepare_train_data computed itself before it reused add_alignment's
        a, b = _training_read(r, 1, motifs, None, False), _training_read(again, 1, motifs, None, False)
        assert np.array_equal(a.dacs, b.dacs) and np.array_equal(a.seq_to_sig_map, b.seq_to_sig_map) and a.str_seq == b.str_seq
        assert np.array_equal(a.focus_bases, b.focus_bases) and (a.shift, a.scale) == (b.shift, b.scale)
    rng = np.random.default_rng(5)
    for _ in range(500):
        n = int(rng.integers(0, 30))
        ops, lens = rng.integers(0, 9, n), rng.integers(0, 30, n)
        forms = ([(int(o), int(l)) for o, l in zip(ops, lens)], (ops, lens), np.stack([ops, lens], 1).reshape(n, 2))
        got = []
        for f in forms:
            try:
                got.append(make_sequence_coordinate_mapping(f).tolist())
            except RemoraError as e:
                got.append(str(e))
        assert got[0] == got[1] == got[2]


def test_native_ref_to_signal_equals_the_two_interpolations():
    """rmr_ref_to_signal (one walk over BAM's uint32 CIGAR) against compute_ref_to_signal (np.interp twice + floor,
    src/remora/data_chunks.py:60-122) on random alignments: every operation, zero-length non-match operations, both strands,
    long deletions / skips (fractional query coordinates), move tables shorter and longer than the CIGAR's query length
    (np.interp's clamping), a buffer that is too small first; the same integers and the same error texts."""
    from remora_amd import RemoraError
    from remora_amd import io as rio
    from remora_amd.data_chunks import compute_ref_to_signal

    rng = np.random.default_rng(11)
    compared = 0
    for trial in range(4000):
        n_ops = int(rng.integers(1, 14))
        ops = rng.choice(9, n_ops, p=[0.4, 0.12, 0.12, 0.04, 0.08, 0.04, 0.02, 0.09, 0.09])
        top = 400 if trial % 7 == 0 else 9
        lens = rng.integers(1, top, n_ops)
        if trial % 10 == 0:
            lens[~np.isin(ops, [0, 7, 8])] = rng.integers(0, 3, int((~np.isin(ops, [0, 7, 8])).sum()))
        cig = ((lens.astype(np.uint32) << 4) | ops.astype(np.uint32)).astype(np.uint32)
        rev = bool(rng.integers(0, 2))
        q_len = int(lens[np.isin(ops, [0, 1, 4, 7, 8])].sum())
        q2s = np.cumsum(rng.integers(0, 12, max(1, q_len + 1 + int(rng.choice([0, 0, 0, -2, 3]))))).astype(np.int64)
        o, ln = (ops[::-1], lens[::-1]) if rev else (ops, lens)
        try:
            want = compute_ref_to_signal(q2s, (o.astype(np.int64), ln.astype(np.int64)))
        except RemoraError as e:
            want = str(e)
        try:
            got = rio._ref_to_signal_of_bam_cigar(cig, rev, q2s, expect=int(rng.integers(1, 50)) if trial % 3 == 0 else
                                                  int(lens[np.isin(ops, [0, 2, 3, 7, 8])].sum()) + 1)
        except RemoraError as e:
            got = str(e)
        if isinstance(want, str) or isinstance(got, str):
            assert want == got, (ops, lens)
        else:
            assert got.dtype == np.int64 and np.array_equal(want, got), (ops, lens, rev)
            compared += 1
    assert compared > 3000
    bad = np.array([(5 << 4) | 9], np.uint32)
    with pytest.raises(RemoraError, match="Invalid cigar op"):
        rio._ref_to_signal_of_bam_cigar(bad, False, np.arange(6), 6)
    with pytest.raises(RemoraError, match="No match operations"):
        rio._ref_to_signal_of_bam_cigar(np.array([(5 << 4) | 4], np.uint32), False, np.arange(6), 6)


def test_batch_trimming_is_python_slicing():
    """io._trim_span (the sp / ts / ns trimming of a whole batch, the array form of `dacs[sp:][ts:ns]` in
    Read.add_alignment, src/remora/io.py:2003-2012) against the slices themselves, bounds beyond every edge included."""
    from remora_amd import io as rio

    rng = np.random.default_rng(11)
    n = 4000
    size = rng.integers(0, 50, n)
    sp, ts, ns = (rng.integers(0, 70, n) for _ in range(3))
    has_ns = rng.random(n) < 0.7
    off, length = rio._trim_span(size, sp, ts, ns, has_ns)
    for k in range(n):
        want = range(int(size[k]))[int(sp[k]) :][int(ts[k]) : (int(ns[k]) if has_ns[k] else None)]
        assert len(want) == length[k]
        if len(want):
            assert want[0] == off[k]


def test_pod5_rows_located_by_array_arithmetic():
    """Pod5File.rows_of_reads (addresses into the mapped Arrow buffers, no bytes object per row) names the same bytes and
    sample counts as the per-cell access of signal_rows, for any order and repetition of reads."""
    import ctypes

    from remora_amd import io as rio

    for name in ("can", "mod"):
        f = rio.Pod5File(os.path.join(DATA, f"{name}_reads.pod5"))
        order = list(range(len(f.read_ids))) + [3, 3, 0]
        first, addr, size, samples = f.rows_of_reads([f._row[f.read_ids[k]] for k in order])
        assert first[0] == 0 and first.size == len(order) + 1
        for j, k in enumerate(order):
            want = f.signal_rows(f.read_ids[k])
            got = [(ctypes.string_at(int(addr[i]), int(size[i])), int(samples[i])) for i in range(first[j], first[j + 1])]
            assert got == [(bytes(b), int(n)) for b, n in want]
        e = f.rows_of_reads([])
        assert e[0].tolist() == [0] and all(x.size == 0 for x in e[1:])


def test_bam_shard_partitions_the_records_in_order(monkeypatch):
    """io.bam_shard (rmr_bam_scan + rmr_bam_seek): for any number of workers and any mark spacing the workers' shares are
    contiguous, in rank order, and together exactly the records of the file; shares differ by less than one mark spacing
    (+1 where the marks do not divide evenly)."""
    from remora_amd import io as rio

    monkeypatch.setenv("REMORA_AMD_BAM_SHARD", "scan")  # (the default splits by byte range: the test above)
    for name in ("can_mappings.bam", "mod_mappings.bam"):
        path = os.path.join(DATA, name)
        whole = [(r.query_name, r.flag, r.voffset) for r in rio.iter_bam_records(path)]
        assert len(whole) >= 14
        for world in (1, 2, 3, 5, 8, 16):
            for every in (1, 2, 64):
                got, sizes = [], []
                for rank in range(world):
                    part = [(r.query_name, r.flag, r.voffset) for r in rio.iter_bam_records(path, shard=(rank, world))] if every == 64 else None
                    st, n = rio.bam_shard(path, rank, world, every=every)
                    mine = ([(r.query_name, r.flag, r.voffset) for r in rio._iter_bam_records_native(path, False, 4, start_voffset=st,
                                                                                                      max_records=n)] if n else [])
                    if world == 1:
                        assert (st, n) == (None, None)
                        mine = whole
                    assert part is None or part == mine
                    got += mine
                    sizes.append(len(mine))
                assert got == whole, (name, world, every)
                if every == 1 and world <= len(whole):
                    assert max(sizes) - min(sizes) <= 1


def test_bam_shard_takes_the_launchers_scan(tmp_path, monkeypatch):
    """dist.launch_ranks scans the BAM once while its ranks start and announces the result through REMORA_AMD_BAM_SCAN;
    bam_shard then gives exactly the shares of a rank's own scan.  A scan of another file, a failed scan or a missing
    file fall back to the rank's own scan, never to wrong offsets."""
    import shutil

    from remora_amd import io as rio

    path = os.path.join(DATA, "can_mappings.bam")
    own = [rio.bam_shard(path, r, 3, every=2) for r in range(3)]
    scan = str(tmp_path / "scan.npz")
    rio.write_bam_scan(path, scan, every=2)
    monkeypatch.setenv(rio.SCAN_ENV, scan)
    marks, n = rio._launcher_bam_scan(path, 2)
    assert (marks.tolist(), n) == (rio.bam_scan(path, 2)[0].tolist(), rio.bam_scan(path, 2)[1])
    calls = []
    real = rio.bam_scan
    monkeypatch.setattr(rio, "bam_scan", lambda *a, **k: (calls.append(a), real(*a, **k))[1])
    assert [rio.bam_shard(path, r, 3, every=2) for r in range(3)] == own and not calls
    # the announced scan is about another file (or another mark spacing): not used
    other = str(tmp_path / "other.bam")
    shutil.copy(os.path.join(DATA, "mod_mappings.bam"), other)
    assert rio._launcher_bam_scan(other, 2) is None and rio._launcher_bam_scan(path, 3) is None
    assert rio.bam_shard(other, 1, 2, every=2) == rio.bam_shard(other, 1, 2, every=2) and len(calls) == 2
    # a failed launcher scan leaves a marker; a scan that never appears times out
    rio.write_bam_scan(str(tmp_path / "missing.bam"), scan, every=2)
    assert rio._launcher_bam_scan(path, 2) is None
    monkeypatch.setenv(rio.SCAN_ENV, str(tmp_path / "never.npz"))
    assert rio._launcher_bam_scan(path, 2, wait_s=0.05) is None


def test_bam_parts_join_into_one_valid_bam(tmp_path):
    """The part files of a multi-GPU infer run (rank 0: header + records, others: records only, no EOF markers) joined by
    concat_bam_parts read back as ONE BAM with every record in input order - by Python's gzip (independent reader) and by
    the native reader; the file ends with the 28-byte BGZF EOF marker."""
    import gzip

    from remora_amd import io as rio

    src = os.path.join(DATA, "can_mappings.bam")
    header = rio.read_bam_header_bytes(src)
    recs = list(rio.iter_bam_records(src))
    raw = [struct_pack_record(r) for r in recs]
    world = 3
    out = str(tmp_path / "joined.bam")
    for rank in range(world):
        st, n = rio.bam_shard(src, rank, world, every=1)
        with rio.BamWriter(f"{out}.part{rank:03d}", header if rank == 0 else b"", eof=False, threads=2) as w:
            for r in rio._iter_bam_records_native(src, False, 8, start_voffset=st, max_records=n):
                w.write(rio.record_with_mod_tags(r, None, None))
    how = rio.concat_bam_parts(out, [f"{out}.part{r:03d}" for r in range(world)])
    # the two appended parts went through the kernel (copy_file_range / sendfile), not through the read-write loop: an
    # output opened with O_APPEND made both fail silently (round-3 advice)
    assert how["read_write"] == 0 and how["copy_file_range"] + how["sendfile"] == world - 1, how
    assert not any(p.startswith("joined.bam.part") for p in os.listdir(tmp_path))
    blob = open(out, "rb").read()
    assert blob.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    plain = gzip.decompress(blob)
    assert plain.startswith(header) and plain[len(header):] == b"".join(raw)
    back = list(rio.iter_bam_records(out))
    assert [r.query_name for r in back] == [r.query_name for r in recs]


def test_one_shot_inflater_equals_zlib_on_every_kind_of_stream(tmp_path):
    """rmr_inflate_raw (the native BAM reader's inflater for BGZF members, fast_inflate.h) against zlib: streams of every
    block type (stored, fixed, dynamic), level and strategy, several blocks per stream, small windows, literal-only and
    match-heavy data inflate to the same bytes; a wrong output size, a truncated or a bit-flipped stream is refused or -
    at worst - yields the right NUMBER of bytes (the reader checks the member's CRC32 and falls back to zlib); and the
    reader gives the same records with and without it."""
    import ctypes
    import zlib

    from remora_amd import _lib as L
    from remora_amd import io as rio

    lib = L.lib()

    def inflate(z, n):
        out = np.empty(max(n, 1), np.uint8)
        src = np.frombuffer(z, np.uint8) if len(z) else np.zeros(1, np.uint8)
        rc = lib.rmr_inflate_raw(src.ctypes.data_as(ctypes.c_void_p), len(z), out.ctypes.data_as(ctypes.c_void_p), n)
        return rc, out[:n].tobytes()

    rng = np.random.default_rng(3)
    raw = next(iter(rio.iter_bam_raw_batches(os.path.join(DATA, "mod_mappings.bam"), batch=14)))[0].raw
    datasets = [b"", b"a", b"ab", bytes(100000), rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),
                rng.integers(0, 4, 90000, dtype=np.uint8).tobytes(), (rng.geometric(0.05, 120000) % 256).astype(np.uint8).tobytes(),
                (rng.geometric(0.5, 120001) % 256).astype(np.uint8).tobytes(), b"abcabcabcd" * 7000,
                b"".join(bytes([i % 256]) * (i % 37 + 1) for i in range(5000)), raw[:300000],
                open(os.path.join(DATA, "can_mappings.bam"), "rb").read()[:150000]]
    for data in datasets:
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                for wbits in ((-15, -9) if len(data) < 100000 else (-15,)):
                    c = zlib.compressobj(level, zlib.DEFLATED, wbits, 9, strategy)
                    z = b"".join([c.compress(data[: len(data) // 3]), c.flush(zlib.Z_SYNC_FLUSH), c.compress(data[len(data) // 3 :]), c.flush()])
                    rc, got = inflate(z, len(data))
                    assert rc == 0 and got == data, (len(data), level, strategy, wbits)
                    assert inflate(z, len(data) + 1)[0] != 0 and (not data or inflate(z, len(data) - 1)[0] != 0)
    data = datasets[-2][:60000]
    z = zlib.compress(data, 6)[2:-4]
    for _ in range(400):  # nothing here may crash or write beyond the buffer it was given
        bad = bytearray(z)
        bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        rc, got = inflate(bytes(bad), len(data))
        assert rc != 0 or len(got) == len(data)
        assert inflate(z[: int(rng.integers(0, len(z)))], len(data))[0] != 0
    # the reader: same records whichever inflater (a fresh process per setting: the switch is read once)
    import subprocess
    import sys

    prog = ("import sys, hashlib; sys.path.insert(0, %r); from remora_amd import io as rio; h = hashlib.sha256();\n"
            "for p in sys.argv[1:]:\n    for r in rio.iter_bam_records(p): h.update(bytes(r.raw))\nprint(h.hexdigest())" % ROOT)
    paths = [os.path.join(DATA, "can_mappings.bam"), os.path.join(DATA, "mod_mappings.bam")]
    digests = [subprocess.run([sys.executable, "-c", prog] + paths, env=dict(os.environ, RMR_FAST_INFLATE=v), capture_output=True, text=True,
                              timeout=300).stdout.strip() for v in ("1", "0")]
    assert digests[0] == digests[1] and len(digests[0]) == 64


def test_native_huffman_bgzf_members_are_valid_deflate(tmp_path):
    """rmr_bgzf_huffman (the writer's encoder at `--bam-level 1`: one dynamic-Huffman block per 0xFF00 bytes of payload, no
    LZ77 matches, stored when that would not shrink) against Python's zlib / gzip as the independent inflater: every member
    has the BGZF header fields, BSIZE, a raw-deflate stream that inflates to ISIZE bytes with the trailer's CRC32, and fits
    64 KiB - for empty, one-byte, constant, two-symbol, uniform (stored), geometric and Fibonacci-skewed payloads (the last
    needs the 15-bit length limit); BamWriter cuts the same members and writes the same stream with and without it."""
    import ctypes
    import gzip
    import struct
    import zlib

    from remora_amd import _lib as L
    from remora_amd import io as rio

    def compress(data, threads):
        n = len(data)
        out = np.empty(max((n + 0xFEFF) // 0xFF00, 1) * 65311, np.uint8)
        out_len = ctypes.c_int64()
        src = np.frombuffer(data, np.uint8) if n else np.zeros(1, np.uint8)
        L.check(L.lib().rmr_bgzf_huffman(src.ctypes.data_as(ctypes.c_void_p), n, threads, out.ctypes.data_as(ctypes.c_void_p), out.size,
                                         ctypes.byref(out_len)))
        return out[: out_len.value].tobytes()

    def members(buf):
        pos, payloads = 0, []
        while pos < len(buf):
            assert buf[pos : pos + 4] == b"\x1f\x8b\x08\x04" and buf[pos + 10 : pos + 16] == b"\x06\x00BC\x02\x00"
            size = struct.unpack("<H", buf[pos + 16 : pos + 18])[0] + 1
            assert size <= 65536
            raw = zlib.decompress(buf[pos + 18 : pos + size - 8], -15)
            crc, isize = struct.unpack("<II", buf[pos + size - 8 : pos + size])
            assert crc == zlib.crc32(raw) & 0xFFFFFFFF and isize == len(raw) <= 0xFF00
            payloads.append(raw)
            pos += size
        return payloads

    rng = np.random.default_rng(1)
    fib = [1, 1]
    while sum(fib) < 60000:
        fib.append(fib[-1] + fib[-2])
    cases = [b"", b"a", b"ab" * 5, bytes(70000), bytes([7]) * 0xFF00, rng.integers(0, 256, 200000, dtype=np.uint8).tobytes(),
             rng.integers(0, 4, 300000, dtype=np.uint8).tobytes(), (rng.geometric(0.02, 400000) % 256).astype(np.uint8).tobytes(),
             b"".join(bytes([i]) * c for i, c in enumerate(fib)), open(os.path.join(DATA, "can_mappings.bam"), "rb").read()]
    for data in cases:
        for threads in (1, 3):
            z = compress(data, threads)
            got = members(z)
            assert b"".join(got) == data and [len(x) for x in got[:-1]] == [0xFF00] * max(len(got) - 1, 0)
        if data:
            assert gzip.decompress(z) == data
    assert len(compress(cases[5], 1)) < len(cases[5]) * 1.002  # incompressible: stored blocks
    assert len(compress(cases[4], 1)) < 8400                   # one symbol: a bit per byte
    # the writer: same members, same stream, whichever encoder
    src = os.path.join(DATA, "can_mappings.bam")
    recs, header, outs = list(rio.iter_bam_records(src)), rio.read_bam_header_bytes(src), {}
    for mode in ("1", "0"):
        os.environ["RMR_BGZF_NATIVE"] = mode
        try:
            path = str(tmp_path / f"o{mode}.bam")
            with rio.BamWriter(path, header, level=1, threads=2) as w:
                for _ in range(90):
                    for r in recs:
                        raw = bytes(r.raw)
                        w.write(struct.pack("<i", len(raw)) + raw)
            outs[mode] = open(path, "rb").read()
        finally:
            os.environ.pop("RMR_BGZF_NATIVE", None)
    a, b = members(outs["1"][:-28]), members(outs["0"][:-28])
    assert a == b and outs["1"][-28:] == outs["0"][-28:] and len(list(rio.iter_bam_records(str(tmp_path / "o1.bam")))) == 90 * len(recs)


def test_bam_writer_splits_a_header_larger_than_one_bgzf_member(tmp_path):
    """A header with thousands of reference sequences (hg38 with alt / decoy contigs) is larger than the 64 KiB a BGZF member
    may hold: every member the writer emits must inflate to at most 65536 bytes (ISIZE), whatever the header's size, and the
    stream must read back whole."""
    import gzip
    import struct

    from remora_amd import io as rio

    text = b"@HD\tVN:1.6\n" + b"".join(b"@SQ\tSN:contig_%05d_alt\tLN:%d\n" % (i, 1000 + i) for i in range(6000))
    refs = b"".join(struct.pack("<i", 17) + b"contig_%05d_alt\0" % i + struct.pack("<i", 1000 + i) for i in range(6000))
    header = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 6000) + refs
    assert len(header) > 3 * 65536
    rec = struct.pack("<iiBBHHHIiii", -1, -1, 3, 0, 4680, 0, 4, 4, -1, -1, 0) + b"r0\0" + bytes([0x12, 0x48]) + b"\xff" * 4
    rec = struct.pack("<i", len(rec)) + rec
    for eof in (True, False):
        path = tmp_path / f"big_header_{eof}.bam"
        with rio.BamWriter(str(path), header, threads=2, eof=eof) as w:
            for _ in range(5):
                w.write(rec)
        blob = open(path, "rb").read()
        off, sizes = 0, []
        while off < len(blob):
            assert blob[off : off + 4] == b"\x1f\x8b\x08\x04" and blob[off + 12 : off + 14] == b"BC"
            bsize = struct.unpack_from("<H", blob, off + 16)[0] + 1
            sizes.append(struct.unpack_from("<I", blob, off + bsize - 4)[0])
            off += bsize
        assert off == len(blob) and max(sizes) <= 65536, max(sizes)
        assert gzip.decompress(blob) == header + 5 * rec


def struct_pack_record(rec):
    import struct

    return struct.pack("<i", len(rec.raw)) + bytes(rec.raw)


_SHARD_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import torch
from remora_amd import dist as rdist
from remora_amd.validate import ValidationLogger, compute_metrics
rank, world, local = rdist.init_process_group("gloo")
rng = np.random.default_rng(5)
n, k = 1001, 3
logits = rng.standard_normal((n, k)) * 2
probs = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
labels = rng.integers(0, k, n)
labels[::3] = probs[::3].argmax(1)                   # a model that is right more often than chance
lo, hi = rdist.shard_range(n, rank, world)
losses = [float(rank + 1), 0.5]
ms = ValidationLogger._global_metrics(probs[lo:hi], labels[lo:hi], losses, 0.1)
acc, conf, ff, facc, fconf, thr = compute_metrics(probs, labels, 0.1)
ok = (ms.acc == acc and np.array_equal(ms.conf_mat, conf) and ms.num_calls == n and abs(ms.filt_frac - ff) < 1e-12
      and ms.filt_acc == facc and np.array_equal(ms.filt_conf_mat, fconf) and ms.filt_thresh == thr
      and abs(ms.loss - np.mean([1.0, 0.5, 2.0, 0.5])) < 1e-12)
objs = rdist.gather_objects({"rank": rank, None: rank + 1})
arr = rdist.gather_arrays(np.arange(rank + 2, dtype=np.int64)[:, None] * np.ones((1, 2), np.int64))
ok = ok and [o["rank"] for o in objs] == [0, 1] and arr.shape == (5, 2) and arr[:, 0].tolist() == [0, 1, 0, 1, 2]
rdist.barrier()
os.write(1, (json.dumps({"rank": rank, "ok": bool(ok), "acc": float(ms.acc), "conf": ms.conf_mat.tolist()}) + "\n").encode())  # one atomic write per rank
torch.distributed.destroy_process_group()
'''


def test_sharded_validation_metrics_gloo_world2(tmp_path):
    """Two ranks each hold half of the calls: ValidationLogger._global_metrics (confusion counts through the all-reduce,
    the filtered columns from the gathered triples) gives every rank exactly compute_metrics of all calls."""
    script = tmp_path / "worker.py"
    script.write_text(_SHARD_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script), ROOT]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.replace("}{", "}\n{").splitlines() if l.startswith("{")]
    assert len(lines) == 2 and all(l["ok"] for l in lines), lines
    assert lines[0]["conf"] == lines[1]["conf"] and lines[0]["acc"] == lines[1]["acc"]


def test_remora_dataset_shard_partitions_rows(tmp_path):
    """RemoraDataset.shard: the ranks' core datasets are contiguous, disjoint row ranges that cover the dataset; label
    counts summed over the shards equal the whole dataset's."""
    from remora_amd.data_chunks import CoreRemoraDataset, RemoraDataset

    dirs = _materialise(tmp_path, ["can_ctrl", "mod_m"])
    for names, props in ((["mod_m"], [1.0]), (["can_ctrl", "mod_m"], [0.5, 0.5])):
        ds = RemoraDataset([CoreRemoraDataset(dirs[n], infinite_iter=False) for n in names], props, batch_size=64)
        whole = ds.get_label_counts()
        for world in (2, 3):
            parts = [ds.shard(r, world) for r in range(world)]
            assert sum(p.size for p in parts) == ds.size
            assert np.array_equal(sum(p.get_label_counts() for p in parts), whole)
            for i in range(len(names)):
                spans = [(p.datasets[i].metadata.dataset_start, p.datasets[i].metadata.dataset_end) for p in parts]
                assert spans[0][0] == ds.datasets[i].metadata.dataset_start and spans[-1][1] == ds.datasets[i].metadata.dataset_end
                assert all(spans[j][1] == spans[j + 1][0] for j in range(world - 1))
            if len(names) == 1:  # one core dataset: the shards' batches visit every row exactly once
                seen = sum(int(b[2].shape[0]) for p in parts for b in p.iter_numpy_batches(("signal", "sequence_lengths", "labels")))
                assert seen == ds.size


def test_native_mm_ml_and_record_rewrite_match_the_python_forms(tmp_path):
    """rmr_format_mm_ml / rmr_records_with_mod_tags (host C++, one call per batch) against util.format_mm_ml_tags and
    io.record_with_mod_tags per read: byte-identical MM strings, ML arrays and output records for random reads - one
    and two modified bases (also a ChEBI code), unsorted and repeated positions, reads without calls, positions on
    non-canonical bases, probabilities of exactly 0 and 1, input records that already carry MM/ML/Mm/Ml tags, records with
    a 'MMZ' byte pattern inside another tag's data."""
    import struct

    from remora_amd import io as rio
    from remora_amd import util

    rng = np.random.default_rng(11)
    md = lambda s: b"MDZ" + s.encode() + b"\x00"  # noqa: E731
    old_mm = b"MMZC+m?,1,2;\x00" + b"MLBC" + struct.pack("<i", 2) + bytes([9, 200])
    old_lower = b"MmZC+h?,0;\x00" + b"MlBC" + struct.pack("<i", 1) + bytes([7])
    trap = b"XXZabMMZcd\x00"  # the header pattern inside another tag's value: the record is walked, nothing is dropped
    recs_spec = []
    seqs = []
    for i in range(40):
        n = int(rng.integers(30, 400))
        seq = "".join(rng.choice(list("ACGT"), n))
        tags = [md(str(n))]
        if i % 4 == 1:
            tags.append(old_mm)
        if i % 4 == 2:
            tags = [old_lower] + tags + [trap]
        if i % 7 == 3:
            tags.append(b"ntC" + bytes([5]))
        recs_spec.append((f"r{i}", 0, seq, [(0, n)], tags))
        seqs.append(seq)
    _tiny_bam(tmp_path / "t.bam", recs_spec)
    recs = list(rio.iter_bam_records(str(tmp_path / "t.bam")))
    assert len(recs) == 40
    for mod_bases, can in ((["m"], "C"), (["h", "m"], "C"), (["27551"], "A")):
        poss, probs, sizes = [], [], []
        for i, seq in enumerate(seqs):
            cand = np.flatnonzero(np.frombuffer(seq.encode(), np.uint8) == ord(can))
            k = 0 if i % 9 == 0 else int(rng.integers(1, max(2, cand.size)))
            p = rng.choice(cand, min(k, cand.size), replace=False) if cand.size and k else np.zeros(0, np.int64)
            if i % 5 == 1 and p.size > 2:
                p[0] = (p[0] + 1) % len(seq)       # a position that is not on a canonical base
                p[1] = p[2]                        # a repeated position
            pr = rng.random((p.size, len(mod_bases))) / len(mod_bases)
            if p.size:
                pr[0, 0] = 0.0
                pr[-1, 0] = 1.0 if len(mod_bases) == 1 else pr[-1, 0]
            poss.append(p.astype(np.int64)); probs.append(pr); sizes.append(p.size)
        seq_off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])])
        call_off = np.concatenate([[0], np.cumsum(sizes)])
        mm, mm_off, ml, ml_off = util.format_mm_ml_tags_batch("".join(seqs).encode(), seq_off, np.concatenate(poss),
                                                              np.concatenate(probs), call_off, mod_bases, can)
        want_recs = []
        for i, seq in enumerate(seqs):
            w_mm, w_ml = util.format_mm_ml_tags(seq, poss[i], probs[i], mod_bases, can)
            assert mm[mm_off[i] : mm_off[i + 1]].tobytes().decode() == w_mm, (mod_bases, i)
            assert ml[ml_off[i] : ml_off[i + 1]].tobytes() == bytes(w_ml), (mod_bases, i)
            want_recs.append(rio.record_with_mod_tags(recs[i], w_mm, w_ml) if sizes[i] else rio.record_with_mod_tags(recs[i], None, None))
        got = rio.records_with_mod_tags_batch(recs, mm, mm_off, ml, ml_off, np.asarray(sizes) > 0)
        assert got == b"".join(want_recs), mod_bases


def test_forced_single_rank_process_group_gloo():
    """REMORA_AMD_DIST_SINGLE=1 (what tests/test_gpu_bench.py uses to send every dist.py helper through RCCL on a 1-GPU
    box): a world of one rank builds its process group and the helpers go through the backend instead of short-cutting."""
    code = ("import numpy as np\nfrom remora_amd import dist as rdist\nimport torch.distributed as dist\n"
            "rdist.init_process_group('gloo', timeout_s=60)\n"
            "assert dist.is_initialized() and dist.get_world_size() == 1 and rdist._collective()\n"
            "assert rdist.first_collective_ms() > 0\n"
            "assert rdist.allreduce_counts(np.array([3, 4], np.int64)).tolist() == [3, 4]\n"
            "assert rdist.gather_objects('x') == ['x'] and rdist.allgather_floats([1.0]).tolist() == [[1.0]]\n"
            "assert rdist.gather_arrays(np.arange(3)).tolist() == [0, 1, 2]\n"
            "rdist.barrier()\ndist.destroy_process_group()\nprint('OK')\n")
    env = dict(os.environ, REMORA_AMD_DIST_SINGLE="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.pop("MASTER_PORT", None)
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "OK" in p.stdout, p.stderr[-2000:]
    # without the switch a single process never touches torch.distributed
    from remora_amd import dist as rdist

    assert not rdist._single_forced() and not rdist._collective() and rdist.first_collective_ms() == 0.0


def test_launch_ranks_starts_the_ranks_itself_and_stops_them_together():
    """dist.launch_ranks: N processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment (no
    torch.distributed.run in between) that can build a process group and reduce; a rank that fails ends the launch with its
    exit code and takes the waiting ranks with it."""
    import time

    from remora_amd import dist as rdist

    ok = ("import os, sys\nimport numpy as np\nfrom remora_amd import dist as rdist\n"
          "rank, world, local = rdist.init_process_group('gloo', timeout_s=60)\n"
          "got = rdist.allreduce_counts(np.array([rank + 1, 10], np.int64)).tolist()\n"
          "assert world == 3 and local == rank and got == [6, 30], (rank, world, local, got)\n"
          "assert os.environ['LOCAL_WORLD_SIZE'] == '3' and sys.argv[1:] == ['--flag', 'x']\n")
    assert rdist.launch_ranks(["--flag", "x"], 3, command=[sys.executable, "-c", ok]) == 0
    bad = "import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(120)\n"
    t0 = time.monotonic()
    assert rdist.launch_ranks([], 3, command=[sys.executable, "-c", bad]) == 7
    assert time.monotonic() - t0 < 30


def test_bam_guess_start_on_random_records(tmp_path):
    """rmr_bam_guess_start against records built to confuse it: random bytes in qualities, sequences and B-array tags
    (every byte pattern, BGZF magic and plausible block_size fields included), names of 1..254 characters, unmapped and
    zero-length records, records of 40 B to 300 KB spanning many BGZF members, every tag type.  At every probed offset
    the guess is the true first record start behind it, and byte-range shares partition the file for any worker count."""
    import random
    import struct

    from remora_amd import io as rio

    rng = random.Random(20260927)
    src = os.path.join(DATA, "can_mappings.bam")
    header = rio.read_bam_header_bytes(src)
    n_ref = struct.unpack_from("<i", header, 8 + struct.unpack_from("<i", header, 4)[0])[0]
    assert n_ref >= 1

    def record(i):
        name = "".join(chr(rng.randrange(33, 127)) for _ in range(rng.choice([1, 3, 36, 36, 36, 120, 253]))).encode() + b"\x00"
        l_seq = rng.choice([0, 0, 1, 7, 300, 5000, 60000]) if i % 7 else rng.randrange(0, 150000)
        n_cig = rng.choice([0, 1, 3, 40])
        unmapped = rng.random() < 0.2
        ref_id, pos = (-1, -1) if unmapped else (rng.randrange(n_ref), rng.randrange(0, 1 << 28))
        cigar = b"".join(struct.pack("<I", (rng.randrange(1, 1 << 20) << 4) | rng.randrange(9)) for _ in range(n_cig))
        seq, qual = rng.randbytes((l_seq + 1) // 2), rng.randbytes(l_seq)
        tags = b""
        for _ in range(rng.randrange(0, 7)):
            tag = bytes([rng.choice(b"ABXYZmnpq"), rng.choice(b"ABXYZmnpq0123")])
            ty = rng.choice("AcCsSiIfZHB")
            if ty in "AcC":
                tags += tag + ty.encode() + rng.randbytes(1)
            elif ty in "sS":
                tags += tag + ty.encode() + rng.randbytes(2)
            elif ty in "iIf":
                tags += tag + ty.encode() + rng.randbytes(4)
            elif ty in "ZH":
                tags += tag + ty.encode() + bytes(rng.randrange(1, 256) for _ in range(rng.randrange(0, 200))) + b"\x00"
            else:
                sub = rng.choice("cCsSiIf")
                cnt = rng.choice([0, 1, 50, 3000, 70000])
                tags += tag + b"B" + sub.encode() + struct.pack("<i", cnt) + rng.randbytes(cnt * {"c": 1, "C": 1, "s": 2, "S": 2}.get(sub, 4))
        if rng.random() < 0.3:  # a decoy: what a record header looks like, inside a B array
            decoy = struct.pack("<iiiBBHHHiiii", 60, 0, 5, 4, 0, 0, 0, 0, 10, -1, -1, 0) + b"abc\x00" + bytes(30)
            tags += b"dcBC" + struct.pack("<i", len(decoy)) + decoy
        body = struct.pack("<iiBBHHHiiii", ref_id, pos, len(name), rng.randrange(256), rng.randrange(65536), n_cig, rng.randrange(4096),
                           l_seq, -1 if unmapped else rng.randrange(-1, n_ref), rng.randrange(-1, 1 << 20), rng.randrange(-1000, 1000))
        body += name + cigar + seq + qual + tags
        return struct.pack("<i", len(body)) + body

    path = str(tmp_path / "random.bam")
    with rio.BamWriter(path, header, level=1) as w:
        for i in range(700):
            w.write(record(i))
    whole = [(r.query_name, r.voffset) for r in rio.iter_bam_records(path)]
    assert len(whole) == 700
    vos, size = [v for _, v in whole], os.path.getsize(path)
    for off in [rng.randrange(size) for _ in range(400)]:
        want = next((v for v in vos if (v >> 16) >= off), None) if off > (vos[0] >> 16) else vos[0]
        assert rio.bam_guess_start(path, off) == want, off
    for world in (2, 7, 32):
        got = []
        for rank in range(world):
            got += [(r.query_name, r.voffset) for r in rio.iter_bam_records(path, shard=(rank, world))]
        assert got == whole, world


def test_launch_ranks_scans_for_its_ranks_in_scan_mode(tmp_path):
    """The exact-pass form end to end on CPU: dist.launch_ranks(scan_bam=...) scans while its ranks start, the ranks'
    io.bam_shard takes the marks from the announced file (not from a scan of their own) and the shares partition the
    file; the scan file is gone afterwards."""
    from remora_amd import dist as rdist
    from remora_amd import io as rio

    path = os.path.join(DATA, "mod_mappings.bam")
    out = str(tmp_path / "share")
    code = ("import os, sys, json\nfrom remora_amd import io as rio\n"
            "calls = []\nreal = rio.bam_scan\nrio.bam_scan = lambda *a, **k: (calls.append(1), real(*a, **k))[1]\n"
            f"rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
            f"st, n = rio.bam_shard({path!r}, rank, world)\n"
            f"names = [r.query_name for r in rio._iter_bam_records_native({path!r}, False, 4, start_voffset=st, max_records=n)] if n else []\n"
            f"json.dump({{'names': names, 'own_scans': len(calls), 'scan_file': os.environ.get(rio.SCAN_ENV)}}, open({out!r} + os.environ['RANK'], 'w'))\n")
    assert rdist.launch_ranks([], 3, scan_bam=path, command=[sys.executable, "-c", code]) == 0
    got = [json.load(open(out + str(r))) for r in range(3)]
    assert sum((g["names"] for g in got), []) == [r.query_name for r in rio.iter_bam_records(path)]
    assert all(g["own_scans"] == 0 and g["scan_file"] for g in got) and not os.path.exists(got[0]["scan_file"])


@pytest.mark.parametrize("workers", [1, 2, 5, 9])
def test_bam_shares_from_a_c_program(tmp_path, workers):
    """tests/c/bam_shares_from_c.c: the byte-range split of a BAM through the C ABI alone (rmr_bam_guess_start +
    rmr_bam_read_batch, pedantic C99, no Python, no GPU): the workers' shares meet end to start and hold exactly the
    file's records."""
    import shutil
    import struct
    import subprocess

    from remora_amd import _lib
    from remora_amd import io as rio

    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    exe, libdir = str(tmp_path / "bam_shares_from_c"), os.path.dirname(_lib.LIB_PATH)
    cc = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                         os.path.join(ROOT, "tests", "c", "bam_shares_from_c.c"), "-o", exe, "-L", libdir, "-lremora_hip",
                         f"-Wl,-rpath,{libdir}"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    big = str(tmp_path / "big.bam")
    recs = list(rio.iter_bam_records(os.path.join(DATA, "can_mappings.bam")))
    with rio.BamWriter(big, rio.read_bam_header_bytes(os.path.join(DATA, "can_mappings.bam")), level=1) as w:
        for _ in range(20):
            for r in recs:
                raw = bytes(r.raw)
                w.write(struct.pack("<i", len(raw)) + raw)
    run = subprocess.run([exe, big, str(workers)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, (run.stdout, run.stderr)
    assert f"{20 * len(recs)} records in the file, {20 * len(recs)} in the shares" in run.stdout
    assert len([ln for ln in run.stdout.splitlines() if ln.startswith("worker ")]) == workers


# ---- rank placement (dist.plan_rank_binding / bind_rank): a rank on the socket of its GPU, from a sysfs tree ----
def _fake_sysfs(root, n_gpus=8):
    """A two-socket node as sysfs shows it: GPUs 0-3 on node 0, 4-7 on node 1; 256 hardware threads, the second hyperthread
    of a core 128 above the first (node0 = 0-63,128-191; node1 = 64-127,192-255)."""
    addrs = []
    for g in range(n_gpus):
        addr = f"0000:{0x05 + 0x10 * g:02x}:00.0"
        d = root / "bus" / "pci" / "devices" / addr
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{g // 4}\n")
        addrs.append(addr)
    for k, cl in enumerate(("0-63,128-191", "64-127,192-255")):
        d = root / "devices" / "system" / "node" / f"node{k}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cl + "\n")
    return addrs


def test_rank_binding_plan_for_8_gpus_x_1_and_x_6(tmp_path):
    from remora_amd import dist as rdist

    addrs = _fake_sysfs(tmp_path)
    node_cpus = [set(range(0, 64)) | set(range(128, 192)), set(range(64, 128)) | set(range(192, 256))]
    # one process per GPU, every core allowed: rank r -> GPU r -> node r // 4, all 128 threads of that socket, 8 helpers
    plan = rdist.plan_rank_binding(addrs, 1, allowed_cpus=range(256), sysfs=str(tmp_path))
    assert [p["device"] for p in plan] == list(range(8)) and [p["numa_node"] for p in plan] == [0] * 4 + [1] * 4
    assert all(set(p["cpus"]) == node_cpus[p["numa_node"]] for p in plan) and all(p["threads"] == 8 for p in plan)
    # six processes per GPU: 48 ranks, device = rank // 6, 24 ranks per socket share its 128 threads -> 5 helpers each
    plan = rdist.plan_rank_binding(addrs, 6, allowed_cpus=range(256), sysfs=str(tmp_path))
    assert len(plan) == 48 and [p["device"] for p in plan] == [r // 6 for r in range(48)]
    assert [p["numa_node"] for p in plan] == [0] * 24 + [1] * 24 and all(p["threads"] == 5 for p in plan)
    assert all(set(p["cpus"]) == node_cpus[p["numa_node"]] for p in plan)
    # a CPU quota of 16 cores (the GPU boxes of this pool): binding as before, helpers = the quota's share, floor 2
    plan = rdist.plan_rank_binding(addrs, 1, allowed_cpus=range(256), sysfs=str(tmp_path), cpu_budget=16)
    assert [p["numa_node"] for p in plan] == [0] * 4 + [1] * 4 and all(p["threads"] == 2 for p in plan)
    # a cpuset that only holds cores of node 0: node-1 ranks are not bound to cores they may not use
    plan = rdist.plan_rank_binding(addrs, 1, allowed_cpus=range(16), sysfs=str(tmp_path))
    assert [p["numa_node"] for p in plan] == [0] * 4 + [-1] * 4 and all(p["cpus"] == list(range(16)) for p in plan)
    # no topology in sysfs (single socket, VM): nobody is bound, every rank sees every allowed core
    plan = rdist.plan_rank_binding([None] * 8, 1, allowed_cpus=range(32), sysfs=str(tmp_path))
    assert all(p["numa_node"] == -1 and p["cpus"] == list(range(32)) and p["threads"] == 4 for p in plan)
    assert rdist.numa_of_pci("0000:ff:00.0", str(tmp_path)) == (-1, [])


def test_bind_rank_applies_the_plan_to_the_process(tmp_path):
    """bind_rank in a child process (its affinity must not leak into the test runner): the node of the rank's GPU holds half
    of the cores the child may use; afterwards the child - and a thread it starts - may run on exactly those."""
    import subprocess

    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("needs two usable cores")
    half = allowed[: len(allowed) // 2]
    for g in range(2):
        d = tmp_path / "bus" / "pci" / "devices" / f"0000:0{g}:00.0"
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{g}\n")
    for k, cpus in enumerate((half, allowed[len(allowed) // 2:])):
        d = tmp_path / "devices" / "system" / "node" / f"node{k}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    code = (
        "import os, sys, json, threading; sys.path.insert(0, %r)\n"
        "from remora_amd import dist as rdist\n"
        "b = rdist.bind_rank(0, local_rank=0, local_world=2, sysfs=%r, gpu_addrs=['0000:00:00.0', '0000:01:00.0'])\n"
        "seen = []\n"
        "t = threading.Thread(target=lambda: seen.append(sorted(os.sched_getaffinity(0)))); t.start(); t.join()\n"
        "print(json.dumps(dict(b=b, mine=sorted(os.sched_getaffinity(0)), thread=seen[0], omp=os.environ['OMP_NUM_THREADS'],"
        " pack=os.environ['RMR_PACK_THREADS'])))\n" % (ROOT, str(tmp_path)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got["b"]["bound"] and got["b"]["numa_node"] == 0 and got["mine"] == half and got["thread"] == half
    assert got["omp"] == got["pack"] == str(got["b"]["threads"]) and 2 <= got["b"]["threads"] <= 8
    # switched off: nothing is touched
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, REMORA_AMD_RANK_BINDING="0", OMP_NUM_THREADS="3", RMR_PACK_THREADS="3"))
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert not got["b"]["bound"] and got["mine"] == allowed and got["omp"] == "3"


def test_collect_reads_c_api_walk_equals_the_interpreters(monkeypatch):
    """data_chunks._collect_reads: the C-API walk over a batch of RemoraRead objects (csrc/pyglue.c) hands rmr_pack_reads
    the same addresses, sizes and scalars as the interpreter's walk; reads it does not take as they are (float dacs, strided
    arrays, int32 mappings) fall to the interpreter, which converts them, and an inconsistent read is refused with the
    reference's message either way."""
    from remora_amd import RemoraError, synth
    from remora_amd.data_chunks import RemoraRead, _collect_reads

    def mk(i, nb, seq_dtype=np.int64):
        r = synth.synth_read(nb, idx=i)
        return RemoraRead(dacs=r["dacs"], shift=np.float32(r["shift"] + i), scale=r["scale"] - i, seq_to_sig_map=r["seq_to_sig_map"],
                          int_seq=r["int_seq"].astype(seq_dtype), read_id=f"r{i}")

    reads = [mk(0, 300), mk(1, 7, np.int8), mk(2, 1200, np.int32), mk(3, 50, np.uint8)]
    monkeypatch.setenv("RMR_PY_GLUE", "1")
    def by_interpreter(res):  # the interpreter's walk keeps (dacs, mapping, bases) tuples, the C walk the attribute objects
        return bool(res[-1]) and isinstance(res[-1][0], tuple)

    a = _collect_reads(reads)
    assert not by_interpreter(a), "the C-API walk must take these reads as they are"
    assert len(a[-1]) == 3 * len(reads) and a[-1][0] is reads[0].dacs and a[-1][5] is reads[1].int_seq  # owned until the gather is done
    monkeypatch.setenv("RMR_PY_GLUE", "0")
    b = _collect_reads(reads)
    assert by_interpreter(b)
    for x, y in zip(a[:-1], b[:-1]):
        assert x.dtype == y.dtype and np.array_equal(x, y)
    assert list(a[0]) == [r.dacs.ctypes.data for r in reads] and list(a[5]) == [8, 1, 4, 1]
    assert a[6][2] == float(reads[2].shift) and a[7][3] == float(reads[3].scale)
    # what the C walk leaves to the interpreter
    monkeypatch.setenv("RMR_PY_GLUE", "1")
    odd = [mk(0, 300), RemoraRead.test_read()]                       # float zeros as dacs
    assert by_interpreter(_collect_reads(odd))
    strided = mk(4, 200)
    strided.dacs = np.repeat(strided.dacs, 2)[::2]                      # a view with a stride
    assert not strided.dacs.flags.c_contiguous
    got = _collect_reads([strided])
    assert by_interpreter(got) and np.array_equal(got[-1][0][0], strided.dacs)
    m32 = mk(5, 100)
    m32.seq_to_sig_map = m32.seq_to_sig_map.astype(np.int32)
    got = _collect_reads([m32])
    assert by_interpreter(got) and got[-1][0][1].dtype == np.int64
    bad = mk(6, 100)
    bad.seq_to_sig_map = bad.seq_to_sig_map[:-1]
    with pytest.raises(RemoraError, match="sizes incompatible"):
        _collect_reads([reads[0], bad])
    assert _collect_reads([])[1].size == 0

    # duck-typed reads whose arrays are computed per access (properties): nothing but the returned `keep` owns them, and the
    # addresses must still hold the read's values after other allocations have had their chance to reuse freed memory
    class Lazy:
        def __init__(self, i):
            self._r = synth.synth_read(400 + i, idx=i)
            self.shift, self.scale = 1.0 + i, 2.0

        dacs = property(lambda self: self._r["dacs"].copy())
        seq_to_sig_map = property(lambda self: self._r["seq_to_sig_map"].astype(np.int64))
        int_seq = property(lambda self: self._r["int_seq"].astype(np.int64))

    import ctypes
    import gc

    lazy = [Lazy(i) for i in range(64)]
    got = _collect_reads(lazy)
    assert not by_interpreter(got) and len(got[-1]) == 3 * 64
    gc.collect()
    junk = [np.full(400 + i, -7, np.int16) for i in range(64)] + [np.full(401 + i, -7, np.int64) for i in range(128)]  # noqa: F841
    for i, r in enumerate(lazy):
        d = np.ctypeslib.as_array(ctypes.cast(int(got[0][i]), ctypes.POINTER(ctypes.c_int16)), (int(got[1][i]),))
        m = np.ctypeslib.as_array(ctypes.cast(int(got[2][i]), ctypes.POINTER(ctypes.c_int64)), (int(got[4][i]) + 1,))
        assert np.array_equal(d, r._r["dacs"]) and np.array_equal(m, r._r["seq_to_sig_map"])


def test_chunk_geometry_searches_from_a_hint_equal_bisection(tmp_path):
    """rmr_geometry.h is ONE function for the geometry kernel and for the host side of rmr_call_read; the host calls it with
    searches that gallop outwards from the focus base (a handful of cache-local probes instead of log2(n_bases) scattered
    ones).  tests/c/geometry_searches.cpp: 1.3 M random searches and 150 k whole rows - both forms agree everywhere."""
    import shutil

    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the image"
    (tmp_path / "hip").mkdir()
    (tmp_path / "hip" / "hip_runtime.h").write_text("#pragma once\n#define __host__\n#define __device__\n")
    exe = str(tmp_path / "geo")
    cc = subprocess.run([gxx, "-O2", "-std=c++17", "-Wall", "-I", str(tmp_path), "-I", os.path.join(ROOT, "remora_amd", "csrc"),
                         os.path.join(ROOT, "tests", "c", "geometry_searches.cpp"), "-o", exe], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.strip().endswith("0 mismatches"), run.stdout + run.stderr


def test_subbatch_cuts_partition_a_batch_of_reads():
    """inference._subbatch_cuts: contiguous, complete, non-empty pieces for any batch size; a short start (the GPU gets work
    after a quarter sub-batch has been staged) and a short end (what remains to be done when the stager is finished)."""
    from remora_amd.inference import _subbatch_cuts

    for sub in (512, 64, 7, 1):
        for n in list(range(1, 70)) + [511, 512, 513, 1024, 1100, 2048, 5000, 100_003]:
            cuts = _subbatch_cuts(n, sub)
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:])) and all(b > a for a, b in cuts)
            assert max(b - a for a, b in cuts) <= max(sub, 1) + sub // 2
    assert [b - a for a, b in _subbatch_cuts(2048, 512)] == [128, 256, 512, 512, 384, 256]
    assert [b - a for a, b in _subbatch_cuts(2048, 512, (64, 128, 256))] == [64, 128, 256, 512, 512, 346, 230]  # the 16-bit models' lead
    for n in (0, 1, 63, 64, 65, 500, 2048, 5000):
        cuts = _subbatch_cuts(n, 512, (64, 128, 256))
        assert [a for a, _ in cuts] == [0] * bool(n) + [b for _, b in cuts][:-1] and (not n or cuts[-1][1] == n) and all(b > a for a, b in cuts)


def test_built_library_has_no_packed_fp32_op_sel_on_lds_fed_registers():
    """tools/lint_pk_lds.py on the built library: no v_pk_fma / mul / add_f32 whose op_sel / op_sel_hi modifier re-selects the
    halves of an operand that an LDS read wrote - the instruction pattern tools/ubench/pk_lds_repro.hip shows returning wrong
    values in lanes 32-63 beside 16-bit MFMAs on gfx950 (profiles/NOTES_r05.md section 2).  hipcc's SLP vectoriser used to
    produce it from scalar code in fused_front_kernel, seq1_dense_kernel and refine_dp_kernel (95 instructions); the library is
    built with -fno-slp-vectorize since."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lint_pk_lds

    if not os.path.exists(lint_pk_lds.OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    for name in ("libremora_hip.so", "libremora_hip_jitter.so"):
        lib = os.path.join(ROOT, "remora_amd", name)
        if not os.path.exists(lib):
            assert name != "libremora_hip.so"
            continue
        res = lint_pk_lds.lint(lib)
        assert res["kernels"] > 50 and res["packed_f32"] > 500, res  # the walk saw the kernels and their packed math
        assert res["flagged"] == [], res["flagged"][:5]


def test_ref_anchor_batch_equals_move_table_then_ref_to_signal_per_record():
    """rmr_ref_anchor_batch (the reference-anchored half of a BAM batch's ingest: io.parse_move_tag, then compute_ref_to_signal,
    per record on native threads) against the per-record forms on the reference's own aligned reads (both test BAMs, both
    strands) and on their move tables made discordant; the statuses are the per-read path's errors in its order."""
    import ctypes

    from remora_amd import _lib as L
    from remora_amd import io as rio

    pp = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    checked = 0
    for name in ("can_mappings.bam", "mod_mappings.bam"):
        rb = next(iter(rio.iter_bam_raw_batches(os.path.join(DATA, name), want_ref=True, batch=64)))[0]
        n = rb.n
        mv_len = np.diff(rb.mv_off)
        seq_len = np.diff(rb.seq_off).astype(np.int64)
        stride = np.array([int(rb.mv[rb.mv_off[i]]) if mv_len[i] else 1 for i in range(n)])
        sig_len = ((mv_len - 1) * stride + 3).astype(np.int64)  # any length with sig_len // stride == entries
        rev = np.ascontiguousarray((rb.flag & 16) != 0, np.uint8)
        ref_len = np.where((rb.ref_ok != 0) & (rb.ref_id >= 0), np.diff(rb.refseq_off), -1).astype(np.int64)
        assert (ref_len > 0).sum() >= 10
        for variant in ("as is", "one base short", "signal short", "no reference", "threads"):
            sl, ql, rl = sig_len.copy(), seq_len.copy(), ref_len.copy()
            if variant == "one base short":
                ql[::2] -= 1
            if variant == "signal short":
                sl[1::3] -= 40
            if variant == "no reference":
                rl[::4] = -1
            r2s_off = np.zeros(n + 1, np.int64)
            np.cumsum(np.maximum(rl, -1) + 1, out=r2s_off[1:])
            r2s = np.full(int(r2s_off[-1]) + 1, -7, np.int64)
            status = np.full(n, 77, np.int32)
            L.check(L.lib().rmr_ref_anchor_batch(n, pp(rb.mv), pp(np.ascontiguousarray(rb.mv_off, np.int64)), pp(sl), pp(ql),
                                                 pp(np.ascontiguousarray(rb.cigar, np.uint32)), pp(np.ascontiguousarray(rb.cigar_off, np.int64)),
                                                 pp(rev), pp(rl), pp(r2s), pp(r2s_off), pp(status), 5 if variant == "threads" else 1))
            for i in range(n):
                mv = rb.mv[rb.mv_off[i] : rb.mv_off[i + 1]]
                if mv.size == 0:
                    assert status[i] == 8
                    continue
                q2s = np.concatenate([np.nonzero(mv[1:])[0] * int(mv[0]), [sl[i]]]).astype(np.int64)  # io.py:397-400
                if q2s.size - 1 != ql[i]:
                    assert status[i] == L.ERR_DISCORDANT_SEQ, (variant, i)
                elif mv.size - 1 != sl[i] // int(mv[0]):
                    assert status[i] == L.ERR_DISCORDANT_SIG, (variant, i)
                elif rl[i] < 0:
                    assert status[i] == 9, (variant, i)
                else:
                    cig = rb.cigar[rb.cigar_off[i] : rb.cigar_off[i + 1]]
                    want = rio._ref_to_signal_of_bam_cigar(cig, bool(rev[i]), q2s, int(rl[i]) + 1)
                    assert status[i] == (0 if want.size == rl[i] + 1 else 1), (variant, i, int(status[i]))
                    if status[i] == 0:
                        assert np.array_equal(r2s[r2s_off[i] : r2s_off[i + 1]], want), (variant, i)
                        checked += 1
            assert r2s[-1] == -7
    assert checked >= 60
    # the reference's own error texts for alignments without a usable CIGAR, and a declared reference length that is not the CIGAR's
    mv = np.array([5, 1, 0, 1, 1, 0, 1], np.int8)
    for cig, rl, want in (([(4 << 4) | 4], 4, 3), ([(4 << 4) | 9], 4, 2), ([(4 << 4) | 0], 7, 1), ([(4 << 4) | 0], 4, 0)):
        status, r2s = np.zeros(1, np.int32), np.zeros(16, np.int64)
        L.check(L.lib().rmr_ref_anchor_batch(1, pp(mv), pp(np.array([0, 7], np.int64)), pp(np.array([31], np.int64)), pp(np.array([4], np.int64)),
                                             pp(np.array(cig, np.uint32)), pp(np.array([0, 1], np.int64)), pp(np.zeros(1, np.uint8)),
                                             pp(np.array([rl], np.int64)), pp(r2s), pp(np.array([0, rl + 1], np.int64)), pp(status), 1))
        assert status[0] == want, (cig, rl, int(status[0]))
    assert r2s[:5].tolist() == [0, 10, 15, 25, 31]
    assert rio._REF_ANCHOR_ERRORS == {1: "Discordant ref seq lengths", 2: "Invalid cigar op(s)", 3: "No match operations found in alignment cigar"}


def test_reference_anchored_records_in_one_native_call(tmp_path):
    """rmr_records_with_mod_tags_ref (io.records_with_mod_tags_flat with reference sequences) against record_with_mod_tags per
    record: a record that gets tags and owns reference bases leaves as `<len>M` + those bases + 0xff qualities
    (src/remora/inference.py:452-458), every other record as before; odd and even lengths, letters outside ACGT."""
    from remora_amd import io as rio

    rb, records = next(iter(rio.iter_bam_raw_batches(os.path.join(DATA, "can_mappings.bam"), want_ref=True, batch=64)))
    recs = records(rb)
    rng = np.random.default_rng(4)
    n = rb.n
    refs, has, mms, mls = [], np.zeros(n, np.uint8), [], []
    for i in range(n):
        k = [0, 1, 2, 7, 8, 501][i % 6]
        refs.append("".join(rng.choice(list("ACGTNRacgt"), k)) if i % 5 else "")
        has[i] = i % 3 != 1
        mms.append(f"C+m?,{i},{2 * i};" if has[i] else "")
        mls.append(rng.integers(0, 256, 2 if has[i] else 0).astype(np.uint8))
    mm = np.frombuffer("".join(mms).encode(), np.uint8)
    mm_off = np.concatenate([[0], np.cumsum([len(x) for x in mms])]).astype(np.int64)
    ml = np.concatenate(mls) if n else np.zeros(0, np.uint8)
    ml_off = np.concatenate([[0], np.cumsum([x.size for x in mls])]).astype(np.int64)
    ref_off = np.concatenate([[0], np.cumsum([len(x) for x in refs])]).astype(np.int64)
    got = rio.records_with_mod_tags_flat(rb.raw, rb.raw_off[:-1], np.diff(rb.raw_off), rb.tags_off, mm if mm.size else np.zeros(1, np.uint8),
                                         mm_off, ml if ml.size else np.zeros(1, np.uint8), ml_off, has, ref_seq="".join(refs).encode(),
                                         ref_off=ref_off)
    want = b"".join(rio.record_with_mod_tags(recs[i], mms[i] if has[i] else None, mls[i] if has[i] else None,
                                             ref_anchored_seq=(refs[i] if has[i] and refs[i] else None)) for i in range(n))
    assert got == want
    # ... and the parsed result is a record of that length with one M operation
    rec0 = next(i for i in range(n) if has[i] and len(refs[i]) > 2)
    one = rio.records_with_mod_tags_flat(rb.raw, rb.raw_off[rec0 : rec0 + 1], np.diff(rb.raw_off)[rec0 : rec0 + 1], rb.tags_off[rec0 : rec0 + 1],
                                         mm, mm_off[rec0 : rec0 + 2] - 0, ml, ml_off[rec0 : rec0 + 2], has[rec0 : rec0 + 1],
                                         ref_seq="".join(refs).encode(), ref_off=ref_off[rec0 : rec0 + 2])
    path = tmp_path / "one.bam"
    with rio.BamWriter(str(path), rio.read_bam_header_bytes(os.path.join(DATA, "can_mappings.bam"))) as w:
        w.write(one)
    back = list(rio.iter_bam_records(str(path)))
    assert len(back) == 1 and back[0].cigartuples == [(0, len(refs[rec0]))] and len(back[0].query_sequence) == len(refs[rec0])


def test_native_focus_bases_follow_the_interpreters_set_order():
    """csrc/pyset_order.c restates CPython's set (insertion, growth, iteration order) on raw arrays so that `dataset prepare` can
    keep the reference's chunk order (src/remora/util.py:413-426 iterates a python set) without a set per read; here against
    the interpreter's own set: plain key lists across the table sizes, and motif hits of ragged batches (empty reads, N bases,
    several motifs whose hits overlap, one to five native threads)."""
    import ctypes

    from remora_amd import data_chunks as dc, util

    g = dc._set_order_glue()
    assert g, "the restatement must match this interpreter (otherwise prepare silently takes the slow path)"
    rng = np.random.RandomState(11)
    for n, top in ((0, 5), (1, 5), (4, 9), (5, 9), (6, 64), (19, 64), (20, 4096), (77, 100), (400, 1 << 30), (12000, 13000), (120000, 1 << 40)):
        keys = np.ascontiguousarray(rng.randint(0, top, n), np.int64)
        out = np.empty(max(n, 1), np.int64)
        cnt = g.rmr_py_set_order(keys.ctypes.data, n, out.ctypes.data)
        s = set()
        s.update(keys.tolist())
        assert out[:cnt].tolist() == list(s), (n, top)
    motif_sets = [[util.Motif("CG", 0)], [util.Motif("C", 0)], [util.Motif("CG", 0), util.Motif("CHH", 0), util.Motif("CHG", 0)],
                  [util.Motif("N", 0)], [util.Motif("DRACH", 2), util.Motif("A", 0)], [util.Motif("GATC", 1), util.Motif("CCWGG", 1)],
                  [util.Motif("ACGTACGTACGTACGT", 15), util.Motif("T", 0)]]
    for trial in range(70):
        n_reads = int(rng.randint(1, 40))
        lens = rng.randint(0, 2500, n_reads) if trial % 3 else rng.randint(0, 12, n_reads)
        off = np.concatenate([[0], np.cumsum(lens)])
        iseq = rng.randint(-1 if trial % 5 == 0 else 0, 4, off[-1]).astype(np.int8)
        mots = motif_sets[trial % len(motif_sets)]
        focus, foc_off = dc.focus_bases_set_order(iseq, off, mots, threads=1 + trial % 5)
        assert foc_off[0] == 0 and foc_off[-1] == focus.size
        for k in range(n_reads):
            want = util.find_focus_bases_in_int_sequence(iseq[off[k] : off[k + 1]].astype(np.int64), mots)
            assert np.array_equal(want, focus[foc_off[k] : foc_off[k + 1]]), (trial, k)
    # a focus position in front of the motif (leading N stripped): keys could be negative -> not restated, the caller keeps the interpreter
    assert dc.focus_bases_set_order(np.zeros(8, np.int8), np.array([0, 8]), [util.Motif("NCG", 0)]) is None
    # the down-sampling draw of prepare's batch path is numpy's legacy choice(replace=False), value and generator state
    a = np.arange(1000, 1400) * 3
    np.random.seed(5)
    x = np.random.choice(a, size=15, replace=False)
    after = np.random.random()
    np.random.seed(5)
    y = a[np.random.permutation(a.size)[:15]]
    assert np.array_equal(x, y) and after == np.random.random()


def test_orient_bases_native_against_the_interpreters_string_operations():
    """rmr_orient_bases (strand-aware bases + integer codes of a batch of records in one native pass) against what the per-read
    path does with bytes.upper / translate / [::-1] and util.seq_to_int's table (src/remora/io.py:2023, :2058-2060;
    src/remora/util.py:131-142): ragged and empty records, lower-case mismatch marks, ambiguity codes, one and several threads."""
    import ctypes

    from remora_amd import _lib as L, io as rio
    from remora_amd.util import _SEQ_TRANS

    lib = L.lib()
    rng = np.random.RandomState(3)
    letters = np.frombuffer(b"ACGTacgtNnRYKMBVDH=", np.uint8)
    for trial in range(12):
        n_rec = int(rng.randint(1, 60))
        lens = rng.randint(0, 400, n_rec)
        lens[rng.randint(0, n_rec)] = 0
        blob = letters[rng.randint(0, letters.size if trial % 2 else 4, int(lens.sum()) + 7)].tobytes()
        off = np.concatenate([[0], np.cumsum(lens)])
        pick = np.sort(rng.choice(n_rec, size=max(1, n_rec // 2), replace=False))
        start, ln = np.ascontiguousarray(off[pick], np.int64), np.ascontiguousarray(lens[pick], np.int64)
        rev = np.ascontiguousarray(rng.randint(0, 2, pick.size), np.uint8)
        for upper in (0, 1):
            total = int(ln.sum())
            fwd, ori, codes = (np.zeros(total + 1, np.uint8), np.zeros(total + 1, np.uint8), np.zeros(total + 1, np.int8))
            p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
            L.check(lib.rmr_orient_bases(blob, p(start), p(ln), p(rev), pick.size, upper, rio._COMP_BYTES, _SEQ_TRANS, p(fwd), p(ori), p(codes),
                                         1 + trial % 4))
            want_f, want_o = [], []
            for s0, m, r in zip(start.tolist(), ln.tolist(), rev.tolist()):
                f = blob[s0 : s0 + m].upper() if upper else blob[s0 : s0 + m]
                want_f.append(f)
                want_o.append(f.translate(rio._COMP_BYTES)[::-1] if r else f)
            want_o = b"".join(want_o)
            assert fwd[:total].tobytes() == b"".join(want_f)
            assert ori[:total].tobytes() == want_o
            assert codes[:total].tobytes() == want_o.translate(_SEQ_TRANS)


def test_readahead_hands_over_items_errors_and_stops_when_closed():
    from remora_amd.io import _readahead

    assert list(_readahead(iter(range(50)), 2)) == list(range(50))
    closed = []

    def failing():
        try:
            yield 1
            yield 2
            raise ValueError("from the producer")
        finally:
            closed.append(True)

    got = []
    with pytest.raises(ValueError, match="from the producer"):
        for x in _readahead(failing(), 2):
            got.append(x)
    assert got == [1, 2] and closed == [True]

    def endless():
        try:
            k = 0
            while True:
                yield k
                k += 1
        finally:
            closed.append("endless")

    g = _readahead(endless(), 2)
    assert [next(g), next(g), next(g)] == [0, 1, 2]
    g.close()  # stops the thread and closes the producer's generator in it
    assert closed[-1] == "endless"


def test_bam_batches_identifiers_only_mode_matches_the_full_parse():
    """iter_bam_raw_batches(light=True) (rmr_bam_read_batch with want_ref bit 1: what count_reads runs on) gives the same
    flags, names, parent ids, tag bits and virtual offsets as the full parse, and empty blobs."""
    from remora_amd import io as rio

    for stem in ("can", "mod"):
        bam = os.path.join(DATA, f"{stem}_mappings.bam")
        pick = lambda rb: (rb.n, rb.flag.tolist(), rb.ref_id.tolist(), rb.pos.tolist(), rb.names, rb.name_off.tolist(), rb.pi, rb.pi_off.tolist(),  # noqa: E731
                           rb.has.tolist(), rb.voffset.tolist(), rb.sm.tolist(), rb.sp.tolist())
        light = [pick(rb) for rb, _ in rio.iter_bam_raw_batches(bam, light=True, batch=5)]
        full = [pick(rb) for rb, _ in rio.iter_bam_raw_batches(bam, batch=5)]
        assert light == full and sum(x[0] for x in light) == 14
        for rb, _ in rio.iter_bam_raw_batches(bam, light=True, batch=5):
            assert rb.raw == b"" and rb.seq == b"" and rb.mv.size == 0 and int(rb.mv_off[-1]) == 0 and int(rb.seq_off[-1]) == 0


def test_weighted_median_is_numpys_median_of_the_expanded_array():
    """io._weighted_median (host half of the median / MAD scaling of reads without sm / sd tags: the GPU hands over a histogram
    of the int16 samples) against np.median on the samples themselves, through Read.pa_signal's arithmetic - equal, not close:
    odd and even counts, ties, a negative calibration scale (the order of the pA values reverses)."""
    from remora_amd import io as rio

    rng = np.random.RandomState(0)
    for t in range(600):
        vals = np.unique(rng.randint(-500, 2000, rng.randint(1, 40))).astype(np.int16)
        cnt = rng.randint(1, 6, vals.size)
        off = float(np.float32(rng.uniform(-300, 300)))
        sc = float(np.float32(rng.uniform(0.1, 0.3))) * (1 if t % 7 else -1)
        dacs = np.repeat(vals, cnt)
        rng.shuffle(dacs)
        pa = (dacs - off) / sc
        med = np.median(pa)
        mad = np.median(np.abs(pa - med))
        pv = (vals - off) / sc
        got = rio._weighted_median(pv, cnt)
        assert got == med and rio._weighted_median(np.abs(pv - got), cnt) == mad


def test_count_reads_shares_add_up_to_the_file():
    """prepare's counting pass (identifiers-only BAM batches, every core inflating) over the ranks' shares of a BAM: the shares
    are disjoint and complete for any world size, with and without the primary filter (get_read_ids, src/remora/io.py:362-391)."""
    from remora_amd import prepare_train_data as p

    for stem in ("can", "mod"):
        pod5, bam = os.path.join(DATA, f"{stem}_reads.pod5"), os.path.join(DATA, f"{stem}_mappings.bam")
        whole = p.count_reads(pod5, bam)
        assert whole == (14, 14) and p.count_reads(pod5, bam, skip_non_primary=False) == (14, 14)
        for world in (2, 3, 5, 16):
            parts = [p.count_reads(pod5, bam, shard=(r, world)) for r in range(world)]
            assert (sum(x[0] for x in parts), sum(x[1] for x in parts)) == whole, (world, parts)
    assert "RMR_BAM_INFLATE_THREADS" not in os.environ  # the pass hands the variable back


@pytest.mark.parametrize("name,padded", [("convlstm_s40_l100_o2", 64), ("conv_s24_l100_o3", 32), ("convlstm_s96_l100_o2", 96),
                                         ("convlstm_s16_l100_o2", 16)])
def test_padded_network_is_the_same_function(O, name, padded):
    """`--size` is any int in the reference (src/remora/parsers.py:858-862); the kernels run at 16 / 32 / 64 or a multiple of 16
    above, so rmr_model_create adds zero-weight channels.  rmr_model_pad_weights is that transform: the padded blob, read back
    as a state dict and evaluated by the ORACLE's forward, returns the reference's logits of the unpadded network (the
    reference-generated golden models), and a size the kernels take as it is comes back unchanged."""
    import ctypes

    from conftest import golden
    from remora_amd import _lib as L
    from remora_amd.engine import _CONV_ORDER, state_to_blob

    g = golden(f"model_{name}.npz")
    state = O.state_from_npz(g)
    size, kb, ka, Lc, num_out = (int(x) for x in g["params"])
    arch, sz, kmer_len, n_out, blob = state_to_blob(state)
    assert sz == size
    lib = L.lib()
    desc = L.ModelDesc(L.ARCH_CONV_LSTM if arch == "conv_lstm" else L.ARCH_CONV_ONLY, size, kmer_len, n_out, Lc, 0)
    assert lib.rmr_model_padded_size(ctypes.byref(desc)) == padded
    pdesc, n_pad = L.ModelDesc(), ctypes.c_size_t()
    L.check(lib.rmr_model_pad_weights(ctypes.byref(desc), blob.ctypes.data, blob.size, ctypes.byref(pdesc), None, 0, ctypes.byref(n_pad)))
    assert pdesc.size == padded and n_pad.value == lib.rmr_model_weight_count(ctypes.byref(pdesc))
    out = np.full(n_pad.value, np.nan, np.float32)
    L.check(lib.rmr_model_pad_weights(ctypes.byref(desc), blob.ctypes.data, blob.size, ctypes.byref(pdesc), out.ctypes.data, out.size,
                                      ctypes.byref(n_pad)))
    assert np.isfinite(out).all()
    if padded == size:
        assert np.array_equal(out, blob)
        return
    # blob -> state dict at the padded size (the inverse of state_to_blob)
    P, ec = padded, 4 * kmer_len
    shapes = {"sig_conv1": (4, 1), "sig_conv2": (16, 4), "sig_conv3": (P, 16), "seq_conv1": (16, ec),
              "seq_conv2": (P, 16) if arch == "conv_lstm" else (32, 16), "seq_conv3": (P, 32),
              "merge_conv1": (P, 2 * P), "merge_conv2": (P, P), "merge_conv3": (P, P), "merge_conv4": (P, P)}
    pstate, at = {}, 0

    def take(shape):
        nonlocal at
        n = int(np.prod(shape))
        v = out[at : at + n].reshape(shape).copy()
        at += n
        return v

    for conv, bn in _CONV_ORDER[arch]:
        kw = np.asarray(state[f"{conv}.weight"]).shape[2]
        oc, ic = shapes[conv]
        pstate[f"{conv}.weight"] = take((oc, ic, kw))
        for key in (f"{conv}.bias", f"{bn}.weight", f"{bn}.bias", f"{bn}.running_mean", f"{bn}.running_var"):
            pstate[key] = take((oc,))
    if arch == "conv_lstm":
        for l in ("lstm1", "lstm2"):
            pstate[f"{l}.weight_ih_l0"], pstate[f"{l}.weight_hh_l0"] = take((4 * P, P)), take((4 * P, P))
            pstate[f"{l}.bias_ih_l0"], pstate[f"{l}.bias_hh_l0"] = take((4 * P,)), take((4 * P,))
        pstate["fc.weight"] = take((num_out, P))
    else:
        pstate["fc.weight"] = take((num_out, 3 * P))
    pstate["fc.bias"] = take((num_out,))
    assert at == out.size
    enc = O.compute_encoded_kmer_batch(kb, ka, g["seqs"], g["maps"], g["lens"])
    got = O.forward(pstate, g["sigs"], enc)
    assert np.abs(got - g["logits"]).max() < 2e-5, float(np.abs(got - g["logits"]).max())
    assert np.abs(got - O.forward(state, g["sigs"], enc)).max() < 1e-6  # zero channels change nothing but the summation tree


def test_winograd_constants_match_the_exact_derivation():
    """k_wino.hip / engine.hip hold the F(4, 5) matrices as constants: the filter transform G (engine.hip `GM`, rows in the kernel's x
    order) and the BT / G / AT table of k_wino.hip's header are the ones oracle/winograd.py derives from the eight points in exact
    rational arithmetic; the derivation itself reproduces the plain sums y[i] = sum_k g[k] d[i + k] exactly; and the even / odd
    evaluation the kernel uses for BT d (restated here line by line from wino_in_transform) equals BT d."""
    import random
    from fractions import Fraction as Fr

    from oracle import winograd as W

    AT, G, BT = W.matrices(4, 5, W.F45_POINTS)
    rnd = random.Random(5)
    for _ in range(25):
        g = [Fr(rnd.randint(-9, 9), rnd.randint(1, 7)) for _ in range(5)]
        d = [Fr(rnd.randint(-9, 9), rnd.randint(1, 7)) for _ in range(8)]
        assert W.apply(AT, G, BT, g, d) == W.correlate(g, d, 4)
    # engine.hip: static const double GM[8][5] = {{...}, ...}; entries are C expressions like 1.0 / 18 or -8.0 / 45
    eng = open(os.path.join(ROOT, "remora_amd", "csrc", "engine.hip")).read()
    blk = eng[eng.index("static const double GM[8][5]") :]
    blk = blk[blk.index("{") : blk.index("};") + 1]
    rows = re.findall(r"\{([^{}]+)\}", blk)
    assert len(rows) == 8
    for k, row in enumerate(rows):
        vals = []
        for e in row.split(","):
            num, _, den = e.strip().partition("/")
            vals.append(Fr(num.strip()) / (Fr(den.strip()) if den else 1))
        assert vals == G[W.F45_KERNEL_ORDER[k]], (k, row)
    # k_wino.hip's header table (BT | G | AT side by side)
    src = open(os.path.join(ROOT, "remora_amd", "csrc", "k_wino.hip")).read()
    tab = src[src.index("//     BT = ") :].split("\n")[:8]
    for i, line in enumerate(tab):
        toks = line.replace("//", "").replace("BT =", "").replace("G =", "").replace("AT =", "").split()
        assert [Fr(t) for t in toks[:8]] == BT[i], (i, line)
        assert [Fr(t) for t in toks[8:13]] == G[i], (i, line)
        if i < 4:
            assert [Fr(t) for t in toks[13:21]] == AT[i], (i, line)
    # wino_in_transform, operation for operation, on exact numbers: v in the kernel's x order
    for _ in range(10):
        d = [Fr(rnd.randint(-50, 50), rnd.randint(1, 9)) for _ in range(8)]
        e1 = -4 * (d[2] + d[6]) + 17 * d[4]
        o1 = -4 * (d[1] + d[5]) + 17 * d[3]
        e2 = -5 * d[4] + (4 * d[6] + d[2])
        o2 = -5 * d[3] + (4 * d[5] + d[1])
        e3 = -5 * d[4] + (4 * d[2] + d[6])
        o3 = -5 * d[3] + (4 * d[1] + d[5])
        v = [e1 + o1, e1 - o1, 2 * o2 + e2, -2 * o2 + e2, -21 * (d[2] - d[4]) + 4 * (d[0] - d[6]), 2 * e3 + o3, 2 * e3 - o3,
             21 * (d[3] - d[5]) + 4 * (d[7] - d[1])]
        want = [sum(BT[x][j] * d[j] for j in range(8)) for x in W.F45_KERNEL_ORDER]
        assert v == want
    # F(4, 3) of the stride-3 kernel: engine.hip `G3` (natural order) and wino_in_transform6 (kernel order 1, 2, 0, 3, 4, 5)
    AT3, G3, BT3 = W.matrices(4, 3, (0, 1, -1, 2, -2, None))
    blk = eng[eng.index("static const double G3[6][3]") :]
    blk = blk[blk.index("{") : blk.index("};") + 1]
    rows = re.findall(r"\{([^{}]+)\}", blk)
    assert len(rows) == 6
    for k, row in enumerate(rows):
        vals = []
        for e in row.split(","):
            num, _, den = e.strip().partition("/")
            vals.append(Fr(num.strip()) / (Fr(den.strip()) if den else 1))
        assert vals == G3[k], (k, row)
    for _ in range(10):
        d = [Fr(rnd.randint(-50, 50), rnd.randint(1, 9)) for _ in range(6)]
        a_, b_ = 4 * d[2] - d[4], 4 * d[1] - d[3]
        c_, e_ = d[4] - d[2], d[3] - d[1]
        v = [a_ + b_, a_ - b_, -5 * d[2] + (4 * d[0] + d[4]), 2 * e_ + c_, -2 * e_ + c_, -5 * d[3] + (4 * d[1] + d[5])]
        assert v == [sum(BT3[x][j] * d[j] for j in range(6)) for x in (1, 2, 0, 3, 4, 5)]
        m = [Fr(rnd.randint(-50, 50), rnd.randint(1, 9)) for _ in range(6)]
        s12, d12, s34, d34 = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4]
        assert [(m[0] + s12) + s34, d12 + 2 * d34, s12 + 4 * s34, d12 + (8 * d34 + m[5])] == [sum(AT3[i][x] * m[x] for x in range(6)) for i in range(4)]
    # the output stage of the kernel (two wave halves) is AT m
    for _ in range(10):
        m = [Fr(rnd.randint(-50, 50), rnd.randint(1, 9)) for _ in range(8)]
        s12, d12, s34, d34, s56, d56 = m[1] + m[2], m[1] - m[2], m[3] + m[4], m[3] - m[4], m[5] + m[6], m[5] - m[6]
        y = [(s12 + s34) + (m[0] + s56), (2 * d34 + d12) + d56 / 2, (4 * s34 + s12) + s56 / 4, (8 * d34 + d12) + (d56 / 8 + m[7])]
        assert y == [sum(AT[i][x] * m[x] for x in range(8)) for i in range(4)]
