"""Pins the CPU oracle (oracle/) to golden vectors produced by RUNNING the reference
(tools/gen_golden.py).  CPU only."""
import json

import numpy as np
import pytest

from conftest import code_to_onehot, golden
from oracle import oracle as O


def test_parse_move_tag_golden():
    g = golden("parse_move_tag.npz")
    for i in range(int(g["num_cases"])):
        mv = g[f"c{i}_mv"]
        sig_len, seq_len, rev, check = (int(x) for x in g[f"c{i}_args"])
        err = str(g[f"c{i}_err"])
        kw = dict(seq_len=None if seq_len < 0 else seq_len, check=bool(check),
                  reverse_signal=bool(rev))
        if err:
            with pytest.raises(O.OracleError, match=err):
                O.parse_move_tag(mv, sig_len, **kw)
        else:
            q2s, mvt, stride = O.parse_move_tag(mv, sig_len, **kw)
            assert np.array_equal(q2s, g[f"c{i}_q2s"])
            assert stride == mv[0]


def test_seq_motif_golden():
    g = golden("seq_motif.npz")
    for si in range(int(g["num_seqs"])):
        s = str(g[f"s{si}_str"])
        int_seq = O.seq_to_int(s)
        assert np.array_equal(int_seq, g[f"s{si}_int"])
        for mi in range(int(g["num_motif_sets"])):
            if int(g[f"s{si}_m{mi}_skipped"]):
                continue
            motifs = list(zip([str(x) for x in g[f"m{mi}_seqs"]], g[f"m{mi}_offs"]))
            fb = O.find_focus_bases(int_seq, motifs)
            # exact order: the reference's python-set iteration order
            assert np.array_equal(fb, g[f"s{si}_m{mi}_focus"]), (si, mi)
    for mi in range(int(g["num_motif_sets"])):
        for raw, off, nraw, noff in zip(g[f"m{mi}_seqs"], g[f"m{mi}_offs"],
                                        g[f"m{mi}_norm_seqs"], g[f"m{mi}_norm_offs"]):
            assert O.normalise_motif(str(raw), off) == (str(nraw), int(noff))


def test_extract_chunks_golden():
    g = golden("extract_chunks.npz")
    for rname in g["read_names"]:
        rname = str(rname)
        shift, scale = g[f"{rname}_shift_scale"]
        sig = O.normalise_signal(g[f"{rname}_dacs"], float(shift), float(scale))
        assert np.array_equal(sig.view(np.uint32), g[f"{rname}_sig"].view(np.uint32))
        for mname in ("CG", "C"):
            fbs = g[f"{rname}_{mname}_focus"]
            assert np.array_equal(
                O.find_focus_bases(g[f"{rname}_int_seq"], [(mname, 0)]), fbs)
            for ci, cfg in enumerate(g["configs"]):
                cc, kcb, bsj, off = (cfg[0], cfg[1]), (cfg[2], cfg[3]), bool(cfg[4]), int(cfg[5])
                pre = f"{rname}_{mname}_c{ci}_"
                ch = O.extract_chunks(sig, g[f"{rname}_map"], g[f"{rname}_int_seq"], fbs, cc,
                                      kcb, bsj, off)
                n = fbs.size
                assert np.array_equal(ch["signal"].reshape(n, -1).view(np.uint32),
                                      g[pre + "signal"].view(np.uint32))
                sl = g[pre + "seq_len"]
                assert np.array_equal(ch["sequence_lengths"], sl)
                for i in range(n):
                    assert np.array_equal(ch["sequence"][i, : sl[i] + sum(kcb)],
                                          g[pre + "seq_w_context"][i, : sl[i] + sum(kcb)])
                    assert np.array_equal(ch["maps_i32"][i, : sl[i] + 1],
                                          g[pre + "seq_to_sig_map"][i, : sl[i] + 1])
                misc = g[pre + "misc"]
                assert np.array_equal(ch["chunk_sig_focus_idx"], misc[:, 0])
                assert np.array_equal(ch["chunk_focus_base"], misc[:, 1])
                assert np.array_equal(ch["read_focus_bases"], misc[:, 2])


def test_encode_kmers_golden():
    g = golden("encode_kmers.npz")
    for i in range(int(g["num_cases"])):
        kb, ka, L = (int(x) for x in g[f"c{i}_args"])
        enc = O.compute_encoded_kmer_batch(kb, ka, g[f"c{i}_seqs"], g[f"c{i}_maps"],
                                           g[f"c{i}_lens"])
        ref = code_to_onehot(g[f"c{i}_enc_code"])
        assert enc.shape == ref.shape
        assert np.array_equal(enc, ref)


def test_trim_golden():
    g = golden("trim_chunk_context.npz")
    for i in range(int(g["num_cases"])):
        scc0, scc1, cc0, cc1, tsc = (int(x) for x in g[f"c{i}_args"])
        seqs, lens = g[f"c{i}_in_seqs"].copy(), g[f"c{i}_in_lens"].copy()
        maps = (g[f"c{i}_in_maps"] - (scc0 - cc0)).astype(np.int16)
        O.trim_sb_chunk_context_core(scc0, scc1, cc0, cc1, tsc, seqs, maps, lens)
        assert np.array_equal(lens, g[f"c{i}_out_lens"])
        # whole arrays are defined by the in-place algorithm, incl. stale tails
        assert np.array_equal(maps, g[f"c{i}_out_maps"])
        assert np.array_equal(seqs, g[f"c{i}_out_seqs"])


MODELS = ["convlstm_s64_l100_o2", "convlstm_s64_l200_o3", "convlstm_s16_l100_o2",
          "convlstm_s64_l100_k23", "conv_s64_l100_o2", "conv_s64_l100_o3",
          # any `--size` (src/remora/parsers.py:858-862): streamed-weight kernels above 64, zero-padded channels otherwise
          "convlstm_s96_l100_o2", "convlstm_s128_l100_o2", "conv_s96_l100_o2", "convlstm_s40_l100_o2", "conv_s24_l100_o3"]


@pytest.mark.parametrize("name", MODELS)
def test_forward_c_oracle_golden(name):
    g = golden(f"model_{name}.npz")
    state = O.state_from_npz(g)
    size, kb, ka, L, num_out = (int(x) for x in g["params"])
    enc = O.compute_encoded_kmer_batch(kb, ka, g["seqs"], g["maps"], g["lens"])
    out = O.forward(state, g["sigs"], enc)
    assert np.abs(out - g["logits"]).max() < 2e-5
    out_d = O.forward(state, g["sigs"][:8], g["dense_seqs"])
    assert np.abs(out_d - g["dense_logits"]).max() < 5e-5


@pytest.mark.parametrize("name", MODELS)
def test_forward_torch_restatement_golden(name):
    import torch
    from oracle import torch_ref

    g = golden(f"model_{name}.npz")
    net = torch_ref.from_state(O.state_from_npz(g))
    size, kb, ka, L, num_out = (int(x) for x in g["params"])
    enc = O.compute_encoded_kmer_batch(kb, ka, g["seqs"], g["maps"], g["lens"])
    with torch.no_grad():
        out = net(torch.from_numpy(g["sigs"]), torch.from_numpy(enc)).numpy()
    assert np.abs(out - g["logits"]).max() < 1e-6


@pytest.mark.parametrize("name", ["cg_5mc", "allc_5hmc_5mc", "conv_cg"])
def test_call_read_mods_golden(name):
    g = golden(f"call_read_mods_{name}.npz")
    state = O.state_from_npz(g)
    md = json.loads(str(g["derived_md_json"]))
    for rname in g["read_names"]:
        rname = str(rname)
        shift, scale = (float(x) for x in g[f"{rname}_shift_scale"])
        nn_out, labels, pos = O.call_read_mods(
            g[f"{rname}_dacs"], shift, scale, g[f"{rname}_map"], g[f"{rname}_int_seq"], state, md)
        assert np.array_equal(pos, g[f"{rname}_pos"])
        assert np.array_equal(labels, g[f"{rname}_labels"])
        if pos.size:
            assert np.abs(nn_out - g[f"{rname}_nn_out"]).max() < 2e-5
            probs = O.softmax_axis1(g[f"{rname}_nn_out"])[:, 1:].astype(np.float64)
            assert np.array_equal(probs, g[f"{rname}_probs"])
            seq = "".join("ACGTN"[b] for b in g[f"{rname}_int_seq"])
            mm, ml = O.format_mm_ml_tags(seq, pos, probs, md["mod_bases"], md["can_base"])
            assert mm == str(g[f"{rname}_mm"])
            assert np.array_equal(np.asarray(list(ml), np.uint8), g[f"{rname}_ml"])
        else:
            assert str(g[f"{rname}_mm"]) == "<EMPTY3>"
    fo = int(g["r_long_focus_offset"])
    o2, _, p2 = O.call_read_mods(g["r_long_dacs"], *(float(x) for x in g["r_long_shift_scale"]),
                                 g["r_long_map"], g["r_long_int_seq"], state, md, focus_offset=fo)
    assert np.array_equal(p2, g["r_long_focus_pos"])
    assert np.abs(o2 - g["r_long_focus_nn_out"]).max() < 2e-5


def test_post_process_golden():
    g = golden("post_process.npz")
    assert np.array_equal(O.softmax_axis1(g["softmax_in"]), g["softmax_out"])
    mm, ml = O.format_mm_ml_tags(str(g["tags_seq"]), g["tags_poss"], g["tags_probs"],
                                 ["h", "m"], "C")
    assert mm == str(g["tags_mm"])
    assert np.array_equal(np.asarray(list(ml), np.uint8), g["tags_ml"])


def test_prepare_batches_golden():
    g = golden("prepare_batches.npz")
    shift, scale = (float(x) for x in g["shift_scale"])
    sig = O.normalise_signal(g["dacs"], shift, scale)
    fbs = O.find_focus_bases(g["int_seq"], [("CG", 0)])
    ch = O.extract_chunks(sig, g["map"], g["int_seq"], fbs, (50, 50), (4, 4))
    assert np.array_equal(ch["signal"].view(np.uint32), g["signal"].view(np.uint32))
    assert np.array_equal(ch["read_focus_bases"], g["read_focus_bases"])
    enc = O.compute_encoded_kmer_batch(4, 4, ch["sequence"], ch["sequence_to_signal_mapping"],
                                       ch["sequence_lengths"])
    assert np.array_equal(enc, code_to_onehot(g["enc_code"]))


# ---------------------------------------------------------------------------------------
# N2: signal-mapping refinement (refine_signal_map.py / refine_signal_map_core.pyx)
# ---------------------------------------------------------------------------------------

REFINE_READS = ["a", "b", "c", "d"]


def _refine_golden():
    return golden("refine_signal_map.npz")


@pytest.mark.parametrize("name", REFINE_READS)
def test_refine_levels_and_bands(name):
    g = _refine_golden()
    lv = O.extract_levels(g[f"{name}_int_seq"], g["kmer_levels"], int(g["center_idx"]))
    np.testing.assert_array_equal(lv, g[f"{name}_levels"])
    for hbw in (5, 2):
        band = O.convert_to_seq_band(O.compute_sig_band(g[f"{name}_map"], lv, hbw))
        np.testing.assert_array_equal(band, g[f"{name}_hbw{hbw}_band_raw"])
        band = np.ascontiguousarray(band, np.int32)
        O.lib().orc_adjust_seq_band(O._p(band), band.shape[1], 2)
        np.testing.assert_array_equal(band, g[f"{name}_hbw{hbw}_band"])


@pytest.mark.parametrize("algo", ["Viterbi", "dwell_penalty"])
@pytest.mark.parametrize("name", REFINE_READS)
def test_refine_banded_dp(name, algo):
    g = _refine_golden()
    sig = ((g[f"{name}_dacs"] - 505.0) / 83.0).astype(np.float32)
    for hbw in (5, 2):
        pre = f"{name}_hbw{hbw}_{algo}_"
        scores, path, tb, _ = O.seq_banded_dp(sig, g[f"{name}_levels"], g[f"{name}_hbw{hbw}_band"], g["sd_arr"], algo)
        np.testing.assert_array_equal(path, g[pre + "path"])
        if pre + "scores" in g:
            np.testing.assert_array_equal(scores, g[pre + "scores"])  # bit-exact fp32 recurrence
            np.testing.assert_array_equal(tb, g[pre + "tb"])


def test_refine_read_flow():
    g = _refine_golden()
    settings = json.loads(str(g["settings_json"]))
    for si, st in enumerate(settings):
        ref = dict(st, levels=g["kmer_levels"], center_idx=int(g["center_idx"]), sd_arr=g["sd_arr"])
        for name in REFINE_READS:
            np.random.seed(1000 + si)
            s2s, shift, scale = O.refine_read(ref, g[f"{name}_dacs"], 505.0, 83.0, g[f"{name}_map"].copy(),
                                                   g[f"{name}_int_seq"])
            np.testing.assert_array_equal(s2s, g[f"s{si}_{name}_map"], err_msg=f"setting {si} read {name}")
            np.testing.assert_allclose([shift, scale], g[f"s{si}_{name}_shift_scale"], rtol=1e-12)


def test_call_read_mods_with_refiner_golden():
    """A model whose metadata carries a k-mer level table: the oracle's refinement (rough re-scale +
    dwell-penalty DP) followed by its call_read_mods reproduces the reference's call_read_mods."""
    g = golden("call_read_mods_cg_5mc_refine.npz")
    state = O.state_from_npz(g)
    md = json.loads(str(g["derived_md_json"]))
    raw = json.loads(str(g["meta_txt"]))
    assert md["base_start_justify"] is True and md["offset"] == 1
    ref = dict(levels=np.frombuffer(raw["refine_kmer_levels"].encode("cp437"), dtype=np.float32),
               center_idx=int(raw["refine_kmer_center_idx"]), do_rough_rescale=bool(raw["refine_do_rough_rescale"]),
               scale_iters=int(raw["refine_scale_iters"]), algo=raw["refine_algo"],
               half_bandwidth=int(raw["refine_half_bandwidth"]),
               sd_arr=np.frombuffer(raw["refine_sd_arr"].encode("cp437"), dtype=np.float32),
               rough_rescale_method="least_squares")
    moved = 0
    for rname in g["read_names"]:
        rname = str(rname)
        shift, scale = (float(x) for x in g[f"{rname}_shift_scale"])
        s2s, shift2, scale2 = O.refine_read(ref, g[f"{rname}_dacs"], shift, scale, g[f"{rname}_map"].copy(),
                                            g[f"{rname}_int_seq"])
        moved += int((s2s != g[f"{rname}_map"]).sum())
        nn_out, labels, pos = O.call_read_mods(g[f"{rname}_dacs"], shift2, scale2, s2s, g[f"{rname}_int_seq"], state, md)
        assert np.array_equal(pos, g[f"{rname}_pos"])
        assert pos.size and np.abs(nn_out - g[f"{rname}_nn_out"]).max() < 2e-5
    assert moved > 100  # the refinement really changed the mappings


def test_vbz_oracle_on_the_reference_pod5_rows():
    """orc_vbz_decode on every signal row of the reference's tests/data/can_reads.pod5 equals the numpy decoder
    whose output feeds the reference-generated read golden (signal checksums in real_reads_can.npz)."""
    import os

    from conftest import GOLDEN
    from remora_amd import io as rio

    f = rio.Pod5File(os.path.join(GOLDEN, "data", "can_reads.pod5"))
    rows, ns = f._sig.column("signal"), f._sig.column("samples")
    total = 0
    for i in range(f._sig.num_rows):
        blob, n = rows[i].as_py(), ns[i].as_py()
        raw = bytes(rio._zstd_decompress(blob))
        np.testing.assert_array_equal(O.vbz_decode(raw, n), O.vbz_decode_numpy(raw, n))
        total += n
    assert total > 500000
    g = golden("real_reads_can.npz")
    from golden_util import pod5_reads_cpu

    read = pod5_reads_cpu(os.path.join(GOLDEN, "data", "can_reads.pod5"), read_ids=[str(g["r0_name"])])[0]
    assert read.signal.dtype == np.int16 and read.signal.size >= int(g["r0_ndacs"])
    with pytest.raises(O.OracleError):
        O.vbz_decode(b"\x01\x02", 8)
