"""Python API of the hot path — drop-in for `remora.inference.call_read_mods`
(src/remora/inference.py:661-712) and the per-label tally the multi-GPU runs reduce."""
import numpy as np

from . import RemoraError
from .constants import DEFAULT_BATCH_SIZE
from .util import Motif, format_mm_ml_tags, softmax_axis1


def call_read_mods(read, model, model_metadata, batch_size=DEFAULT_BATCH_SIZE, focus_offset=None,
                   return_mm_ml_tags=False, return_mod_probs=False):
    """Call modified bases on one read; arguments and return values as in the reference:
    (nn_out f32[N,num_out], labels i64[N], pos i64[N]) by default;
    (probs f64[N,num_mods], labels, pos) with return_mod_probs; (MM str, ML array('B')) with
    return_mm_ml_tags; three empty arrays when the read yields no chunk (:698-699)."""
    if focus_offset is None:
        read.set_motif_focus_bases([Motif(*m) for m in model_metadata["motifs"]])
    else:
        read.focus_bases = np.array([focus_offset])
    read.prepare_batches(model_metadata, batch_size)
    if len(read.batches) == 0:
        return np.array([]), np.array([]), np.array([])
    nn_out, labels, pos = read.run_model(model)
    if not return_mod_probs and not return_mm_ml_tags:
        return nn_out, labels, pos
    probs = softmax_axis1(nn_out)[:, 1:].astype(np.float64)
    if return_mm_ml_tags:
        return format_mm_ml_tags(seq=read.str_seq, poss=pos, probs=probs, mod_bases=model_metadata["mod_bases"],
                                 can_base=model_metadata["can_base"])
    return probs, labels, pos


def find_focus_bases_batch(reads, motifs):
    """Motif hits for many reads at once (sorted ascending per read).  Vectorised restatement of
    Motif.findall (src/remora/util.py:281-297) over the concatenated sequences; unlike
    find_focus_bases_in_int_sequence (:413-426) the per-read order is ascending, not python-set
    order (downstream consumers sort by position anyway, src/remora/util.py:506,518)."""
    lens = np.array([r.int_seq.size for r in reads], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)])
    cat = np.concatenate([np.asarray(r.int_seq, dtype=np.int64) for r in reads]) if len(reads) else np.zeros(0, np.int64)
    read_of = np.repeat(np.arange(len(reads)), lens)
    hit_any = np.zeros(cat.size, dtype=bool)
    cat1 = (cat + 1).astype(np.intp)  # -1 (N) -> 0
    for mot in motifs:
        m = len(mot.raw_motif)
        nwin = cat.size - m + 1
        if nwin <= 0:
            continue
        hit = np.ones(nwin, dtype=bool)
        for po, allowed in enumerate(mot.int_pattern):
            lut = np.zeros(5, dtype=bool)
            lut[np.asarray(allowed) + 1] = True
            hit &= lut[cat1[po : po + nwin]]
        # a window must not straddle two reads
        hit &= read_of[:nwin] == read_of[m - 1 : m - 1 + nwin]
        focus = np.flatnonzero(hit) + mot.focus_pos
        focus = focus[(focus >= 0) & (focus < cat.size)]
        hit_any[focus] = True
    pos = np.flatnonzero(hit_any)
    owner = read_of[pos]
    counts = np.bincount(owner, minlength=len(reads))
    local = pos - offs[owner]
    return np.split(local, np.cumsum(counts)[:-1]) if len(reads) else []


def call_reads_mods(reads, model, model_metadata, return_mod_probs=False):
    """Batched form of call_read_mods for a list of RemoraRead objects: one motif scan, one chunk
    extraction and one fused inference for all reads (the reference processes reads one by one
    in Python, src/remora/inference.py:62-137, 661-712).  Returns a list of per-read
    (nn_out | probs, labels, pos) tuples; pos ascending within a read."""
    from .data_chunks import extract_chunk_arrays

    motifs = [Motif(*m) for m in model_metadata["motifs"]]
    focus = find_focus_bases_batch(reads, motifs)
    refiner = model_metadata.get("sig_map_refiner")
    if refiner is not None and getattr(refiner, "is_loaded", False):
        for err in refiner.refine_reads(reads):  # one GPU pass per DP round for the whole batch
            if err is not None:
                raise err
    for r, fb in zip(reads, focus):
        r.focus_bases = fb
    arrs, _ = extract_chunk_arrays(reads, model_metadata["chunk_context"], model_metadata["kmer_context_bases"],
                                   model_metadata["base_start_justify"], model_metadata["offset"])
    counts = np.array([len(fb) for fb in focus], dtype=np.int64)
    if len(arrs) == 0:
        return [(np.array([]), np.array([]), np.array([])) for _ in reads]
    out = model.infer_chunks(arrs.signal, arrs.sequence, arrs.mapping, arrs.lengths, arrs.kmer_context_bases)
    out = out.cpu().numpy()
    pos = arrs.read_focus_bases.cpu().numpy()
    if return_mod_probs:
        out = softmax_axis1(out)[:, 1:].astype(np.float64)
    cuts = np.cumsum(counts)[:-1]
    res = []
    for o, l, p in zip(np.split(out, cuts), np.split(arrs.labels, cuts), np.split(pos, cuts)):
        res.append((o, l, p) if p.size else (np.array([]), np.array([]), np.array([])))
    return res


def infer_from_pod5_and_bam(pod5_path, in_bam_path, model, model_metadata, out_bam_path, num_reads=None,
                            reads_per_batch=256, reverse_signal=None, skip_non_primary=True):
    """`remora infer from_pod5_and_bam` for one model, basecall-anchored
    (src/remora/inference.py:462-641): every input alignment is written to `out_bam_path` with
    MM/ML tags from the model (records whose read cannot be called are written unchanged and
    counted by reason, as the reference does).  Reads are grouped into batches that go through
    ONE chunk extraction and ONE fused inference on the GPU.  Returns {reason: count, ...} with
    the key None counting successfully called reads."""
    from collections import Counter

    from . import io as rio

    if reverse_signal is None:
        reverse_signal = bool(model_metadata.get("reverse_signal", False))
    pa_scaling = model_metadata.get("pa_scaling")
    stats = Counter()
    header = rio.read_bam_header_bytes(in_bam_path)

    def flush(batch, writer):
        good = []
        for io_read, err in batch:
            if err is None:
                try:
                    good.append((io_read, io_read.into_remora_read(False)))
                    continue
                except RemoraError as e:
                    err = f"Read prep error: {e}"
            stats[err] += 1
            writer.write(rio.record_with_mod_tags(io_read.record, None, None))
        if not good:
            return
        results = call_reads_mods([rr for _, rr in good], model, model_metadata, return_mod_probs=True)
        for (io_read, rr), (probs, _, pos) in zip(good, results):
            if pos.size == 0:
                stats[f"No {model_metadata['can_base']} mod calls"] += 1
                writer.write(rio.record_with_mod_tags(io_read.record, None, None))
                continue
            mm, ml = format_mm_ml_tags(seq=io_read.seq, poss=pos, probs=probs, mod_bases=model_metadata["mod_bases"],
                                       can_base=model_metadata["can_base"])
            stats[None] += 1
            writer.write(rio.record_with_mod_tags(io_read.record, mm, ml))

    with rio.BamWriter(out_bam_path, header) as writer:
        batch = []
        for i, item in enumerate(rio.iter_reads_from_pod5_and_bam(pod5_path, in_bam_path, reverse_signal=reverse_signal,
                                                                  pa_scaling=pa_scaling, skip_non_primary=skip_non_primary)):
            if num_reads is not None and i >= num_reads:
                break
            batch.append(item)
            if len(batch) >= reads_per_batch:
                flush(batch, writer)
                batch = []
        if batch:
            flush(batch, writer)
    return dict(stats)
