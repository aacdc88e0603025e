"""`remora dataset prepare`: labelled chunk datasets from POD5 + BAM, the ETL that sits in
front of the hot path's on-disk input format.  Mirrors src/remora/prepare_train_data.py:
`extract_chunks` (:33-118) and `extract_chunk_dataset` (:124-276) with the same arguments and per-read
semantics (reference- or basecall-anchored reads, motif or BED-selected focus bases, signal-mapping
refinement, `max_chunks_per_read` down-sampling with numpy's global RNG, chunk checks, max_seq_len filter,
read_ids / read_focus_bases extra arrays, final shuffle).

What is different is the execution shape: reads are joined BAM-stream against POD5 random access and handled
`reads_per_batch` at a time - ONE upload, ONE banded-DP refinement launch, ONE extraction launch per batch
(`extract_chunk_arrays`) whose outputs already are the dataset's row layout and go to the memmaps with
`write_chunk_arrays` - instead of one Python `Chunk` object per focus base across three process pools."""
import os
from collections import defaultdict

import numpy as np

from . import RemoraError
from . import io as rio
from .data_chunks import (CoreRemoraDataset, DatasetMetadata, DeviceReads, RemoraRead, _extract_device,
                          compute_ref_to_signal)
from .engine import _torch


def _training_read(io_read, int_label, motifs, focus_ref_pos, basecall_anchor):
    """RemoraRead with labels and focus bases for one aligned io.Read (prepare_train_data.py:53-91)."""
    if basecall_anchor:
        rr = io_read.into_remora_read(use_reference_anchor=False)
        rr.focus_bases = io_read.get_basecall_anchored_focus_bases(motifs=motifs,
                                                                   select_focus_reference_positions=focus_ref_pos)
        rr.labels = np.full(len(io_read.seq), int_label, dtype=int)
        return rr
    if io_read.ref_to_signal is None:  # (add_alignment already composed move table and CIGAR when it parsed the alignment)
        io_read.ref_to_signal = compute_ref_to_signal(io_read.query_to_signal, io_read.cigar)
    if io_read.ref_to_signal.size != len(io_read.ref_seq) + 1:
        raise RemoraError(f"discordant ref seq lengths: move+cigar:{io_read.ref_to_signal.size} "
                          f"ref_seq:{len(io_read.ref_seq)}")
    r2s = io_read.ref_to_signal
    rr = RemoraRead(dacs=io_read.dacs[r2s[0] : r2s[-1]], shift=io_read.shift_dacs_to_norm,
                    scale=io_read.scale_dacs_to_norm, seq_to_sig_map=r2s - r2s[0], str_seq=io_read.ref_seq,
                    labels=np.full(len(io_read.ref_seq), int_label, dtype=int), read_id=io_read.read_id)
    if focus_ref_pos is None:
        rr.set_motif_focus_bases(motifs)
    else:
        rr.focus_bases = io_read.get_filtered_focus_positions(focus_ref_pos)
    return rr


def _refine(reads, sig_map_refiner, engine):
    """Signal-mapping refinement of a batch (RemoraRead.refine_signal_mapping for every read, batched on the
    device); -> (DeviceReads of the surviving reads, surviving reads, {index in `reads`: error text})."""
    refiner = sig_map_refiner
    loaded = refiner is not None and getattr(refiner, "is_loaded", False)
    errs = {}
    if loaded and refiner.scale_iters > 0:
        for i, err in enumerate(refiner.refine_reads(reads)):
            if err is not None:
                errs[i] = str(err)
        alive = [r for i, r in enumerate(reads) if i not in errs]
        return (DeviceReads(alive, engine) if alive else None), alive, errs
    alive = list(reads)
    if not alive:
        return None, alive, errs
    dr = DeviceReads(alive, engine)
    if loaded and refiner.do_rough_rescale:
        refiner.rough_rescale_device(dr, alive)
    if loaded and refiner.scale_iters == 0:
        try:
            refiner.refine_device_reads(dr, alive)
        except RemoraError:
            # some read has an invalid band: find it read by read, then redo the batch without the failures
            for i, r in enumerate(alive):
                try:
                    r.seq_to_sig_map, _, _ = refiner.refine_sig_map(r.shift, r.scale, r.seq_to_sig_map, r.int_seq, r.dacs)
                except RemoraError as e:
                    errs[i] = str(e)
            alive = [r for i, r in enumerate(alive) if i not in errs]
            dr = DeviceReads(alive, engine) if alive else None
    return dr, alive, errs


def extract_chunk_arrays_from_reads(read_errs, int_label, motifs, focus_ref_pos, sig_map_refiner, max_chunks_per_read,
                                    chunk_context, kmer_context_bases, base_start_justify, offset, basecall_anchor,
                                    engine=None):
    """Batched `extract_chunks`: -> (ChunkArrays | None, read id of every chunk, keep mask, per-read results).
    Per-read results, in input order: (error text | None, (first, last+1) chunk rows | None).  Reads whose
    RemoraRead.check fails are left out of the results, as in the reference (:95-99)."""
    torch = _torch()
    results, cands = [], []  # results[slot] = [error, row range]; None marks a read dropped by check()
    for io_read, err in read_errs:
        if err is None and io_read.ref_seq is None:
            err = "No reference sequence (missing MD tag)"
        results.append([err, None])
        if err is None:
            cands.append((len(results) - 1, _training_read(io_read, int_label, motifs, focus_ref_pos, basecall_anchor)))
    dr, alive, errs = _refine([rr for _, rr in cands], sig_map_refiner, engine)
    checked = []
    for k, (slot, rr) in enumerate(cands):
        if k in errs:
            results[slot][0] = errs[k]
            continue
        rr.downsample_focus_bases(max_chunks_per_read)  # np.random.choice, in read order like the reference
        try:
            rr.check()
            checked.append((slot, rr))
        except RemoraError:
            results[slot] = None
    if dr is not None and len(checked) != len(alive):
        dr = DeviceReads([rr for _, rr in checked], engine) if checked else None
    arrs, ids, keep = None, [], np.zeros(0, bool)
    if dr is not None:
        # focus bases that still sit on a motif (iter_chunks(motifs=...), data_chunks.py:435-440)
        focus_list, labels_list = [], []
        foc_off = np.zeros(len(checked) + 1, np.int64)
        for i, (slot, rr) in enumerate(checked):
            fbs = np.asarray(rr.focus_bases if rr.focus_bases is not None else [], dtype=np.int64)
            if fbs.size:
                on_motif = np.zeros(fbs.size, bool)
                for m in motifs:
                    on_motif |= m.match_many(rr.int_seq, fbs)
                fbs = fbs[on_motif]
            focus_list.append(fbs)
            labels_list.append(np.asarray(rr.labels)[fbs] if fbs.size else np.zeros(0, np.int64))
            ids += [rr.read_id] * fbs.size
            foc_off[i + 1] = foc_off[i] + fbs.size
            results[slot][1] = (int(foc_off[i]), int(foc_off[i + 1]))
        if foc_off[-1] > 0:
            focus = torch.from_numpy(np.concatenate(focus_list)).to(dr.engine.torch_device)
            arrs, _ = _extract_device(dr, focus, foc_off, chunk_context, kmer_context_bases, base_start_justify, offset,
                                      np.concatenate(labels_list).astype(np.int64))
            keep = ~torch.isnan(arrs.signal).any(dim=2).any(dim=1).cpu().numpy()  # Chunk.check: "Signal contains NaN"
    return arrs, ids, keep, [tuple(r) for r in results if r is not None]


def _prefetched(iterable, device, depth=2):
    """`iterable` consumed in a thread of its own, `depth` items ahead (the ingest of the next BAM batches - BGZF inflate, zstd,
    the decode and assembly kernels on the ingest engine - runs under the extraction and the dataset writes of the current
    one; the role the reference gives its reader processes, src/remora/prepare_train_data.py:190-235).  Exceptions of the
    producer surface in the consumer; a consumer that leaves early stops the producer."""
    import queue
    import threading
    import time

    q, stop, end = queue.Queue(maxsize=max(int(depth), 1)), threading.Event(), object()

    def put(item):
        while not stop.is_set():
            try:
                return q.put(item, timeout=0.1)
            except queue.Full:
                continue

    def produce():
        try:
            if device is not None:
                _torch().cuda.set_device(device)  # a new thread starts on device 0
            for item in iterable:
                put(item)
                if stop.is_set():
                    return
            put(end)
        except BaseException as e:  # noqa: BLE001 - handed to the consumer
            put(e)

    th = threading.Thread(target=produce, daemon=True)
    th.start()
    try:
        while True:
            item = q.get()
            if item is end:
                return
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:
        # the producer runs GPU ingest and owns the native BAM handle: it is stopped AND joined before the caller goes on
        # (write_metadata, interpreter exit), as io._readahead and validate.py do - drain the queue so a blocked put returns
        stop.set()
        deadline = time.monotonic() + 30.0
        while th.is_alive() and time.monotonic() < deadline:
            try:
                q.get_nowait()
            except queue.Empty:
                th.join(0.02)


def extract_chunk_arrays_from_ingest(ib, int_label, motifs, sig_map_refiner, max_chunks_per_read, chunk_context,
                                     kmer_context_bases, base_start_justify, offset):
    """`extract_chunk_arrays_from_reads` for a reference-anchored io.IngestBatch (reads assembled on the GPU, no io.Read /
    RemoraRead object per read): the same four return values, or None when the batch has to go read by read after all (a
    refiner that rejects a read's band).  What stays per read on the host is what decides the bytes of the dataset: the focus
    bases in the reference's python-set order (util.find_focus_bases_in_int_sequence, src/remora/util.py:413-426) and their
    down-sampling with numpy's global generator in read order (prepare_train_data.py:80-91)."""
    from . import util

    torch = _torch()
    results = []
    for e in ib.err:
        if e == "Read prep error: Missing reference alignment":  # (the text of infer's into_remora_read; prepare's is this one)
            e = "No reference sequence (missing MD tag)"
        results.append([e, None])
    arrs, ids, keep = None, [], np.zeros(0, bool)
    if not ib.good.size:
        return arrs, ids, keep, [tuple(r) for r in results]
    dr, stubs = ib.dr, ib.reads
    refiner = sig_map_refiner
    loaded = refiner is not None and getattr(refiner, "is_loaded", False)
    if loaded:
        try:
            if refiner.do_rough_rescale:
                refiner.rough_rescale_device(dr, stubs)
            if refiner.scale_iters == 0:
                refiner.refine_device_reads(dr, stubs)
        except RemoraError:
            return None
    iseq = ib.iseq  # the reads' base codes, back to back (io._ingest_batch)
    so = ib.seq_off.tolist()
    rb = ib.rb
    has_pi = (rb.has & 64) != 0
    focus_list = []
    foc_off = np.zeros(ib.good.size + 1, np.int64)
    # all reads' focus bases in the interpreter's set order by one native call (csrc/pyset_order.c); None: these motifs / this
    # interpreter are outside what it restates -> util.find_focus_bases_in_int_sequence read by read
    from .data_chunks import focus_bases_set_order

    native = focus_bases_set_order(iseq, ib.seq_off, motifs, threads=int(os.environ.get("RMR_PACK_THREADS", "4") or 4))
    if native is not None:
        all_fbs, all_off = native
        all_off = all_off.tolist()
    for g, k in enumerate(ib.good.tolist()):
        if native is not None:
            fbs = all_fbs[all_off[g] : all_off[g + 1]]
        else:
            fbs = util.find_focus_bases_in_int_sequence(iseq[so[g] : so[g + 1]], motifs)
        if fbs.size > max_chunks_per_read:  # RemoraRead.downsample_focus_bases: np.random.choice(fbs, size, replace=False),
            fbs = fbs[np.random.permutation(fbs.size)[:max_chunks_per_read]]  # which is this draw (legacy RandomState.choice)
        focus_list.append(fbs.astype(np.int64, copy=False))
        foc_off[g + 1] = foc_off[g] + fbs.size
        i = int(ib.keep[k])
        rid = (rb.pi[rb.pi_off[i] : rb.pi_off[i + 1]] if has_pi[i] else rb.names[rb.name_off[i] : rb.name_off[i + 1]]).decode("latin-1")
        ids += [rid] * fbs.size
        results[k][1] = (int(foc_off[g]), int(foc_off[g + 1]))
    if foc_off[-1] > 0:
        focus = torch.from_numpy(np.concatenate(focus_list)).to(dr.engine.torch_device)
        arrs, _ = _extract_device(dr, focus, foc_off, chunk_context, kmer_context_bases, base_start_justify, offset,
                                  np.full(int(foc_off[-1]), int_label, np.int64))
        keep = ~torch.isnan(arrs.signal).any(dim=2).any(dim=1).cpu().numpy()  # Chunk.check: "Signal contains NaN"
    return arrs, ids, keep, [tuple(r) for r in results]


def extract_chunks(read_errs, int_label, motifs, focus_ref_pos, sig_map_refiner, max_chunks_per_read, chunk_context,
                   kmer_context_bases, base_start_justify, offset, basecall_anchor, engine=None):
    """The reference's signature and return shape (prepare_train_data.py:33-118): a list of
    (list of Chunk | None, error text | None), one entry per read that was not dropped by RemoraRead.check.
    Extraction itself is the batched GPU path."""
    from .data_chunks import Chunk

    arrs, ids, keep, results = extract_chunk_arrays_from_reads(
        read_errs, int_label, motifs, focus_ref_pos, sig_map_refiner, max_chunks_per_read, chunk_context,
        kmer_context_bases, base_start_justify, offset, basecall_anchor, engine)
    if arrs is not None:
        sig = arrs.signal.cpu().numpy()[:, 0]
        seqs, maps, geo = arrs.sequence.cpu().numpy(), arrs.mapping.cpu().numpy(), arrs.geo.cpu().numpy()
        ctx = sum(arrs.kmer_context_bases)
    out = []
    for err, rows in results:
        if err is not None:
            out.append((None, err))
            continue
        chunks = []
        for i in range(*(rows or (0, 0))):
            if not keep[i]:
                continue
            sl = int(geo[i, 0])
            chunks.append(Chunk(signal=sig[i].copy(), seq_w_context=seqs[i, : sl + ctx].copy(),
                                seq_to_sig_map=maps[i, : sl + 1].astype(np.int32),
                                kmer_context_bases=arrs.kmer_context_bases, chunk_sig_focus_idx=int(geo[i, 1]),
                                chunk_focus_base=int(geo[i, 2]), read_focus_base=int(geo[i, 3]), read_id=ids[i],
                                label=int(arrs.labels[i])))
        out.append((chunks, None))
    return out


def count_reads(pod5_path, bam_path, skip_non_primary=True, shard=None):
    """Number of BAM records (after the primary filter) whose parent read has signal in the POD5 file, and the
    total record count - the quantities of get_read_ids (src/remora/io.py:362-391).  `shard=(rank, world)`: of that
    rank's share of the BAM only."""
    signals = rio.Pod5File(pod5_path)
    total = both = 0
    # the pass is bound by the inflate of the file: nothing else runs in this process yet, so the cores this RANK was given
    # inflate (dist.bind_rank's plan when there is one; RMR_BAM_INFLATE_THREADS, read by the reader itself, when the user set it)
    from . import dist as rdist

    if os.environ.get("RMR_BAM_INFLATE_THREADS"):
        threads = 0
    elif rdist.LAST_BINDING and rdist.LAST_BINDING.get("threads"):
        threads = max(2, 2 * int(rdist.LAST_BINDING["threads"]))
    else:
        threads = max(8, min(16, rio._eff_cpus()))
    batches = rio.iter_bam_raw_batches(bam_path, want_ref=False, batch=2048, shard=shard, light=True, inflate_threads=threads)  # flags and names only
    first = next(batches, None)
    import itertools

    for rb, _ in itertools.chain([first] if first is not None else [], batches):
        keep = np.nonzero((rb.flag & 0x900) == 0)[0] if skip_non_primary else np.arange(rb.n)
        total += int(keep.size)
        has_pi = (rb.has & 64) != 0
        no, po = rb.name_off.tolist(), rb.pi_off.tolist()
        for i in keep.tolist():
            rid = (rb.pi[po[i] : po[i + 1]] if has_pi[i] else rb.names[no[i] : no[i + 1]]).decode("latin-1")
            both += rid in signals
    return both, total


def extract_chunk_dataset(bam_path, pod5_path, out_path, mod_base, mod_base_control, motifs, focus_ref_pos,
                          chunk_context, min_samps_per_base, max_chunks_per_read, pa_scaling, sig_map_refiner,
                          kmer_context_bases, base_start_justify, offset, num_reads, num_extract_alignment_threads=1,
                          num_extract_chunks_threads=1, skip_non_primary=True, basecall_anchor=False, rev_sig=False,
                          save_every=100_000, skip_shuffle=False, reads_per_batch=256, engine=None, rank=0, world=1):
    """POD5 + BAM -> CoreRemoraDataset directory (prepare_train_data.py:124-276; the two worker-count arguments
    are accepted for signature compatibility and unused: the batch is the unit of parallelism here).  Returns
    (dataset, {reason: count}).

    `world` > 1 (one process per GPU or several, torch.distributed initialised by the caller - `python -m remora_amd dataset
    prepare --gpus N [--procs-per-gpu P]`): every rank extracts the chunks of its own share of the BAM (io.shard_of, as
    `infer` does) into `<out_path>.partRRR`; after a barrier rank 0 appends the parts' rows in rank order to the dataset at
    `out_path` - the rows of a single-process run, in its order - and shuffles once unless told not to.  The reference
    spreads the same work over worker processes behind queues (prepare_train_data.py:124-276).  Rank 0 returns the
    dataset, the others None; the reason counts are summed over the ranks."""
    import shutil

    from . import dist as rdist

    world, rank = int(world), int(rank)
    shard = (rank, world) if world > 1 else None
    if world > 1 and num_reads is not None:
        raise RemoraError("--num-reads names the FIRST reads of the file: not available when the file is split over ranks")
    import time as _time

    t_start = _time.perf_counter()
    num_both, num_records = count_reads(pod5_path, bam_path, skip_non_primary, shard=shard)
    t_counted = _time.perf_counter()
    if world > 1:
        shares = rdist.gather_objects((num_both, num_records))
        num_records, total_both = sum(x[1] for x in shares), sum(x[0] for x in shares)
    else:
        total_both = num_both
    if num_records == 0:
        raise RemoraError("No records found in BAM file.")
    num_reads = num_both if num_reads is None else min(int(num_reads), num_both)
    if total_both == 0:
        return None, {}
    max_seq_len = sum(chunk_context) // min_samps_per_base

    def new_dataset(path, n_reads):
        return CoreRemoraDataset(
            data_path=path, mode="w",
            metadata=DatasetMetadata(
                allocate_size=max_chunks_per_read * n_reads, max_seq_len=max_seq_len,
                mod_bases=[] if mod_base_control else [mod_base[0]], mod_long_names=[] if mod_base_control else [mod_base[1]],
                motif_sequences=[m.raw_motif for m in motifs], motif_offsets=[m.focus_pos for m in motifs],
                extra_arrays={"read_ids": ("<U36", "Read identifier"),
                              "read_focus_bases": ("int64", "Position within read training sequence")},
                chunk_context=chunk_context, kmer_context_bases=kmer_context_bases, reverse_signal=rev_sig,
                pa_scaling=pa_scaling, sig_map_refiner=sig_map_refiner, base_start_justify=base_start_justify, offset=offset))

    part_path = f"{str(out_path).rstrip('/')}.part{rank:03d}" if world > 1 else out_path
    dataset = new_dataset(part_path, max(num_reads, 1))
    errs = defaultdict(int)
    int_label = 0 if mod_base_control else 1
    next_save = save_every

    clock = {"ingest": 0.0, "extract": 0.0, "write": 0.0}  # RMR_INFER_TIMING=1 prints it

    def run(batch):
        nonlocal next_save
        t0 = _time.perf_counter()
        got = None
        if isinstance(batch, rio.IngestBatch):
            got = extract_chunk_arrays_from_ingest(batch, int_label, motifs, sig_map_refiner, max_chunks_per_read, chunk_context,
                                                   kmer_context_bases, base_start_justify, offset)
            if got is None:
                batch = batch.per_read()
        if got is None:
            got = extract_chunk_arrays_from_reads(
                batch, int_label, motifs, focus_ref_pos, sig_map_refiner, max_chunks_per_read, chunk_context,
                kmer_context_bases, base_start_justify, offset, basecall_anchor, engine)
        arrs, ids, keep, results = got
        for err, _rows in results:
            if err is not None:
                errs[err] += 1
        errs["No chunks extracted"] += len(batch) - len(results)  # reads dropped by RemoraRead.check (:204-206)
        if arrs is None:
            return
        errs["Sequence too long"] += int(((arrs.lengths.cpu().numpy() > max_seq_len) & keep).sum())
        t1 = _time.perf_counter()
        clock["extract"] += t1 - t0
        try:
            dataset.write_chunk_arrays(arrs, keep=keep, read_ids=ids)
        except RemoraError as e:
            errs[str(e)] += 1
        clock["write"] += _time.perf_counter() - t1
        if dataset.size >= next_save:
            dataset.flush()
            next_save += save_every

    refiner_iterative = (sig_map_refiner is not None and getattr(sig_map_refiner, "is_loaded", False) and sig_map_refiner.scale_iters > 0)
    # reference anchor, motif-selected focus bases, forward signal: the batch ingest of `infer --reference-anchored`
    # (io.iter_ingest_batches: the reads of a BAM batch assembled on the GPU) - everything else read by read
    if (not basecall_anchor and focus_ref_pos is None and not rev_sig and not refiner_iterative and
            os.environ.get("RMR_PREPARE_BATCH_INGEST", "1") != "0"):
        seen, t_loop = 0, _time.perf_counter()
        # (pa_scaling only travels in the dataset's metadata: training reads are scaled by sm / sd, prepare_train_data.py:66-72)
        dev_idx = engine.device if engine is not None else _torch().cuda.current_device()
        for ib in _prefetched(rio.iter_ingest_batches(pod5_path, bam_path, pa_scaling=None, skip_non_primary=skip_non_primary,
                                                      batch=reads_per_batch, shard=shard, device=dev_idx, ref_anchored=True), dev_idx):
            if seen >= num_reads:
                break
            if seen + len(ib) > num_reads:
                ib = ib.head(num_reads - seen) if isinstance(ib, rio.IngestBatch) else ib[: num_reads - seen]
            seen += len(ib)
            run(ib)
        if os.environ.get("RMR_INFER_TIMING"):
            import sys as _sys

            print(f"[prepare rank {rank}/{world}] {seen} reads: count_reads {t_counted - t_start:.2f}s, batches {_time.perf_counter() - t_loop:.2f}s "
                  f"of which extract {clock['extract']:.2f}s write {clock['write']:.2f}s (the rest: waiting for the ingest thread)",
                  file=_sys.stderr, flush=True)
    else:
        batch, seen = [], 0
        for read_err in rio.iter_reads_from_pod5_and_bam(pod5_path, bam_path, reverse_signal=rev_sig, pa_scaling=pa_scaling,
                                                         skip_non_primary=skip_non_primary, shard=shard,
                                                         device=engine.device if engine is not None else None):
            if seen >= num_reads:
                break
            seen += 1
            batch.append(read_err)
            if len(batch) >= reads_per_batch:
                run(batch)
                batch = []
        if batch:
            run(batch)
    errs = {k: v for k, v in errs.items() if v}
    if world == 1:
        dataset.write_metadata()
        if not skip_shuffle:
            dataset.shuffle()
        dataset.flush()
        return dataset, errs
    # ---- several ranks: the parts become one dataset ----
    dataset.write_metadata()
    dataset.flush()
    del dataset
    summed = defaultdict(int)
    for part_errs in rdist.gather_objects(errs):  # (also the barrier behind which every part is on disk)
        for k, v in part_errs.items():
            summed[k] += v
    rdist.barrier()
    out = None
    if rank == 0:
        out = new_dataset(out_path, total_both)
        for r in range(world):
            path = f"{str(out_path).rstrip('/')}.part{r:03d}"
            part = CoreRemoraDataset(data_path=path, mode="r")
            a0, a1 = int(part.metadata.dataset_start), int(part.metadata.dataset_end)
            for st in range(a0, a1, 100_000):
                out.write_batch({name: np.asarray(part.arrays[name][st : min(st + 100_000, a1)]) for name in out.array_names})
            del part
            shutil.rmtree(path)
        out.write_metadata()
        if not skip_shuffle:
            out.shuffle()
        out.flush()
    rdist.barrier()
    return out, dict(summed)
