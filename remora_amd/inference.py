"""Python API of the hot path — drop-in for `remora.inference.call_read_mods`
(src/remora/inference.py:661-712) and the per-label tally the multi-GPU runs reduce."""
import os
import queue as _queue
import threading
from collections import defaultdict as _defaultdict

import numpy as np

from . import RemoraError
from .constants import DEFAULT_BATCH_SIZE
from .engine import _torch
from .util import Motif, format_mm_ml_tags, softmax_axis1


def _native_call_read(read, model, model_metadata):
    """(nn_out f32[N,num_out], labels i64[N], pos i64[N]) of one read whose focus bases are set, through ONE native call
    (rmr_call_read: staging, signal normalisation, chunk geometry and rows, the network, logits back; one stream synchronisation) - what
    RemoraRead.prepare_batches + run_model do between them (src/remora/data_chunks.py:468-540)."""
    import ctypes

    from . import _lib as L
    from .data_chunks import _validated_int16_dacs

    focus = np.ascontiguousarray(read.focus_bases, dtype=np.int64).ravel()
    n = focus.size
    dacs = np.ascontiguousarray(_validated_int16_dacs(read))
    s2s = np.ascontiguousarray(read.seq_to_sig_map, dtype=np.int64).ravel()
    seq = np.ascontiguousarray(read.int_seq).ravel()
    if seq.dtype.kind not in "iu" or seq.dtype.itemsize not in (1, 2, 4, 8):
        seq = seq.astype(np.int64)
    if s2s.size != seq.size + 1:
        raise RemoraError(f"Invalid read: seq ({seq.size}) and mapping ({s2s.size}) sizes incompatible")
    cc, kcb = model_metadata["chunk_context"], model_metadata["kmer_context_bases"]
    rd = L.Read(dacs.ctypes.data, dacs.size, s2s.ctypes.data, seq.ctypes.data, seq.dtype.itemsize, 0, seq.size,
                float(read.shift), float(read.scale), focus.ctypes.data, n, int(cc[0]), int(cc[1]), int(kcb[0]), int(kcb[1]),
                int(bool(model_metadata["base_start_justify"])), int(model_metadata["offset"]))
    nn_out = np.empty((n, model.num_out), np.float32)
    pos = np.empty(n, np.int64)
    L.check(L.lib().rmr_call_read(model._h, ctypes.byref(rd), nn_out.ctypes.data, pos.ctypes.data))
    labels = np.full(n, -1, np.int64)
    if read.labels is not None:
        labels[:] = np.asarray(read.labels)[focus]
    return nn_out, labels, pos


def call_read_mods(read, model, model_metadata, batch_size=DEFAULT_BATCH_SIZE, focus_offset=None,
                   return_mm_ml_tags=False, return_mod_probs=False):
    """Call modified bases on one read; arguments and return values as in the reference:
    (nn_out f32[N,num_out], labels i64[N], pos i64[N]) by default;
    (probs f64[N,num_mods], labels, pos) with return_mod_probs; (MM str, ML array('B')) with
    return_mm_ml_tags; three empty arrays when the read yields no chunk (:698-699).
    With a HipModel the read goes through one native call (`_native_call_read`; `read.batches` is then not filled in -
    RMR_NATIVE_CALL_READ=0 selects prepare_batches + run_model, which give the same bits)."""
    from .engine import HipModel

    if focus_offset is None:
        read.set_motif_focus_bases([Motif(*m) for m in model_metadata["motifs"]])
    else:
        read.focus_bases = np.array([focus_offset])
    if isinstance(model, HipModel) and os.environ.get("RMR_NATIVE_CALL_READ", "1") != "0":
        read.batches = []
        read.refine_signal_mapping(model_metadata.get("sig_map_refiner"))
        if read.focus_bases is None or len(read.focus_bases) == 0:
            return np.array([]), np.array([]), np.array([])
        nn_out, labels, pos = _native_call_read(read, model, model_metadata)
    else:
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


def iter_call_reads_mods(read_batches, model, model_metadata, return_mod_probs=False):
    """call_reads_mods over a stream of read batches.  Yields (reads, results) per batch, in order, results as
    call_reads_mods returns them.  Without a loaded refiner the batches go through the three-thread pipeline of
    `_pipelined_parts` (staging, extraction and the per-read split of neighbouring batches run under each other's
    inference); with one, the host staging of batch k+1 (own thread, own HIP stream) runs under the GPU work of batch
    k; batches whose refiner re-scales iteratively (scale_iters > 0) are staged inline, because that refinement
    rewrites the reads first."""
    from concurrent.futures import ThreadPoolExecutor

    from .data_chunks import DeviceReads

    torch = _torch()
    refiner = model_metadata.get("sig_map_refiner")
    loaded = refiner is not None and getattr(refiner, "is_loaded", False)
    if not loaded and os.environ.get("RMR_READS_SUBBATCH", "512") != "0":
        yield from _pipelined_parts(read_batches, model, model_metadata, return_mod_probs)
        return
    inline = loaded and refiner.scale_iters > 0
    engine = getattr(model, "engine", None)
    upload_stream = None

    def stage(reads):
        nonlocal upload_stream
        if inline or len(reads) == 0:
            return None
        if upload_stream is None:
            upload_stream = torch.cuda.Stream(device=(engine.torch_device if engine is not None else None))
        with torch.cuda.stream(upload_stream):
            dr = DeviceReads(reads, engine)  # synchronises the upload stream before returning
        return dr

    it = iter(read_batches)
    with ThreadPoolExecutor(max_workers=1) as pool:
        try:
            cur = next(it)
        except StopIteration:
            return
        fut = pool.submit(stage, cur)
        while True:
            dr = fut.result()
            try:
                nxt = next(it)
                fut = pool.submit(stage, nxt)
            except StopIteration:
                nxt, fut = None, None
            yield cur, call_reads_mods(cur, model, model_metadata, return_mod_probs, device_reads=dr)
            if fut is None:
                return
            cur = nxt


_PIPE = {}  # GPU index -> the pipeline's torch streams (upload, 3 workers) and its two thread pools
_PIPE_LOCK = threading.Lock()


def _pipelined_parts(parts, model, model_metadata, return_mod_probs):
    """call_reads_mods for every batch of reads of the iterable `parts`, on five threads: two stage (gather into
    pinned memory + upload on the upload stream, two pinned buffers each taking turns), three take turns over the staged batches
    (motif scan, extraction, inference, per-read split), so that the kernels of one batch run under the host work of
    its neighbours.  Each engine serialises its GPU calls (one mutex per engine: extraction runs on a second engine
    with its own stream); every C call and every copy releases the GIL.  Yields (batch, results) in order, results
    identical to the unpipelined call.  At most five batches are resident and six in flight at a time.  A single-pass
    signal-mapping refiner (scale_iters <= 0) can run per batch inside the workers (opt-in, see call_reads_mods);
    iterative re-scaling (scale_iters > 0) rewrites the reads on the host first and never comes here."""
    import collections
    import queue
    from concurrent.futures import ThreadPoolExecutor

    from .data_chunks import DeviceReads
    from .engine import get_prep_engine

    torch = _torch()
    engine = getattr(model, "engine", None)
    tdev = engine.torch_device if engine is not None else None
    # extraction runs on a second engine (own stream): its small kernels and their host round trips do not queue behind
    # the inference of the neighbouring batch on the model's stream
    prep = get_prep_engine(engine.device if engine is not None else None)
    refiner = model_metadata.get("sig_map_refiner")
    if refiner is not None and getattr(refiner, "is_loaded", False):
        refiner._device_refiner(prep.device)  # created once, here, not by whichever worker comes first
    slots = threading.BoundedSemaphore(5)
    # the four torch streams and the threads are made once per GPU: torch's caching allocator keeps its free blocks per
    # stream (fresh streams on every call would turn every allocation of the call into a hipMalloc) and the stager's
    # pinned buffers belong to its thread
    key = prep.device
    with _PIPE_LOCK:
        if key not in _PIPE:
            streams = [torch.cuda.Stream(device=tdev) for _ in range(4)]
            # the worker streams are handed out through ONE queue per GPU: two generators alive at once (two threads in
            # call_reads_mods, interleaved iter_call_reads_mods iterators) share the three worker threads, and whichever
            # worker runs takes a stream nobody else holds; the stager threads queue their uploads on the one upload stream
            fs = queue.SimpleQueue()
            for st in streams[1:]:
                fs.put(st)
            _PIPE[key] = dict(streams=streams, free_streams=fs,
                              # a pool of one stager thread and a pool of two: a call picks one (below); every thread
                              # has pinned buffers of its own
                              stager={1: ThreadPoolExecutor(max_workers=1, thread_name_prefix="rmr-stage"),
                                      2: ThreadPoolExecutor(max_workers=2, thread_name_prefix="rmr-stage2")},
                              workers=ThreadPoolExecutor(max_workers=3, thread_name_prefix="rmr-work"))
            import atexit

            atexit.register(lambda p=_PIPE[key]: ([x.shutdown(wait=False) for x in p["stager"].values()], p["workers"].shutdown(wait=False)))
        pipe = _PIPE[key]
    upload = pipe["streams"][0]
    free_streams = pipe["free_streams"]

    # How many sub-batches are gathered into pinned memory side by side.  The gather runs at the rate of its copy threads
    # (70 GB/s with eight), well below the host memory's: with a 16-bit model, whose kernels need 4 us a read, ONE stager was
    # busy 0.7 of the call and two are worth +10 % (150 -> 165 k reads/s); the fp32 model is bound by its kernels (15 us a
    # read) and loses 9 % to a second stager's threads (64.5 -> 58.8 k reads/s: profiles/r05_ab_reads_*.log).
    n_stagers = int(os.environ.get("RMR_READS_STAGERS", "0")) or (2 if getattr(model, "dtype", "fp32") in ("bf16", "f16") else 1)
    stager = pipe["stager"][2 if n_stagers >= 2 else 1]

    def stage(part):
        slots.acquire()
        if len(part) == 0:
            return None
        with torch.cuda.stream(upload):
            return DeviceReads(part, prep, async_upload=True)  # the worker waits for the copy (wait_ready)

    def work(part, staged):
        # a torch stream per worker: its copies (.cpu() / .to(device)) then wait for this batch's work only, not for
        # the neighbour's inference on the model's stream
        mine = free_streams.get()
        try:
            dr = staged.result()
            if dr is None:
                return []
            with torch.cuda.stream(mine):
                return call_reads_mods(part, model, model_metadata, return_mod_probs, device_reads=dr)
        finally:
            free_streams.put(mine)
            slots.release()

    flight = collections.deque()
    try:
        for part in parts:
            flight.append((part, pipe["workers"].submit(work, part, stager.submit(stage, part))))
            if len(flight) >= 6:
                done, fut = flight.popleft()
                yield done, fut.result()
        while flight:
            done, fut = flight.popleft()
            yield done, fut.result()
    finally:
        for _, fut in flight:  # a failure or an abandoned generator: nothing of this call keeps running behind it
            try:
                fut.result()
            except Exception:  # noqa: BLE001
                pass


def _subbatch_cuts(n, sub, lead=None):
    """[(start, stop)] of the sub-batches of a batch of n reads: short ones first (the GPU starts after the staging of
    `sub / 4` reads instead of `sub`), whole ones in the middle, and the rest in two tapering pieces - what is left to do when
    the stager has finished is the work on the LAST sub-batch, so that one is small.  `lead`: sizes of the leading sub-batches
    (default sub / 4, sub / 2; the 16-bit models, whose kernels outrun the staging, start on sub / 8:
    profiles/r05_ab_reads_lead_cuts.log)."""
    cuts, pos = [], 0
    for size in (lead or (sub // 4, sub // 2)):
        size = max(1, int(size))
        if pos < n:
            cuts.append((pos, min(pos + size, n)))
            pos = cuts[-1][1]
    while n - pos > sub + sub // 2:
        cuts.append((pos, pos + sub))
        pos += sub
    left = n - pos
    if left > sub // 2:
        first = (left * 3 + 4) // 5
        cuts.append((pos, pos + first))
        pos += first
    if pos < n:
        cuts.append((pos, n))
    return cuts


def _call_reads_mods_pipelined(reads, sub, model, model_metadata, return_mod_probs):
    """One large batch through `_pipelined_parts`, cut into sub-batches by `_subbatch_cuts`."""
    out = []
    lead = (sub // 8, sub // 4, sub // 2) if getattr(model, "dtype", "fp32") in ("bf16", "f16") else None
    for _, res in _pipelined_parts((reads[a:b] for a, b in _subbatch_cuts(len(reads), sub, lead)), model, model_metadata, return_mod_probs):
        out.extend(res)
    return out


def call_reads_mods(reads, model, model_metadata, return_mod_probs=False, device_reads=None):
    """Batched form of call_read_mods for a list of RemoraRead objects: the reads are uploaded once, then the
    (optional) signal-mapping refinement, the motif scan, the chunk extraction and the fused inference all run
    on the resident arrays (the reference processes reads one by one in Python,
    src/remora/inference.py:62-137, 661-712).  Returns a list of per-read (nn_out | probs, labels, pos) tuples;
    pos ascending within a read.  `read.focus_bases` is left holding the read's motif hits."""
    from .data_chunks import DeviceReads, _extract_device, device_to_pinned_async

    if len(reads) == 0:
        return []
    motifs = [Motif(*m) for m in model_metadata["motifs"]]
    refiner = model_metadata.get("sig_map_refiner")
    loaded = refiner is not None and getattr(refiner, "is_loaded", False)
    sub = int(os.environ.get("RMR_READS_SUBBATCH", "512"))
    # with a loaded refiner the batch stays whole by default: the banded DP of a call costs one read's latency whatever the
    # batch size (18 ms for 2048 or for 512 reads of 5 kb), so sub-batches multiply it (RMR_READS_PIPELINE_REFINER=1 opts in)
    piped_refiner = loaded and refiner.scale_iters <= 0 and os.environ.get("RMR_READS_PIPELINE_REFINER") == "1"
    if device_reads is None and (not loaded or piped_refiner) and sub > 0 and len(reads) >= 2 * sub:
        return _call_reads_mods_pipelined(reads, sub, model, model_metadata, return_mod_probs)
    if loaded and refiner.scale_iters > 0:
        for err in refiner.refine_reads(reads):  # DP rounds interleaved with host re-scaling
            if err is not None:
                raise err
    dr = device_reads if device_reads is not None and not (loaded and refiner.scale_iters > 0) else \
        DeviceReads(reads, getattr(model, "engine", None))
    dr.wait_ready()
    if loaded and refiner.scale_iters <= 0 and refiner.do_rough_rescale:
        refiner.rough_rescale_device(dr, reads)  # sorts + gathers on the GPU, 19-point fits on the host
    if loaded and refiner.scale_iters == 0:
        refiner.refine_device_reads(dr, reads)  # one banded-DP pass on the resident arrays
    focus, foc_off = dr.motif_focus_bases(motifs)
    handoff = hasattr(model, "engine") and model.engine is not dr.engine  # extraction and network on engines of their own
    arrs, _ = _extract_device(dr, focus, foc_off, model_metadata["chunk_context"], model_metadata["kmer_context_bases"],
                              model_metadata["base_start_justify"], model_metadata["offset"], sync=not handoff)
    bounds = [int(x) for x in foc_off]  # per-read slices of the concatenated results (np.split costs 5 us a piece)
    if len(arrs) == 0:
        dr.engine.synchronize()
        for r in reads:
            r.focus_bases = np.zeros(0, np.int64)
        return [(np.array([]), np.array([]), np.array([])) for _ in reads]
    # the network is queued BEFORE anything is fetched back - and, when the extraction ran on an engine of its own, behind an
    # event of that engine's stream instead of a host wait: the copy of the focus bases (1.3 MB for 512 reads) runs under the
    # network's kernels instead of in front of them (profiles/r05_reads_timeline.md: 1.7 ms of GPU idle time per sub-batch)
    if handoff:
        model.engine.wait_for(dr.engine)
    out = model.infer_chunks(arrs.signal, arrs.sequence, arrs.mapping, arrs.lengths, arrs.kmer_context_bases)
    if handoff:
        dr.engine.synchronize()  # the focus bases are final (their kernel ran in front of the extraction's)
    # both results come back through pinned buffers of this thread with ONE wait: the copy of the focus bases is queued now
    # and crosses PCIe under the network's kernels (behind the next sub-batch's upload it took 1-2 ms of a worker's time)
    torch = _torch()
    h_focus = device_to_pinned_async(focus, 4)
    if hasattr(model, "engine"):
        model.engine.wait_submitted()  # the copy below may run on another stream than the engine's (pipelined callers)
    h_out = device_to_pinned_async(out, 3)
    torch.cuda.current_stream(out.device).synchronize()
    focus_host, out = h_focus.numpy().copy(), h_out.numpy().copy()
    for i, r in enumerate(reads):
        r.focus_bases = focus_host[bounds[i] : bounds[i + 1]]
    # the chunks' positions: the focus base after the model's offset, clipped into the read (data_chunks.py:443-446) - known
    # on the host, no second copy back
    last = np.repeat(np.diff(dr.seq_off) - 1, np.diff(foc_off))
    pos = np.clip(focus_host + int(model_metadata["offset"]), 0, last)
    if return_mod_probs:
        out = softmax_axis1(out)[:, 1:].astype(np.float64)
    labels, res = arrs.labels, []
    for i in range(len(reads)):
        a, b = bounds[i], bounds[i + 1]
        res.append((out[a:b], labels[a:b], pos[a:b]) if b > a else (np.array([]), np.array([]), np.array([])))
    return res


def infer_from_pod5_and_bam(pod5_path, in_bam_path, model, model_metadata, out_bam_path, num_reads=None,
                            reads_per_batch=256, reverse_signal=None, skip_non_primary=True, ref_anchored=False, prefetch=2,
                            rank=0, world=1, label_counts_out=None, bam_level=None, shard_future=None):
    """`remora infer from_pod5_and_bam` for one model or a list of models (one per canonical base, with a list of
    metadata dicts), basecall-anchored by default or reference-anchored
    (`--reference-anchored`: calls at reference positions, output records rewritten to `<len>M` + reference
    sequence) (src/remora/inference.py:462-641): every input alignment is written to `out_bam_path` with
    MM/ML tags from the model (records whose read cannot be called are written unchanged and
    counted by reason, as the reference does).  Reads are grouped into batches that go through
    ONE chunk extraction and ONE fused inference on the GPU.  Returns {reason: count, ...} with
    the key None counting successfully called reads.

    `world` > 1 (one process per GPU, torch.distributed initialised by the caller — `python -m remora_amd infer
    from_pod5_and_bam --gpus N`): rank r takes a contiguous share of the alignments (io.shard_of: by byte range of the
    file, nothing to coordinate; no reader process hands reads out, no read crosses a GPU boundary), writes `<out>.partRRR` (whole BGZF members; rank 0's carries the
    header), and after a barrier rank 0 joins the parts in rank order — the records keep the input order.  The per-label
    call counts go through the ONE collective of the path (dist.allreduce_counts: RCCL all-reduce of int64[num_out]);
    the per-reason read counts are summed over the ranks; every rank returns the global numbers.  `num_reads` then
    limits each rank's share.  `label_counts_out` (a dict) receives {can_base: int64[num_out] calls per label}.
    `bam_level`: zlib level of the output's BGZF members (None = 6, htslib's default; 1 costs a third of the CPU time for
    ~10 % larger files - deflate is the largest single host cost of a file-to-file run)."""
    from collections import Counter

    from . import dist as rdist
    from . import io as rio

    md0 = model_metadata[0] if isinstance(model_metadata, (list, tuple)) else model_metadata
    if reverse_signal is None:
        reverse_signal = bool(md0.get("reverse_signal", False))
    pa_scaling = md0.get("pa_scaling")
    stats = Counter()
    header = rio.read_bam_header_bytes(in_bam_path)
    world, rank = int(world), int(rank)
    shard = (shard_future if shard_future is not None else (rank, world)) if world > 1 else None  # future of io.bam_shard(...)
    part_path = f"{out_bam_path}.part{rank:03d}" if world > 1 else out_bam_path

    models = list(model) if isinstance(model, (list, tuple)) else [model]
    mds = list(model_metadata) if isinstance(model_metadata, (list, tuple)) else [model_metadata]
    if len(models) != len(mds):
        raise RemoraError("one metadata dict per model is required")

    import time as _time

    clock = {"prep": 0.0, "gpu": 0.0, "tags_write": 0.0, "wait_ingest": 0.0, "ingest": 0.0, "close": 0.0}  # RMR_INFER_TIMING=1 prints it

    def call(batch):
        """Main thread: read preparation and the GPU passes of one batch -> a job for emit()."""
        tq = _time.perf_counter()
        if isinstance(batch, rio.IngestBatch):  # arrays already on the GPU (io.iter_ingest_batches): nothing to prepare
            for e in batch.err:
                if e is not None:
                    stats[e] += 1
            clock["prep"] += _time.perf_counter() - tq
            tg = _time.perf_counter()
            # one pass per model over the SAME resident reads (no model of a multi-model batch refines: see batch_ingest below)
            per_model = [call_reads_mods(batch.reads, mdl, md, return_mod_probs=True, device_reads=batch.dr) if batch.good.size else []
                         for mdl, md in zip(models, mds)]
            batch.dr, batch.reads = None, []  # the results are host arrays: the batch's device memory is free for the next one
            clock["gpu"] += _time.perf_counter() - tg
            return batch, None, per_model
        items, good = [], []  # items: per input record None (callable: the next entry of `good`) or the record to copy
        for io_read, err in batch:
            if err is None:
                try:
                    good.append((io_read, io_read.into_remora_read(ref_anchored)))
                    items.append(None)
                    continue
                except RemoraError as e:
                    err = f"Read prep error: {e}"
            stats[err] += 1
            items.append(io_read.record)
        # one pass per model (the reference runs one model per canonical base, src/remora/inference.py:277-316);
        # every model works on its own copy of the reads because refinement rewrites their mappings
        per_model = []
        tg = _time.perf_counter()
        clock["prep"] += tg - tq
        if good:
            for mdl, md in zip(models, mds):
                reads = [rr.copy() for _, rr in good] if len(models) > 1 else [rr for _, rr in good]
                per_model.append(call_reads_mods(reads, mdl, md, return_mod_probs=True))
        clock["gpu"] += _time.perf_counter() - tg
        return items, good, per_model

    def batch_tags(results, md, seq_bytes, seq_off, mi=0, count=True):
        """MM / ML of the callable reads of a batch for model `mi` in one native call -> (mm, mm_off, ml, ml_off, has uint8[n_good]);
        `count`: also book the reads into `stats` (the one-model case; several models are booked by join_model_tags)."""
        from .util import format_mm_ml_tags_batch

        sizes = np.fromiter((r[2].size for r in results), np.int64, len(results))
        live = [r for r in results if r[2].size]
        pos = np.concatenate([r[2] for r in live]) if live else np.zeros(0, np.int64)
        probs = np.concatenate([r[0] for r in live]) if live else np.zeros((0, len(md["mod_bases"])))
        if pos.size:  # calls per label (0 = canonical): argmax over [1 - sum(p_mod), p_mod...], first maximum wins
            full = np.concatenate([1.0 - probs.sum(axis=1, keepdims=True), probs], axis=1)
            label_counts[mi] += np.bincount(full.argmax(axis=1), minlength=label_counts[mi].size)
        call_off = np.zeros(len(results) + 1, np.int64)
        np.cumsum(sizes, out=call_off[1:])
        mm, mm_off, ml, ml_off = format_mm_ml_tags_batch(seq_bytes, seq_off, pos, probs, call_off, md["mod_bases"], md["can_base"])
        has = (sizes > 0).astype(np.uint8)
        n_ok = int(has.sum())
        if count:
            stats[None] += n_ok
            if n_ok < len(results):
                stats[f"No {md['can_base']} mod calls"] += len(results) - n_ok
        return mm, mm_off, ml, ml_off, has

    def join_model_tags(parts):
        """The tags of several models for the same reads, joined per read in model order (the per-read path's "".join(mm_all) /
        ml_all.extend, src/remora/inference.py:429-459): parts = [(mm, mm_off, ml, ml_off, has)] per model -> one such tuple.
        A read counts as called when any model called it; one no model called is booked under all their reasons."""
        n_good = parts[0][4].size
        has = np.zeros(n_good, np.uint8)
        for p_ in parts:
            has |= p_[4]
        out = []
        for col in (0, 2):  # mm bytes, ml bytes
            lens = [np.diff(p_[col + 1]) for p_ in parts]
            off = np.zeros(n_good + 1, np.int64)
            np.cumsum(np.sum(lens, axis=0), out=off[1:])
            buf = np.zeros(max(int(off[-1]), 1), np.uint8)
            start = off[:-1].copy()
            for p_, ln in zip(parts, lens):
                src = np.asarray(p_[col], np.uint8)
                tot = int(ln.sum())
                if tot:
                    shift = np.repeat(start - p_[col + 1][:-1], ln)  # destination minus source position, per byte
                    idx = np.arange(tot, dtype=np.int64) + np.repeat(p_[col + 1][:-1], ln) - np.repeat(np.cumsum(ln) - ln, ln)
                    buf[idx + shift] = src[idx]
                start = start + ln
            out += [buf, off]
        n_ok = int(has.sum())
        stats[None] += n_ok
        if n_ok < n_good:
            stats[",".join(sorted(f"No {md['can_base']} mod calls" for md in mds))] += n_good - n_ok
        return out[0], out[1], out[2], out[3], has

    def spread(off_good, good_pos, n):
        """Offsets int64[n + 1] for all n records of a batch from those of its callable ones (positions good_pos, ascending):
        records in between own empty slices."""
        full = np.zeros(n + 1, np.int64)
        full[good_pos + 1] = off_good[1:]
        return np.maximum.accumulate(full)

    def emit(job, writer):
        """Writer thread: MM/ML tags and output records of one batch, every record at its place in the input order
        (uncallable reads and reads without calls are copied without modified-base tags)."""
        items, good, per_model = job
        tw = _time.perf_counter()
        clock["tags_write"] -= tw
        if isinstance(items, rio.IngestBatch):
            ib, n = items, len(items)
            rb = ib.rb
            has = np.zeros(n, np.uint8)
            if ib.good.size:
                if len(mds) == 1:
                    mm, mm_off, ml, ml_off, has_g = batch_tags(per_model[0], mds[0], ib.seq, ib.seq_off)
                else:
                    mm, mm_off, ml, ml_off, has_g = join_model_tags([batch_tags(per_model[mi], mds[mi], ib.seq, ib.seq_off, mi, count=False)
                                                                     for mi in range(len(mds))])
                has[ib.good] = has_g
                mm_off, ml_off = spread(mm_off, ib.good, n), spread(ml_off, ib.good, n)
            else:
                mm, ml, mm_off, ml_off = np.zeros(1, np.uint8), np.zeros(1, np.uint8), np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
            writer.write(rio.records_with_mod_tags_flat(rb.raw, rb.raw_off[ib.keep], rb.raw_off[ib.keep + 1] - rb.raw_off[ib.keep],
                                                        rb.tags_off[ib.keep], mm, mm_off, ml, ml_off, has,
                                                        ref_seq=ib.ref_fwd if ref_anchored else None, ref_off=ib.ref_fwd_off))
            clock["tags_write"] += _time.perf_counter()
            return
        n = len(items)
        good_pos = np.asarray([k for k, it in enumerate(items) if it is None], np.int64)
        if len(mds) == 1 and not ref_anchored:
            # one model, basecall-anchored (the common case): MM/ML strings and the rewritten records of the whole batch in
            # two native calls (rmr_format_mm_ml, rmr_records_with_mod_tags) - byte for byte the per-read Python path below
            has = np.zeros(n, np.uint8)
            if good:
                seqs = [io_read.seq for io_read, _ in good]
                seq_off = np.zeros(len(seqs) + 1, np.int64)
                np.cumsum([len(x) for x in seqs], out=seq_off[1:])
                mm, mm_off, ml, ml_off, has_g = batch_tags(per_model[0], mds[0], "".join(seqs).encode("latin-1"), seq_off)
                has[good_pos] = has_g
                mm_off, ml_off = spread(mm_off, good_pos, n), spread(ml_off, good_pos, n)
            else:
                mm, ml, mm_off, ml_off = np.zeros(1, np.uint8), np.zeros(1, np.uint8), np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
            it_good = iter(good)
            records = [next(it_good)[0].record if it is None else it for it in items]
            writer.write(rio.records_with_mod_tags_batch(records, mm, mm_off, ml, ml_off, has))
            clock["tags_write"] += _time.perf_counter()
            return
        k = -1
        for it in items:
            if it is not None:
                writer.write(rio.record_with_mod_tags(it, None, None))
                continue
            k += 1
            io_read, rr = good[k]
            import array

            mm_all, ml_all, errs = [], array.array("B"), []
            for mi, (md, results) in enumerate(zip(mds, per_model)):
                probs, _, pos = results[k]
                if pos.size == 0:
                    errs.append(f"No {md['can_base']} mod calls")
                    continue
                # calls per label (0 = canonical): argmax over [1 - sum(p_mod), p_mod...], first maximum wins
                full = np.concatenate([1.0 - probs.sum(axis=1, keepdims=True), probs], axis=1)
                label_counts[mi] += np.bincount(full.argmax(axis=1), minlength=label_counts[mi].size)
                mm, ml = format_mm_ml_tags(seq=io_read.ref_seq if ref_anchored else io_read.seq, poss=pos, probs=probs,
                                           mod_bases=md["mod_bases"], can_base=md["can_base"])
                mm_all.append(mm)
                ml_all.extend(ml)
            if not mm_all:
                stats[",".join(sorted(errs))] += 1
                writer.write(rio.record_with_mod_tags(io_read.record, None, None))
                continue
            stats[None] += 1
            fwd = None
            if ref_anchored:
                fwd = io_read.ref_seq if io_read.ref_reg.strand == "+" else rio.revcomp(io_read.ref_seq)
            writer.write(rio.record_with_mod_tags(io_read.record, "".join(mm_all), ml_all, ref_anchored_seq=fwd))
        clock["tags_write"] += _time.perf_counter()

    label_counts = [np.zeros(len(md["mod_bases"]) + 1, np.int64) for md in mds]

    refiner0 = mds[0].get("sig_map_refiner")
    iterative = refiner0 is not None and getattr(refiner0, "is_loaded", False) and refiner0.scale_iters > 0
    # forward signal, either anchor: the batch ingest (io.iter_ingest_batches) - trimming, move tables, scaling and the read
    # arrays of a whole BAM batch on the GPU, no Python object per read.  Several models (one per canonical base,
    # src/remora/inference.py:286,311-315) share the resident reads when none of them refines the mapping (refinement rewrites
    # it per model: those, reverse signal and iterative re-scaling go read by read)
    any_refiner = any(getattr(md.get("sig_map_refiner"), "is_loaded", False) for md in mds)
    batch_ingest = ((len(models) == 1 or not any_refiner) and not reverse_signal and not iterative
                    and os.environ.get("RMR_INFER_BATCH_INGEST", "1") != "0")

    def batches():
        if batch_ingest:
            seen = 0
            for ib in rio.iter_ingest_batches(pod5_path, in_bam_path, pa_scaling=pa_scaling, skip_non_primary=skip_non_primary,
                                              batch=reads_per_batch, shard=shard, device=models[0].engine.device,
                                              ref_anchored=ref_anchored):
                if num_reads is not None and seen + len(ib) >= num_reads:
                    left = num_reads - seen
                    if left > 0:
                        yield ib.head(left) if isinstance(ib, rio.IngestBatch) else ib[:left]
                    return
                seen += len(ib)
                yield ib
            return
        batch = []
        for i, item in enumerate(rio.iter_reads_from_pod5_and_bam(pod5_path, in_bam_path, reverse_signal=reverse_signal,
                                                                  pa_scaling=pa_scaling, skip_non_primary=skip_non_primary,
                                                                  decode_batch=reads_per_batch,
                                                                  parse_ref_align=ref_anchored, shard=shard,
                                                                  device=models[0].engine.device)):
            if num_reads is not None and i >= num_reads:
                break
            batch.append(item)
            if len(batch) >= reads_per_batch:
                yield batch
                batch = []
        if batch:
            yield batch

    # ingest (BGZF inflate, zstd, record parsing) of the next batches runs in a thread while the GPU works on the
    # current one - the role the reference gives to its reader / prepare processes (src/remora/inference.py:488-560)
    import threading

    q = _queue.Queue(maxsize=max(int(prefetch), 1))
    stop = threading.Event()

    def produce():
        try:
            import torch

            torch.cuda.set_device(models[0].engine.torch_device)  # a new thread starts on device 0
            ti = _time.perf_counter()
            for b in batches():
                clock["ingest"] += _time.perf_counter() - ti
                while not stop.is_set():
                    try:
                        q.put(b, timeout=0.1)
                        break
                    except _queue.Full:
                        continue
                if stop.is_set():
                    return
                ti = _time.perf_counter()
            q.put(None)
        except BaseException as e:  # noqa: BLE001 - handed to the consumer
            q.put(e)

    t = threading.Thread(target=produce, daemon=True)
    t.start()
    # tags + output of batch k run in a thread of their own while the main thread prepares batch k + 1 and waits for its
    # GPU passes (ctypes and the BGZF pool release the GIL); jobs are taken in order, so the output does not change
    wq = _queue.Queue(maxsize=2)
    werr = []

    def write_loop(writer):
        while True:
            job = wq.get()
            if job is None:
                return
            if werr:
                continue  # drain after a failure
            try:
                emit(job, writer)
            except BaseException as e:  # noqa: BLE001 - re-raised by the main thread
                werr.append(e)

    try:
        with rio.BamWriter(part_path, header if rank == 0 else b"", eof=world == 1, level=bam_level) as writer:
            wt = threading.Thread(target=write_loop, args=(writer,), daemon=True)
            wt.start()
            try:
                while True:
                    tq0 = _time.perf_counter()
                    b = q.get()
                    clock["wait_ingest"] += _time.perf_counter() - tq0
                    if b is None:
                        break
                    if isinstance(b, BaseException):
                        raise b
                    job = call(b)
                    if werr:
                        break
                    wq.put(job)
            finally:
                wq.put(None)
                wt.join()
            if werr:
                raise werr[0]
            tc = _time.perf_counter()
        clock["close"] = _time.perf_counter() - tc
    except BaseException:
        # a failed rank (e.g. a byte-range share whose guessed boundary the rank in front could not confirm) leaves no
        # `<out>.partNNN` behind; the launcher ends the other ranks, whose handlers remove theirs
        if world > 1 and os.path.exists(part_path):
            os.remove(part_path)
        raise
    finally:
        stop.set()
    if os.environ.get("RMR_INFER_TIMING"):
        import sys as _sys

        tm = os.times()  # CPU seconds of this process (all its threads) since start, children excluded
        print(f"[infer rank {rank}/{world}] " + " ".join(f"{k} {v:.2f}s" for k, v in clock.items()) +
              f" | process cpu user {tm.user:.2f}s sys {tm.system:.2f}s", file=_sys.stderr, flush=True)
    if world > 1:
        # every part is complete on disk; the ONE data collective: per-label call counts (int64[num_out] per model)
        flat = rdist.allreduce_counts(np.concatenate(label_counts))
        label_counts = np.split(flat, np.cumsum([c.size for c in label_counts])[:-1])
        merged = Counter()
        for st in rdist.gather_objects(dict(stats)):  # also the barrier behind which rank 0 may read the parts
            merged.update(st)
        stats = merged
        if rank == 0:
            rio.concat_bam_parts(out_bam_path, [f"{out_bam_path}.part{r:03d}" for r in range(world)])
        rdist.barrier()
    if label_counts_out is not None:
        for md, c in zip(mds, label_counts):
            label_counts_out[md["can_base"]] = np.asarray(c, np.int64)
    return dict(stats)


# ---------------------------------------------------------------------------------------------
# Batching stages of `remora infer`, same names and bookkeeping as
# the reference so that its queue plumbing can drive them unchanged.  The one deliberate
# difference: a batch carries the chunks' compact k-mer arrays (`PackedKmers`: int8 sequence,
# int16 mapping, int16 lengths = 72 B per chunk @C100) in the slot where the reference carries
# the dense one-hot `enc_kmers` (14,400 B per chunk): the one-hot is never materialised, the
# fused kernels expand it in LDS.  `PackedKmers.dense()` gives the reference's array when wanted.
# ---------------------------------------------------------------------------------------------


class PackedKmers:
    """Compact k-mer inputs of a batch of chunks (CoreRemoraDataset layout, data_chunks.py:942-948)."""

    __slots__ = ("sequence", "mapping", "lengths", "kmer_context_bases")

    def __init__(self, sequence, mapping, lengths, kmer_context_bases):
        self.sequence, self.mapping, self.lengths = sequence, mapping, lengths
        self.kmer_context_bases = tuple(int(x) for x in kmer_context_bases)

    def __len__(self):
        return int(self.lengths.shape[0])

    def __getitem__(self, sl):
        return PackedKmers(self.sequence[sl], self.mapping[sl], self.lengths[sl], self.kmer_context_bases)

    def dense(self):
        """float32 [n, 4*kmer_len, chunk_len] one-hot, as compute_encoded_kmer_batch returns it (GPU kernel)."""
        from .encoded_kmers import compute_encoded_kmer_batch

        return compute_encoded_kmer_batch(*self.kmer_context_bases, np.ascontiguousarray(self.sequence),
                                          np.ascontiguousarray(self.mapping), np.ascontiguousarray(self.lengths))


def _put_item(item, out_q):
    while True:
        try:
            return out_q.put(item, timeout=0.1)
        except _queue.Full:
            continue


def _queue_iter(in_q, num_proc=1):
    done = 0
    while done < num_proc:
        try:
            item = in_q.get(timeout=0.1)
        except _queue.Empty:
            continue
        if item is StopIteration:
            done += 1
        else:
            yield item


def prepare_reads(read_errs, models_metadata, ref_anchored=False):
    """[(io_read, err)] -> [(io_read, {can_base: chunk dict} | None, err)] (src/remora/inference.py:62-137).
    A chunk dict holds `signal` f32[n,1,L], `kmers` (PackedKmers) and `read_focus_bases` i64[n] as host
    arrays; chunk extraction runs on the GPU for the whole read at once instead of through an in-memory
    CoreRemoraDataset per read and model."""
    from .data_chunks import extract_chunk_arrays

    out = []
    for io_read, err in read_errs:
        if err is not None:
            out.append((io_read, None, err))
            continue
        try:
            remora_read = io_read.into_remora_read(ref_anchored)
        except RemoraError as e:
            out.append((io_read, None, f"Read prep error: {e}"))
            continue
        except Exception as e:  # noqa: BLE001  (the reference reports and carries on, :95-99)
            out.append((io_read, None, f"Unexpected error: {e}"))
            continue
        chunks = {}
        for md in models_metadata:
            rr = remora_read.copy()
            rr.set_motif_focus_bases([Motif(*m) for m in md["motifs"]])
            rr.refine_signal_mapping(md.get("sig_map_refiner"))
            arrs = None
            if rr.focus_bases is not None and len(rr.focus_bases):
                arrs, _ = extract_chunk_arrays([rr], md["chunk_context"], md["kmer_context_bases"],
                                               md["base_start_justify"], md["offset"])
            if arrs is None or len(arrs) == 0:
                out.append((io_read, None, f"No {md['can_base']} mod calls"))
                continue
            chunks[md["can_base"]] = {
                "signal": arrs.signal.cpu().numpy(),
                "kmers": PackedKmers(arrs.sequence.cpu().numpy(), arrs.mapping.cpu().numpy(), arrs.lengths.cpu().numpy(),
                                     md["kmer_context_bases"]),
                "read_focus_bases": arrs.read_focus_bases.cpu().numpy(),
            }
        out.append((io_read, chunks, None))
    return out


def prep_nn_input(read_errs):
    """:152-168 - the per-read chunk dicts are already network inputs here."""
    if len(read_errs) == 0:
        return [(None, None, "No valid mappings")]
    return [(io_read, None, err) if err is not None else (io_read, chunks, None) for io_read, chunks, err in read_errs]


def batch_reads(prepped_nn_inputs, batches_q, batch_size, models_metadata):
    """Pack per-read chunk dicts into batches of exactly `batch_size` chunks per canonical base and put
    `(can_base, sigs f32[B,1,L], kmers PackedKmers[B], read_pos i64[B], b_reads)` on `batches_q`;
    `b_reads` = [[io_read, b_st, b_en, err], ...] with the reference's meaning (b_st None: the read
    continues from the previous batch; b_en None: it continues into the next), :171-262."""
    md_dict = dict((md["can_base"], md) for md in models_metadata)
    can_bases = list(md_dict)

    class _Acc:
        def __init__(self, cb):
            L = md_dict[cb]["chunk_len"]
            self.sig = np.empty((batch_size, 1, L), dtype=np.float32)
            self.pos = np.empty(batch_size, dtype=int)
            self.seq = self.map = None
            self.len = np.zeros(batch_size, dtype=np.int16)
            self.kcb = None
            self.fill = 0
            self.reads = []

        def put(self, r_chunks, lo, hi):
            n = hi - lo
            k = r_chunks["kmers"]
            if self.seq is None or k.sequence.shape[1] > self.seq.shape[1] or k.mapping.shape[1] > self.map.shape[1]:
                # reads differ in their widest chunk: widen the batch arrays (sequence pads with -1, mapping with 0)
                sw = max(k.sequence.shape[1], 0 if self.seq is None else self.seq.shape[1])
                mw = max(k.mapping.shape[1], 0 if self.map is None else self.map.shape[1])
                seq = np.full((batch_size, sw), -1, dtype=np.int8)
                mp = np.zeros((batch_size, mw), dtype=np.int16)
                if self.seq is not None:
                    seq[: self.fill, : self.seq.shape[1]] = self.seq[: self.fill]
                    mp[: self.fill, : self.map.shape[1]] = self.map[: self.fill]
                self.seq, self.map = seq, mp
            self.kcb = k.kmer_context_bases
            self.sig[self.fill : self.fill + n] = r_chunks["signal"][lo:hi]
            self.pos[self.fill : self.fill + n] = r_chunks["read_focus_bases"][lo:hi]
            self.seq[self.fill : self.fill + n] = -1
            self.map[self.fill : self.fill + n] = 0
            self.seq[self.fill : self.fill + n, : k.sequence.shape[1]] = k.sequence[lo:hi]
            self.map[self.fill : self.fill + n, : k.mapping.shape[1]] = k.mapping[lo:hi]
            self.len[self.fill : self.fill + n] = k.lengths[lo:hi]
            self.fill += n

        def emit(self, cb):
            n = self.fill
            kmers = PackedKmers(self.seq[:n], self.map[:n], self.len[:n], self.kcb) if self.seq is not None else None
            _put_item((cb, self.sig[:n], kmers, self.pos[:n], self.reads), batches_q)

    acc = dict((cb, _Acc(cb)) for cb in can_bases)
    for read_nn_inputs in prepped_nn_inputs:
        for io_read, bases_chunks, err in read_nn_inputs:
            if err is not None:
                for cb in can_bases:
                    acc[cb].reads.append([io_read, None, None, err])
                continue
            for cb, r_chunks in bases_chunks.items():
                num_chunks = r_chunks["read_focus_bases"].size
                consumed = 0
                while acc[cb].fill + num_chunks - consumed >= batch_size:  # the read fills this batch up
                    b_st = acc[cb].fill if consumed == 0 else None
                    take = batch_size - acc[cb].fill
                    acc[cb].put(r_chunks, consumed, consumed + take)
                    acc[cb].reads.append([io_read, b_st, None, None])
                    acc[cb].emit(cb)
                    consumed += take
                    acc[cb] = _Acc(cb)
                b_st = acc[cb].fill if consumed == 0 else None
                acc[cb].put(r_chunks, consumed, num_chunks)
                acc[cb].reads.append([io_read, b_st, acc[cb].fill, None])
    for cb in can_bases:
        if acc[cb].fill > 0:
            acc[cb].emit(cb)
    _put_item(StopIteration, batches_q)


def run_model_batched(batches_q, called_batches_q, models, models_metadata, batch_size):
    """:277-316 - `models[can_base]` is a HipModel; a batch goes through the fused chunk-arrays -> logits path
    (one H2D of 472 B per chunk instead of 14.8 KB).  `nn_out` is a torch tensor, as in the reference."""
    for can_base, b_sigs, b_kmers, b_read_pos, b_reads in _queue_iter(batches_q):
        model = models[can_base]
        if isinstance(b_kmers, PackedKmers):
            nn_out = model.infer_chunks(b_sigs, b_kmers.sequence, b_kmers.mapping, b_kmers.lengths, b_kmers.kmer_context_bases)
        else:  # a dense one-hot from a reference-side producer
            import torch

            dev = next(model.parameters()).device
            nn_out = model(torch.from_numpy(np.ascontiguousarray(b_sigs)).to(dev),
                           torch.from_numpy(np.ascontiguousarray(b_kmers)).to(dev))
        if isinstance(nn_out, np.ndarray):
            import torch

            nn_out = torch.from_numpy(nn_out)
        _put_item((can_base, nn_out, b_read_pos, b_reads), called_batches_q)
    _put_item(StopIteration, called_batches_q)


def unbatch_reads(curr_read, b_nn_out, b_read_pos, b_reads):
    """Re-assemble per-read outputs from one called batch (:331-367).  Returns (completed reads, read still open)."""
    comp = []
    for io_read, b_st, b_en, err in b_reads:
        if err is not None:
            if curr_read is not None:
                comp.append(curr_read)
            comp.append((io_read, None, None, err))
            curr_read = None
        elif b_st is None:  # continues from the previous batch
            if curr_read is None:
                raise RemoraError("Unbatching encountered None read")
            if curr_read[0].read_id != io_read.read_id:
                raise RemoraError("Unbatching encountered mismatching reads")
            io_read, r_out, r_pos, _ = curr_read
            curr_read = (io_read, np.concatenate([r_out, b_nn_out[:b_en]], axis=0),
                         np.concatenate([r_pos, b_read_pos[:b_en]]), None)
        else:
            if curr_read is not None:
                comp.append(curr_read)
            curr_read = (io_read, b_nn_out[b_st:b_en], b_read_pos[b_st:b_en], None)
    return comp, curr_read


def unbatch(called_batches_q, called_reads_q, models_metadata):
    """Called batches -> `(io_read, [(can_base, nn_out, read_pos), ...], err)` once every model has finished
    a read (:370-415)."""

    def finished(reads):
        calls, errs = [], set()
        for can_base, (io_read, nn_out, r_pos, err) in reads:
            errs.add(err)
            if err is None:
                calls.append((can_base, nn_out, r_pos))
        r_err = None if any(e is None for e in errs) else ",".join(sorted(errs))
        return io_read, calls, r_err

    can_bases = [md["can_base"] for md in models_metadata]
    curr = dict((cb, None) for cb in can_bases)
    comp = _defaultdict(list)
    for can_base, nn_out, b_read_pos, b_reads in _queue_iter(called_batches_q):
        done, curr[can_base] = unbatch_reads(curr[can_base], nn_out.cpu().numpy(), b_read_pos, b_reads)
        for c in done:
            comp[c[0].read_id].append((can_base, c))
        for rid in [rid for rid, cs in comp.items() if len(cs) == len(can_bases)]:
            _put_item(finished(comp[rid]), called_reads_q)
            del comp[rid]
    if curr[can_bases[0]] is not None:
        _put_item(finished([(cb, curr[cb]) for cb in can_bases]), called_reads_q)
    _put_item(StopIteration, called_reads_q)


def post_process_reads(read_mapping, models_metadata, ref_anchored=False):
    """(io_read, mod_calls, err) -> (io_read, MM string, ML array) | (io_read, err): softmax, drop the canonical
    class, MM/ML formatting per model (:429-459).  The BAM record itself is rebuilt by io.record_with_mod_tags."""
    import array

    io_read, mod_calls, err = read_mapping
    if err is not None:
        return io_read, err
    md_dict = dict((md["can_base"], md) for md in models_metadata)
    mm_tags, ml_arr = [], array.array("B")
    for can_base, nn_out, r_poss in mod_calls:
        probs = softmax_axis1(nn_out)[:, 1:].astype(np.float64)
        seq = io_read.ref_seq if ref_anchored else io_read.seq
        mm, ml = format_mm_ml_tags(seq=seq, poss=r_poss, probs=probs, mod_bases=md_dict[can_base]["mod_bases"],
                                   can_base=can_base)
        mm_tags.append(mm)
        ml_arr.extend(ml)
    return io_read, "".join(mm_tags), ml_arr
