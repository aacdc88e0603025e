"""Read model and chunk extraction of the hot path — the inference half of
`remora.data_chunks` (src/remora/data_chunks.py): RemoraRead :126-540, Chunk :544-641.

The per-chunk Python of the reference (extract_chunk :331-423 called once per focus base,
write_chunk :1376-1418) is replaced by two kernel launches for the whole read
(rmr_chunk_geometry + rmr_chunk_fill); the arrays stay on the GPU in the CoreRemoraDataset
layout (:786-816) and feed `HipModel.infer_chunks` without ever materialising the one-hot.
"""
import ctypes
import dataclasses

import os

import numpy as np

from . import RemoraError
from . import _lib as L
from . import util
from .engine import HipModel, get_engine, _torch


@dataclasses.dataclass
class Chunk:
    """Host-side view of one extracted chunk (src/remora/data_chunks.py:544-641)."""

    signal: np.ndarray
    seq_w_context: np.ndarray
    seq_to_sig_map: np.ndarray
    kmer_context_bases: tuple
    chunk_sig_focus_idx: int
    chunk_focus_base: int
    read_focus_base: int
    read_id: str = None
    label: int = None

    @property
    def kmer_len(self):
        return sum(self.kmer_context_bases) + 1

    @property
    def seq_len(self):
        return self.seq_w_context.size - sum(self.kmer_context_bases)

    @property
    def seq(self):
        kb = self.kmer_context_bases[0]
        return self.seq_w_context[kb : kb + self.seq_len]

    def check(self):
        """Validity rules of the reference (:584-620)."""
        if self.signal.size <= 0:
            raise RemoraError("No signal for chunk")
        if np.any(np.isnan(self.signal)):
            raise RemoraError("Signal contains NaN")
        if self.seq_w_context.size - sum(self.kmer_context_bases) != self.seq_to_sig_map.size - 1:
            raise RemoraError("Invalid sig to seq map length")
        if self.seq_to_sig_map[0] < 0:
            raise RemoraError("Seq to sig map starts before 0")
        if self.seq_to_sig_map[-1] > self.signal.size:
            raise RemoraError("Seq to sig map ends after signal")


# BAM CIGAR operations (M I D N S H P = X) that align a base / consume the query / consume the reference
MATCH_OPS = np.array([True, False, False, False, False, False, False, True, True])
QUERY_OPS = np.array([True, True, False, False, True, False, False, True, True])
REF_OPS = np.array([True, False, True, True, False, False, False, True, True])


def map_ref_to_signal(*, query_to_signal, ref_to_query_knots):
    """Signal coordinate of every reference position: the (fractional) query coordinate of each
    reference base interpolated through the move table, floored (src/remora/data_chunks.py:60-74)."""
    return np.floor(np.interp(ref_to_query_knots, np.arange(query_to_signal.size), query_to_signal)).astype(int)


def make_sequence_coordinate_mapping(cigar):
    """Query coordinate (float) of every reference position 0..ref_len from the cigartuples: exact inside
    match runs, linearly interpolated across insertions / deletions (src/remora/data_chunks.py:77-115).
    `cigar`: [(op, length), ...] as pysam gives it, or the same as an int array of shape (n, 2) / a pair of arrays
    (ops, lengths) - what the native BAM reader holds; no Python tuple per operation then."""
    if isinstance(cigar, np.ndarray) and cigar.ndim == 2 and cigar.shape[1] == 2:
        ops, lens = cigar[:, 0].astype(np.int64), cigar[:, 1].astype(np.int64)
    elif isinstance(cigar, tuple) and len(cigar) == 2 and isinstance(cigar[0], np.ndarray):
        ops, lens = np.asarray(cigar[0], np.int64), np.asarray(cigar[1], np.int64)
    else:
        cigar = list(cigar)
        if not cigar:
            raise RemoraError("No match operations found in alignment cigar")
        ops, lens = (np.array(x) for x in zip(*cigar))
    if ops.size and (ops.min() < 0 or ops.max() > 8):
        raise RemoraError("Invalid cigar op(s)")
    matched = np.nonzero(MATCH_OPS[ops])[0] if ops.size else np.zeros(0, np.int64)
    if not matched.size:  # (the reference pops the non-match operations off the end until it finds a match or nothing)
        raise RemoraError("No match operations found in alignment cigar")
    ops, lens = ops[: matched[-1] + 1], lens[: matched[-1] + 1]
    if lens.min() < 0:
        raise RemoraError("Cigar lengths may not be negative")
    is_match = MATCH_OPS[ops]
    # two knots per match run: its first and its last base
    back = np.array([lens[is_match], np.ones_like(lens[is_match])])
    ref_end = np.cumsum(np.where(REF_OPS[ops], lens, 0))
    query_end = np.cumsum(np.where(QUERY_OPS[ops], lens, 0))
    ref_knots = np.concatenate([[0], (ref_end[is_match] - back).T.flatten(), [ref_end[-1]]])
    query_knots = np.concatenate([[0], (query_end[is_match] - back).T.flatten(), [query_end[-1]]])
    return np.interp(np.arange(ref_knots[-1] + 1), ref_knots, query_knots)


def compute_ref_to_signal(query_to_signal, cigar):
    """src/remora/data_chunks.py:118-122"""
    return map_ref_to_signal(query_to_signal=query_to_signal, ref_to_query_knots=make_sequence_coordinate_mapping(cigar))


class ChunkArrays:
    """Chunks of one or more reads as GPU-resident arrays in the CoreRemoraDataset layout
    (signal f32[n,1,L], sequence i8[n,W], sequence_to_signal_mapping i16[n,W'],
    sequence_lengths i16[n]) plus read_focus_bases / labels / geometry."""

    def __init__(self, signal, sequence, mapping, lengths, read_focus_bases, labels, geo,
                 kmer_context_bases, chunk_context):
        self.signal, self.sequence, self.mapping, self.lengths = signal, sequence, mapping, lengths
        self.read_focus_bases, self.labels, self.geo = read_focus_bases, labels, geo
        self.read_focus_bases_host = None  # filled in when the caller already knows them (saves a device read-back)
        self.kmer_context_bases = tuple(int(x) for x in kmer_context_bases)
        self.chunk_context = tuple(int(x) for x in chunk_context)

    def __len__(self):
        return int(self.lengths.shape[0])

    def enc_kmers(self):
        from .encoded_kmers import compute_encoded_kmer_batch

        return compute_encoded_kmer_batch(*self.kmer_context_bases, self.sequence, self.mapping, self.lengths)

    def as_reference_batch(self):
        """(signal, enc_kmers, labels, read_focus_bases) numpy tuple, the element type of
        RemoraRead.batches in the reference (src/remora/data_chunks.py:506-514)."""
        return (self.signal.cpu().numpy(), self.enc_kmers().cpu().numpy(), self.labels.copy(),
                self.read_focus_bases.cpu().numpy())

    def __iter__(self):  # tuple-unpacking compatibility
        return iter(self.as_reference_batch())


import threading

_PINNED = threading.local()  # one staging buffer per thread (the streaming API stages in a worker thread)


def _pinned_bytes(count, slot=0):
    """Grow-only pinned host staging buffer `slot` of this thread (pinned allocation is too slow to do per batch)."""
    torch = _torch()
    bufs = getattr(_PINNED, "bufs", None)
    if bufs is None:
        bufs = _PINNED.bufs = {}
    buf = bufs.get(slot)
    if buf is None or buf.numel() < count:
        buf = bufs[slot] = torch.empty(max(int(count * 1.25), 1 << 20), dtype=torch.uint8, pin_memory=True)
    return buf


def device_to_numpy(t):
    """`t.cpu().numpy()` through this thread's pinned staging buffer: a pageable destination makes the runtime bounce a
    large copy through its own small staging buffers (about 3 GB/s; pinned: PCIe rate).  The result is an ordinary
    pageable array (one host memcpy out of the staging buffer): callers keep per-read slices of it for as long as they
    like without holding page-locked memory, which is neither swappable nor ever trimmed by torch's host allocator."""
    torch = _torch()
    nbytes = t.numel() * t.element_size()
    if not t.is_cuda or nbytes < (1 << 20):
        return t.cpu().numpy()
    t = t.contiguous()
    host = _pinned_bytes(nbytes, slot=3)[:nbytes].view(t.dtype).view(t.shape)
    host.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return host.numpy().copy()


def device_to_pinned_async(t, slot):
    """Queue the copy of a CUDA tensor into this thread's pinned staging buffer `slot` on the current stream and return the
    pinned view WITHOUT waiting: the caller synchronises the stream once for several such copies and then takes
    `.numpy().copy()` of each (device_to_numpy = this + the wait, for one tensor)."""
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    host = _pinned_bytes(max(nbytes, 1), slot=slot)[:nbytes].view(t.dtype).view(t.shape)
    host.copy_(t, non_blocking=True)
    return host


def _pinned_slot_async():
    """(slot, event) for an upload that is NOT waited for by its issuer: two staging buffers of this thread take turns;
    the event of a slot is that of the last upload out of it, to be waited for before the buffer is written again."""
    torch = _torch()
    st = getattr(_PINNED, "turn", None)
    if st is None:
        st = _PINNED.turn = {"next": 1, "events": {}}
    slot = st["next"]
    st["next"] = 3 - slot  # 1 <-> 2 (slot 0 is the synchronous buffer)
    ev = st["events"].get(slot)
    if ev is not None:
        ev.synchronize()
    ev = st["events"][slot] = torch.cuda.Event()
    return slot, ev


def _validated_int16_dacs(r):
    """dacs of a read as the int16 values the device layout (rmr_reads.dacs) holds.  The reference's RemoraRead.sig
    (data_chunks.py:191-197) takes any dtype; here anything int16 represents exactly goes through (e.g. the float
    zeros of RemoraRead.test_read), anything else would be truncated or wrapped silently and is refused."""
    a = np.asarray(r.dacs).ravel()
    if a.dtype == np.int16:
        return a
    with np.errstate(invalid="ignore", over="ignore"):
        ai = a.astype(np.int16)
    if not np.array_equal(ai, a):
        raise RemoraError(
            f"read {getattr(r, 'read_id', '?')}: dacs ({a.dtype}) are not int16-representable (non-integer or out "
            "of range); pass raw int16 ADC values with shift/scale - already-normalised float signal is not "
            "supported by the GPU extraction path")
    return ai


_GLUE = None


def _glue():
    """_pyglue.so (csrc/pyglue.c) through ctypes.PyDLL - called with the GIL held, takes Python objects."""
    global _GLUE
    if _GLUE is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_pyglue.so")
        if not os.path.exists(path):
            raise RemoraError(f"{path} not found - build it first: python -c 'import __graft_entry__ as g; g.build()'")
        g = ctypes.PyDLL(path)
        g.rmr_py_collect_reads.restype = ctypes.c_int64
        g.rmr_py_collect_reads.argtypes = [ctypes.py_object, ctypes.c_int64] + [ctypes.c_void_p] * 8 + [ctypes.py_object]
        _GLUE = g
    return _GLUE


_SET_ORDER = None
_SET_SCRATCH = threading.local()


def _set_order_glue():
    """csrc/pyset_order.c through ctypes.CDLL (no CPython API in there: the GIL is released during the call), or False when
    its restatement of the interpreter's set does not reproduce THIS interpreter's iteration order on a fixed sample (checked
    once; callers then keep util.find_focus_bases_in_int_sequence)."""
    global _SET_ORDER
    if _SET_ORDER is None:
        if os.environ.get("RMR_PY_GLUE", "1") == "0":
            _SET_ORDER = False
            return _SET_ORDER
        _glue()  # existence check + message
        g = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_pyglue.so"))
        g.rmr_py_set_order.restype = ctypes.c_int64
        g.rmr_py_set_order.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        g.rmr_py_focus_bases_set_order.restype = ctypes.c_int64
        g.rmr_py_focus_bases_set_order.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32] + [ctypes.c_void_p] * 5 + [ctypes.c_int32]
        ok = True
        rng = np.random.RandomState(7)
        for n, top in ((3, 10), (5, 40), (40, 5000), (700, 6000), (3000, 1 << 20), (70000, 1 << 22)):
            keys = np.ascontiguousarray(rng.randint(0, top, n), np.int64)
            out = np.empty(n, np.int64)
            cnt = g.rmr_py_set_order(keys.ctypes.data, n, out.ctypes.data)
            s = set()
            s.update(keys.tolist())
            ok = ok and cnt == len(s) and out[:cnt].tolist() == list(s)
        _SET_ORDER = g if ok else False
        if not ok:
            # said once: the prepared datasets stay byte-identical (the interpreter's own set decides the order), at about a
            # third of the batch path's rate - on CPython 3.8-3.12 the native restatement matches and this is never reached
            import logging
            import sys

            logging.getLogger("Remora").info(
                "focus-base order: the native restatement of the interpreter's set (csrc/pyset_order.c) does not reproduce %s %s; "
                "`dataset prepare` keeps the interpreter's own set per read (same bytes, slower)", sys.implementation.name,
                sys.version.split()[0])
    return _SET_ORDER


def focus_bases_set_order(iseq, seq_off, motifs, threads=4):
    """util.find_focus_bases_in_int_sequence for every read of a batch in one native call (int8 base codes `iseq`, read g at
    seq_off[g]:seq_off[g+1]) -> (focus i64[F] read-local, in the interpreter's set order inside each read; foc_off i64[n+1]),
    or None when the native restatement is unavailable for these motifs / this interpreter."""
    g = _set_order_glue()
    if not g or not 1 <= len(motifs) <= 64:
        return None
    nm = len(motifs)
    ln = np.zeros(nm, np.int32)
    fp = np.zeros(nm, np.int32)
    mask = np.zeros((nm, 16), np.uint8)
    for m, mot in enumerate(motifs):
        if len(mot.raw_motif) > 16:
            return None
        ln[m], fp[m] = len(mot.raw_motif), int(mot.focus_pos)
        for k, allowed in enumerate(mot.int_pattern):
            mask[m, k] = int(sum(1 << int(b) for b in allowed))
    iseq = np.ascontiguousarray(iseq, np.int8)
    seq_off = np.ascontiguousarray(seq_off, np.int64)
    n = seq_off.size - 1
    # scratch with room for one entry per base, kept between calls (20 MB for a batch of 512 x 5 kb reads: touching fresh pages
    # every call cost more than the scan)
    need = max(int(seq_off[-1]), 1)
    focus = getattr(_SET_SCRATCH, "buf", None)
    if focus is None or focus.size < need:
        focus = _SET_SCRATCH.buf = np.empty(need + need // 4, np.int64)
    foc_off = np.zeros(n + 1, np.int64)
    tot = g.rmr_py_focus_bases_set_order(iseq.ctypes.data, seq_off.ctypes.data, n, nm, ln.ctypes.data, fp.ctypes.data,
                                         mask.ctypes.data, focus.ctypes.data, foc_off.ctypes.data, threads)
    if tot == -3:
        return None
    if tot < 0:
        raise MemoryError("rmr_py_focus_bases_set_order")
    return focus[:tot].copy(), foc_off


def _collect_reads(reads):
    """Per read: addresses of dacs / seq_to_sig_map / int_seq, their sizes, the bases' itemsize, shift and scale - what
    rmr_pack_reads gathers from.  -> (p_dacs u64[n], sig_n i64[n], p_maps u64[n], p_seqs u64[n], seq_n i64[n], itemsize i32[n],
    shift f64[n], scale f64[n], keep) where `keep` holds whatever had to be converted.  Reads whose arrays are already in the
    gather's layout (int16 dacs, int64 mapping, integer bases, all C-contiguous - what io.Read.into_remora_read and the
    reference's constructors produce) are walked through the C API (csrc/pyglue.c, 0.3 us a read); anything else by the
    interpreter, which converts what it can and refuses the rest with the messages of RemoraRead.check."""
    nr = len(reads)
    p_d, p_m, p_s = np.empty(nr, np.uint64), np.empty(nr, np.uint64), np.empty(nr, np.uint64)
    sig_n, seq_n, isz = np.empty(nr, np.int64), np.empty(nr, np.int64), np.empty(nr, np.int32)
    shift, scale = np.empty(nr, np.float64), np.empty(nr, np.float64)
    if nr and os.environ.get("RMR_PY_GLUE", "1") != "0":
        keep = []  # the glue appends every array whose address it hands back: owned here until the gather has copied from it
        got = _glue().rmr_py_collect_reads(reads if isinstance(reads, (list, tuple)) else list(reads), nr, p_d.ctypes.data,
                                           sig_n.ctypes.data, p_m.ctypes.data, p_s.ctypes.data, seq_n.ctypes.data, isz.ctypes.data,
                                           shift.ctypes.data, scale.ctypes.data, keep)
        if got == nr:
            return p_d, sig_n, p_m, p_s, seq_n, isz, shift, scale, keep
    keep = []
    for i, r in enumerate(reads):
        if r.seq_to_sig_map.size != r.int_seq.size + 1:
            raise RemoraError(f"Invalid read: seq ({r.int_seq.size}) and mapping ({r.seq_to_sig_map.size}) sizes incompatible")
        d = np.ascontiguousarray(_validated_int16_dacs(r))
        mp = np.ascontiguousarray(r.seq_to_sig_map, dtype=np.int64).ravel()
        sq = np.ascontiguousarray(r.int_seq).ravel()
        if sq.dtype.kind not in "iu" or sq.dtype.itemsize not in (1, 2, 4, 8):
            sq = sq.astype(np.int64)
        keep.append((d, mp, sq))
        p_d[i], p_m[i], p_s[i] = d.ctypes.data, mp.ctypes.data, sq.ctypes.data
        sig_n[i], seq_n[i], isz[i] = d.size, sq.size, sq.dtype.itemsize
        shift[i], scale[i] = float(r.shift), float(r.scale)
    return p_d, sig_n, p_m, p_s, seq_n, isz, shift, scale, keep


class DeviceReads:
    """The arrays of a batch of reads, concatenated and resident in HBM (the rmr_reads layout of
    include/remora_hip.h) - uploaded once and shared by the motif scan, the signal-mapping refinement
    and the chunk extraction.  All arrays travel in ONE pinned buffer / ONE copy (256-byte aligned segments);
    the device tensors are typed views of that allocation."""

    def __init__(self, reads, engine=None, async_upload=False, _narrow_maps=None):
        """`async_upload`: return once the copy is queued on the current torch stream (the staging buffer is one of two
        that take turns); whoever uses the arrays calls `wait_ready()` first.  Lets a staging thread gather batch k+1
        while batch k is still crossing PCIe.
        The mapping crosses PCIe as int32 (sample indices inside a read: a seventh of the batch's bytes less) and is widened to
        the int64 of the rmr_reads layout by a cast queued behind the copy on the same stream; a batch with a value that does
        not fit is gathered again as int64 (RMR_READS_NARROW_MAPS=0: always int64)."""
        torch = _torch()
        narrow = (os.environ.get("RMR_READS_NARROW_MAPS", "1") != "0") if _narrow_maps is None else bool(_narrow_maps)
        self._ready = None
        self.engine = engine if engine is not None else get_engine()
        dev = self.engine.torch_device
        nr = len(reads)
        self.n_reads = nr
        p_d, sig_n, p_m, p_s, seq_n, isz, shift, scale, keep = _collect_reads(reads)
        self.sig_off = np.zeros(nr + 1, np.int64)
        self.seq_off = np.zeros(nr + 1, np.int64)
        np.cumsum(sig_n, out=self.sig_off[1:])
        np.cumsum(seq_n, out=self.seq_off[1:])
        n_sig, n_seq = int(self.sig_off[-1]), int(self.seq_off[-1])
        segs = [("s2s", np.int32 if narrow else np.int64, n_seq + nr), ("d_sig_off", np.int64, nr + 1), ("d_seq_off", np.int64, nr + 1),
                ("shift", np.float64, nr), ("scale", np.float64, nr), ("dacs", np.int16, n_sig), ("iseq", np.int8, n_seq)]
        offs, total = {}, 0
        for name, dt, cnt in segs:
            offs[name] = total
            total += (cnt * np.dtype(dt).itemsize + 255) & ~255
        slot, ready = _pinned_slot_async() if async_upload else (0, None)
        buf = _pinned_bytes(max(total, 256), slot)
        host = buf.numpy()
        view = {name: host[offs[name] : offs[name] + cnt * np.dtype(dt).itemsize].view(dt) for name, dt, cnt in segs}
        # the per-read arrays are gathered into the pinned buffer by native threads (rmr_pack_reads); their addresses and
        # sizes come from _collect_reads (C API walk; a per-read numpy slice copy cost 25-30 us, the interpreter's walk 5-8)
        lib = L.lib()
        so, qo = np.empty(nr + 1, np.int64), np.empty(nr + 1, np.int64)
        if narrow:
            fit = ctypes.c_int(1)
            L.check(lib.rmr_pack_reads_narrow(nr, p_d.ctypes.data, sig_n.ctypes.data, p_m.ctypes.data, p_s.ctypes.data, seq_n.ctypes.data,
                                              isz.ctypes.data, view["dacs"].ctypes.data, view["s2s"].ctypes.data, view["iseq"].ctypes.data,
                                              so.ctypes.data, qo.ctypes.data, int(os.environ.get("RMR_PACK_THREADS", "8")), ctypes.byref(fit)))
            if not fit.value:  # (the slot's previous upload was waited for above; nothing of this one is queued yet)
                if ready is not None:
                    _PINNED.turn["next"] = slot  # hand the slot back: the retry takes it again
                    _PINNED.turn["events"].pop(slot, None)
                self.__init__(reads, engine, async_upload, _narrow_maps=False)
                return
        else:
            L.check(lib.rmr_pack_reads(nr, p_d.ctypes.data, sig_n.ctypes.data, p_m.ctypes.data, p_s.ctypes.data, seq_n.ctypes.data,
                                       isz.ctypes.data, view["dacs"].ctypes.data, view["s2s"].ctypes.data, view["iseq"].ctypes.data,
                                       so.ctypes.data, qo.ctypes.data, int(os.environ.get("RMR_PACK_THREADS", "8"))))
        del keep
        view["d_sig_off"][:] = self.sig_off
        view["d_seq_off"][:] = self.seq_off
        view["shift"][:] = shift
        view["scale"][:] = scale
        dbuf = buf[: max(total, 256)].to(dev, non_blocking=True)
        s2s_wide = None
        if narrow:  # widened on the stream of the copy, in front of the event / the wait below
            nb32 = (n_seq + nr) * 4
            s2s_wide = dbuf[offs["s2s"] : offs["s2s"] + nb32].view(torch.int32).to(torch.int64)
        if ready is not None:
            ready.record(torch.cuda.current_stream(dev))
            self._ready = ready
        else:
            torch.cuda.current_stream(dev).synchronize()  # the staging buffer is reused by the next batch
        for name, dt, cnt in segs:
            nbytes = cnt * np.dtype(dt).itemsize
            setattr(self, name, dbuf[offs[name] : offs[name] + nbytes].view(getattr(torch, np.dtype(dt).name)))
        if s2s_wide is not None:
            self.s2s = s2s_wide

    @classmethod
    def from_device(cls, engine, sig_off, seq_off, dacs, s2s, iseq, d_sig_off, d_seq_off, shift, scale):
        """A batch whose arrays were assembled on the GPU (io.iter_ingest_batches: rmr_assemble_reads): CUDA tensors in the
        rmr_reads layout, `sig_off` / `seq_off` the host copies of the offsets, `shift` / `scale` float64 host arrays."""
        torch = _torch()
        self = cls.__new__(cls)
        self._ready = None
        self.engine = engine
        self.n_reads = int(len(sig_off) - 1)
        self.sig_off, self.seq_off = np.ascontiguousarray(sig_off, np.int64), np.ascontiguousarray(seq_off, np.int64)
        self.dacs, self.s2s, self.iseq, self.d_sig_off, self.d_seq_off = dacs, s2s, iseq, d_sig_off, d_seq_off
        dev = engine.torch_device
        self.shift = torch.from_numpy(np.ascontiguousarray(shift, np.float64)).to(dev)
        self.scale = torch.from_numpy(np.ascontiguousarray(scale, np.float64)).to(dev)
        return self

    def wait_ready(self):
        """Host wait for an asynchronous upload (no-op otherwise)."""
        if self._ready is not None:
            self._ready.synchronize()
            self._ready = None

    def set_scaling(self, shift, scale):
        torch = _torch()
        dev = self.engine.torch_device
        self.shift = torch.from_numpy(np.asarray(shift, np.float64)).to(dev)
        self.scale = torch.from_numpy(np.asarray(scale, np.float64)).to(dev)

    def motif_focus_bases(self, motifs):
        """Focus bases of all motif hits, ascending inside each read: (focus i64[F] read-local, on the device;
        foc_off i64[n_reads+1] on the host).  GPU counterpart of RemoraRead.set_motif_focus_bases for a batch
        (src/remora/data_chunks.py:310-317)."""
        torch = _torch()
        ms = L.MotifSet()
        if not 1 <= len(motifs) <= 8:
            raise RemoraError("1..8 motifs supported")
        ms.n_motifs = len(motifs)
        for m, mot in enumerate(motifs):
            if len(mot.raw_motif) > 16:
                raise RemoraError("motifs longer than 16 bases are not supported on the GPU scan")
            ms.len[m] = len(mot.raw_motif)
            ms.focus_pos[m] = int(mot.focus_pos)
            for k, allowed in enumerate(mot.int_pattern):
                ms.mask[m][k] = int(sum(1 << int(b) for b in allowed))
        total = int(self.seq_off[-1])
        dev = self.engine.torch_device
        foc_off = np.zeros(self.n_reads + 1, np.int64)
        if not total:
            return torch.zeros(0, dtype=torch.int64, device=dev), foc_off
        lib = L.lib()
        # pass 1: hits per read (one wavefront per read); offsets on the host (they are needed there anyway: the results
        # are split per read); pass 2: read-local positions, ascending, compacted by wave ballots
        counts = torch.empty(self.n_reads, dtype=torch.int64, device=dev)
        L.check(lib.rmr_motif_focus_counts(self.engine.handle, self.iseq.data_ptr(), self.d_seq_off.data_ptr(), self.n_reads,
                                           ctypes.byref(ms), counts.data_ptr()))
        self.engine.synchronize()
        np.cumsum(counts.cpu().numpy(), out=foc_off[1:])
        n_hits = int(foc_off[-1])
        focus = torch.empty(max(n_hits, 1), dtype=torch.int64, device=dev)
        if n_hits:
            self.d_foc_off = torch.from_numpy(foc_off).to(dev)
            L.check(lib.rmr_motif_focus_fill(self.engine.handle, self.iseq.data_ptr(), self.d_seq_off.data_ptr(), self.n_reads,
                                             ctypes.byref(ms), self.d_foc_off.data_ptr(), focus.data_ptr()))
        return focus[:n_hits], foc_off


def _extract_device(dr, focus, foc_off, chunk_context, kmer_context_bases, base_start_justify, offset, labels=None, sync=True):
    """Chunk extraction for device-resident reads and focus bases (see extract_chunk_arrays).  `sync=False`: return with the
    fill kernel still queued on the reads' engine (`dr.engine`) - for a caller that hands the arrays to another engine with
    Engine.wait_for instead of a host wait; the kernels' inputs stay referenced by the returned arrays until those are dropped."""
    torch = _torch()
    eng, lib = dr.engine, L.lib()
    dev = eng.torch_device
    n_chunks = int(foc_off[-1])
    if isinstance(focus, np.ndarray):  # host focus bases: one upload for focus + offsets
        packed = torch.from_numpy(np.concatenate([np.ascontiguousarray(foc_off, np.int64),
                                                  focus.astype(np.int64, copy=False), np.zeros(1, np.int64)])).to(dev)
        d_foc_off, focus = packed[: foc_off.size], packed[foc_off.size :]
    else:
        d_foc_off = torch.from_numpy(np.ascontiguousarray(foc_off)).to(dev)
        if focus.numel() == 0:
            focus = torch.zeros(1, dtype=torch.int64, device=dev)
    rs = L.Reads(dr.n_reads, dr.dacs.data_ptr(), dr.d_sig_off.data_ptr(), dr.s2s.data_ptr(), dr.iseq.data_ptr(),
                 dr.d_seq_off.data_ptr(), dr.shift.data_ptr(), dr.scale.data_ptr(), focus.data_ptr(),
                 d_foc_off.data_ptr(), int(chunk_context[0]), int(chunk_context[1]),
                 int(kmer_context_bases[0]), int(kmer_context_bases[1]),
                 2 if base_start_justify == 2 else int(bool(base_start_justify)), int(offset))
    h_foc = np.ascontiguousarray(foc_off, np.int64)
    if dr.sig_off.dtype == np.int64 and dr.seq_off.dtype == np.int64:  # host copies of the offsets: no fetch per call
        rs.host_sig_off, rs.host_seq_off, rs.host_focus_off = dr.sig_off.ctypes.data, dr.seq_off.ctypes.data, h_foc.ctypes.data
    L_chunk = int(chunk_context[0]) + int(chunk_context[1])
    sig = torch.empty(max(int(dr.sig_off[-1]), 1), dtype=torch.float32, device=dev)
    geo = torch.empty((max(n_chunks, 1), 6), dtype=torch.int64, device=dev)
    max_sl = ctypes.c_int64(0)
    L.check(lib.rmr_chunk_geometry(eng.handle, ctypes.byref(rs), sig.data_ptr(), geo.data_ptr(),
                                   ctypes.byref(max_sl), L.MEM_DEVICE))
    sig = sig[: int(dr.sig_off[-1])]
    geo = geo[:n_chunks]
    msl = int(max_sl.value)
    seq_w = msl + int(kmer_context_bases[0]) + int(kmer_context_bases[1])
    map_w = msl + 1
    signal = torch.empty((n_chunks, 1, L_chunk), dtype=torch.float32, device=dev)
    sequence = torch.empty((n_chunks, max(seq_w, 1)), dtype=torch.int8, device=dev)
    mapping = torch.empty((n_chunks, max(map_w, 2)), dtype=torch.int16, device=dev)
    lengths = torch.empty(n_chunks, dtype=torch.int16, device=dev)
    rfb = torch.empty(n_chunks, dtype=torch.int64, device=dev)
    if n_chunks:
        L.check(lib.rmr_chunk_fill(eng.handle, ctypes.byref(rs), sig.data_ptr(), geo.data_ptr(), signal.data_ptr(),
                                   sequence.data_ptr(), sequence.shape[1], mapping.data_ptr(), mapping.shape[1],
                                   lengths.data_ptr(), rfb.data_ptr(), L.MEM_DEVICE))
    if labels is None:
        labels = np.full(n_chunks, -1, np.int64)
    arrs = ChunkArrays(signal, sequence, mapping, lengths, rfb, labels, geo, kmer_context_bases, chunk_context)
    if sync:
        eng.synchronize()  # the kernels above read the staged inputs: keep them alive until the stream drains
    else:
        arrs._inputs = (d_foc_off, focus, sig, dr)  # ... or for as long as the arrays they are turned into
    return arrs, sig


def extract_chunk_arrays(reads, chunk_context, kmer_context_bases, base_start_justify=False, offset=0,
                         engine=None):
    """Chunk arrays for a list of RemoraRead objects whose `focus_bases` are set.
    GPU counterpart of iter_chunks + extract_chunk + write_chunk for every focus base
    (src/remora/data_chunks.py:425-466, :331-423, :1376-1418).  Returns (ChunkArrays, sig)
    where sig is the normalised signal of all reads (float32, CUDA)."""
    torch = _torch()
    dr = DeviceReads(reads, engine)
    nr = len(reads)
    foc_off = np.zeros(nr + 1, np.int64)
    for i, r in enumerate(reads):
        foc_off[i + 1] = foc_off[i] + (0 if r.focus_bases is None else len(r.focus_bases))
    n_chunks = int(foc_off[-1])
    fl = [np.asarray(r.focus_bases).ravel() for r in reads if r.focus_bases is not None and len(r.focus_bases)]
    focus = np.concatenate(fl).astype(np.int64, copy=False) if fl else np.zeros(1, np.int64)
    labels = np.full(n_chunks, -1, np.int64)
    for i, r in enumerate(reads):
        if r.labels is not None and foc_off[i + 1] > foc_off[i]:
            labels[foc_off[i] : foc_off[i + 1]] = np.asarray(r.labels)[np.asarray(r.focus_bases)]
    arrs, sig = _extract_device(dr, focus[:n_chunks] if n_chunks else np.zeros(0, np.int64), foc_off, chunk_context,
                                kmer_context_bases, base_start_justify, offset, labels)
    if n_chunks:  # focus base after the offset, clipped into the read (data_chunks.py:443-446): known on the host
        last = np.repeat(np.diff(dr.seq_off) - 1, np.diff(foc_off))
        arrs.read_focus_bases_host = np.clip(focus[:n_chunks] + int(offset), 0, last)
    return arrs, sig


@dataclasses.dataclass
class RemoraRead:
    """Same fields and methods as the reference's RemoraRead for the inference path
    (src/remora/data_chunks.py:126-540)."""

    dacs: np.ndarray
    shift: float
    scale: float
    seq_to_sig_map: np.ndarray
    int_seq: np.ndarray = None
    str_seq: str = None
    read_id: str = None
    labels: np.ndarray = None
    focus_bases: np.ndarray = None
    batches: list = None

    def __post_init__(self):
        if self.int_seq is None:
            if self.str_seq is None:
                raise RemoraError("Must provide sequence to initialize RemoraRead")
            self.int_seq = util.seq_to_int(self.str_seq)
        else:
            self.int_seq = np.asarray(self.int_seq)
            self.str_seq = util.int_to_seq(self.int_seq)
        self.dacs = np.asarray(self.dacs)
        self.seq_to_sig_map = np.asarray(self.seq_to_sig_map)
        self._sig = None

    @classmethod
    def test_read(cls, nbases=20, signal_per_base=10):
        """Spoofed read (src/remora/data_chunks.py:178-189)."""
        return cls(np.zeros(nbases * signal_per_base), 0.0, 1.0,
                   np.arange(nbases * signal_per_base + 1, step=signal_per_base),
                   np.arange(nbases) % 4, "test_read", np.zeros(nbases, dtype=np.int64))

    @property
    def sig(self):
        """((dacs - shift) / scale).astype(float32), float64 arithmetic (:191-197); computed by
        the normalise kernel."""
        if self._sig is None:
            saved, self.focus_bases = self.focus_bases, None
            try:
                _, sig = extract_chunk_arrays([self], (1, 1), (0, 0))
            finally:
                self.focus_bases = saved
            self._sig = sig.cpu().numpy()
        return self._sig

    def check(self):
        """:222-249"""
        if self.seq_to_sig_map.size != self.int_seq.size + 1:
            raise RemoraError(f"Invalid read: seq ({self.int_seq.size}) and mapping "
                              f"({self.seq_to_sig_map.size}) sizes incompatible")
        if self.seq_to_sig_map[0] != 0:
            raise RemoraError("Invalid read: mapping start")
        if self.seq_to_sig_map[-1] != self.dacs.size:
            raise RemoraError("Invalid read: mapping end")
        if self.int_seq.max() > 3:
            raise RemoraError("Invalid read: Invalid base")
        if self.int_seq.min() < -1:
            raise RemoraError("Invalid read: Invalid base")

    def copy(self):
        return RemoraRead(
            dacs=self.dacs.copy(), shift=self.shift, scale=self.scale, seq_to_sig_map=self.seq_to_sig_map,
            int_seq=None if self.int_seq is None else self.int_seq.copy(), str_seq=self.str_seq,
            read_id=self.read_id, labels=None if self.labels is None else self.labels.copy(),
            focus_bases=None if self.focus_bases is None else self.focus_bases.copy())

    def refine_signal_mapping(self, sig_map_refiner, check_read=False):
        """Re-scale and re-map the read against the k-mer level table of the model (:267-308);
        no-op for refiners without a table.  The banded DP runs on the GPU
        (remora_amd.refine_signal_map.SigMapRefiner.refine_sig_map)."""
        if sig_map_refiner is None or not getattr(sig_map_refiner, "is_loaded", False):
            return
        if sig_map_refiner.do_rough_rescale:
            self.shift, self.scale = sig_map_refiner.rough_rescale(
                self.shift, self.scale, self.seq_to_sig_map, self.int_seq, self.dacs)
            self._sig = None
        if sig_map_refiner.scale_iters >= 0:
            self.seq_to_sig_map, self.shift, self.scale = sig_map_refiner.refine_sig_map(
                self.shift, self.scale, self.seq_to_sig_map, self.int_seq, self.dacs)
            self._sig = None
        if check_read:
            self.check()

    def set_motif_focus_bases(self, motifs):
        """:310-317"""
        self.focus_bases = util.find_focus_bases_in_int_sequence(self.int_seq, motifs)

    def downsample_focus_bases(self, max_sites):
        if self.focus_bases is not None and self.focus_bases.size > max_sites:
            self.focus_bases = np.random.choice(self.focus_bases, size=max_sites, replace=False)

    # ---- chunk extraction -------------------------------------------------------------
    def extract_chunk_arrays(self, chunk_context, kmer_context_bases, base_start_justify=False, offset=0,
                             motifs=None):
        fbs = self.focus_bases
        if motifs is not None and fbs is not None:
            fbs_arr = np.asarray(fbs, dtype=np.int64)
            on_motif = np.zeros(fbs_arr.size, bool)
            for m in motifs:
                on_motif |= m.match_many(self.int_seq, fbs_arr)
            saved, self.focus_bases = fbs, fbs_arr[on_motif]
            try:
                arrs, _ = extract_chunk_arrays([self], chunk_context, kmer_context_bases, base_start_justify, offset)
            finally:
                self.focus_bases = saved
            return arrs
        arrs, _ = extract_chunk_arrays([self], chunk_context, kmer_context_bases, base_start_justify, offset)
        return arrs

    def extract_chunk(self, focus_sig_idx, chunk_context, kmer_context_bases, label=-1, read_focus_base=-1, check_chunk=False,
                      signal_padding=False):
        """One chunk around a SIGNAL position, with the reference's arguments and return value (:331-423): the window
        `focus_sig_idx - chunk_context[0] .. + chunk_context[1]`, zero-padded where it leaves the read, the bases it covers
        with their k-mer context (-1 outside the read) and the mapping re-based to the chunk.  The same two kernels as the
        batch path (geometry + fill) for one focus position.  `signal_padding` (:357-363; no caller of the reference passes
        it): the zeros beside a read's end are replaced by the mirrored signal - the reference's two numpy assignments on this
        one chunk, taken from the normalised signal; a read too short to mirror from makes numpy refuse, as it does there."""
        dr = DeviceReads([self])
        arrs, _ = _extract_device(dr, np.array([int(focus_sig_idx)], np.int64), np.array([0, 1], np.int64), chunk_context,
                                  kmer_context_bases, 2, int(read_focus_base))
        sl = int(arrs.geo[0, 0])
        ctx = sum(arrs.kmer_context_bases)
        geo = arrs.geo.cpu().numpy()[0]
        chunk_sig = arrs.signal.cpu().numpy()[0, 0]
        if signal_padding:
            sig, chunk_len = self.sig, int(sum(chunk_context))
            sig_start, sig_end = int(focus_sig_idx) - int(chunk_context[0]), int(focus_sig_idx) + int(chunk_context[1])
            if not (sig_start >= 0 and sig_end <= sig.size):
                fill_st, fill_en, seq_to_sig_offset = 0, chunk_len, 0
                if sig_start < 0:
                    fill_st = seq_to_sig_offset = -sig_start
                    sig_start = 0
                if sig_end > sig.size:
                    fill_en = sig.size - sig_start + seq_to_sig_offset
                    sig_end = sig.size
                chunk_sig = chunk_sig.copy()
                chunk_sig[:fill_st] = sig[sig_start + fill_st : sig_start : -1]
                chunk_sig[fill_en:] = sig[sig_end : sig_end - chunk_sig.size + fill_en - 1 : -1]
        ch = Chunk(signal=chunk_sig, seq_w_context=arrs.sequence.cpu().numpy()[0, : sl + ctx],
                   seq_to_sig_map=arrs.mapping.cpu().numpy()[0, : sl + 1].astype(np.int32), kmer_context_bases=arrs.kmer_context_bases,
                   chunk_sig_focus_idx=int(geo[1]), chunk_focus_base=int(geo[2]), read_focus_base=int(read_focus_base),
                   read_id=self.read_id, label=label)
        if check_chunk:
            ch.check()
        return ch

    def iter_chunks(self, chunk_context, kmer_context_bases, base_start_justify=False, offset=0,
                    check_chunks=False, motifs=None):
        """Generator of host `Chunk` objects, same arguments as the reference (:425-466); all
        chunks are extracted on the GPU in one go and sliced here."""
        arrs = self.extract_chunk_arrays(chunk_context, kmer_context_bases, base_start_justify, offset, motifs)
        n = len(arrs)
        if n == 0:
            return
        sig = arrs.signal.cpu().numpy()[:, 0]
        seqs = arrs.sequence.cpu().numpy()
        maps = arrs.mapping.cpu().numpy()
        geo = arrs.geo.cpu().numpy()
        ctx = sum(arrs.kmer_context_bases)
        for i in range(n):
            sl = int(geo[i, 0])
            ch = Chunk(signal=sig[i].copy(), seq_w_context=seqs[i, : sl + ctx].copy(),
                       seq_to_sig_map=maps[i, : sl + 1].astype(np.int32),
                       kmer_context_bases=arrs.kmer_context_bases, chunk_sig_focus_idx=int(geo[i, 1]),
                       chunk_focus_base=int(geo[i, 2]), read_focus_base=int(geo[i, 3]), read_id=self.read_id,
                       label=int(arrs.labels[i]))
            if check_chunks:
                try:
                    ch.check()
                except RemoraError:
                    continue
            yield ch

    def prepare_batches(self, model_metadata, batch_size=None):
        """:468-514.  `batch_size` is accepted and ignored, as in the reference (it is not
        forwarded there either)."""
        self.batches = []
        self.refine_signal_mapping(model_metadata.get("sig_map_refiner"))
        if self.focus_bases is None or len(self.focus_bases) == 0:
            return
        arrs = self.extract_chunk_arrays(model_metadata["chunk_context"], model_metadata["kmer_context_bases"],
                                         model_metadata["base_start_justify"], model_metadata["offset"])
        if len(arrs) == 0:
            return
        self.batches.append(arrs)

    def run_model(self, model):
        """:516-540 -> (nn_out f32[N,num_out], labels i64[N], pos i64[N])."""
        torch = _torch()
        outs, labs, poss = [], [], []
        for arrs in self.batches:
            if isinstance(model, HipModel):
                out = model.infer_chunks(arrs.signal, arrs.sequence, arrs.mapping, arrs.lengths,
                                         arrs.kmer_context_bases)
            else:  # any other callable with the reference's model(sigs, enc_kmers) contract
                device = next(model.parameters()).device
                out = model(arrs.signal.to(device), arrs.enc_kmers().to(device)).detach()
            outs.append(out.cpu().numpy())
            labs.append(arrs.labels)
            poss.append(arrs.read_focus_bases_host if arrs.read_focus_bases_host is not None
                        else arrs.read_focus_bases.cpu().numpy())
        return np.concatenate(outs, axis=0), np.concatenate(labs), np.concatenate(poss)


# =======================================================================================
# On-disk chunk datasets (read side): the directory format of the
# reference's CoreRemoraDataset (src/remora/data_chunks.py:926-1702) — `metadata.jsn` plus raw
# C-contiguous memmaps `signal.npy` f32[N,1,L], `sequence.npy` i8[N,W], `sequence_to_signal_
# mapping.npy` i16[N,W'], `sequence_lengths.npy` i16[N], `labels.npy` i64[N] (no .npy header,
# :1280-1311).  Rows go to the GPU as they are; a smaller model context than the stored one is
# applied per batch exactly as the reference does (trim_sb_kmer_context_bases :1512-1534,
# trim_sb_chunk_context :1536-1576 -> the T1 kernel).
# =======================================================================================
import hashlib
import json
import os
from copy import deepcopy
from glob import glob

from . import constants

DATASET_VERSION = 3


@dataclasses.dataclass
class DatasetMetadata:
    """What a chunk dataset says about itself: the reference's DatasetMetadata (src/remora/data_chunks.py:645-888),
    same field names, defaults, derived properties and metadata.jsn text.  Items can also be read and written
    with [] (md["dataset_end"])."""

    allocate_size: int
    max_seq_len: int
    mod_bases: list
    mod_long_names: list
    motif_sequences: list
    motif_offsets: list
    dataset_start: int = 0
    dataset_end: int = 0
    version: int = DATASET_VERSION
    modified_base_labels: bool = True
    extra_arrays: dict = None
    chunk_context: tuple = constants.DEFAULT_CHUNK_CONTEXT
    base_start_justify: bool = False
    offset: int = 0
    kmer_context_bases: tuple = constants.DEFAULT_KMER_CONTEXT_BASES
    reverse_signal: bool = False
    pa_scaling: tuple = None
    sig_map_refiner: object = None
    rough_rescale_method: str = "least_squares"
    _stored_kmer_context_bases: tuple = None
    _stored_chunk_context: tuple = None

    def __post_init__(self):
        if isinstance(self.mod_bases, str):
            self.mod_bases = list(self.mod_bases)
        self.mod_bases = [str(mb) for mb in self.mod_bases]
        self.mod_long_names = list(self.mod_long_names)
        if len(self.mod_bases) != len(self.mod_long_names):
            raise RemoraError(f"mod_bases ({self.mod_bases}) must be the same length as mod_long_names "
                              f"({self.mod_long_names})")
        self.chunk_context = tuple(int(x) for x in self.chunk_context)
        self.kmer_context_bases = tuple(int(x) for x in self.kmer_context_bases)
        if self._stored_chunk_context is not None:
            self._stored_chunk_context = tuple(self._stored_chunk_context)
        if self._stored_kmer_context_bases is not None:
            self._stored_kmer_context_bases = tuple(self._stored_kmer_context_bases)
        self.check_motifs()

    # dict-style access
    def __getitem__(self, key):
        if key.startswith("refine_"):
            return self._refiner_dict()[key]
        return getattr(self, key)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def get(self, key, default=None):
        return getattr(self, key, default)

    def check_motifs(self):
        motifs = [util.Motif(*m) for m in self.motifs]
        ambiguous = [m for m in motifs if m.focus_base not in "ACGT"]
        if ambiguous:
            raise RemoraError(f"Cannot create dataset at motifs with ambiguous bases {ambiguous}")
        focus = {m.focus_base for m in motifs}
        if len(focus) > 1:
            raise RemoraError(f"Cannot create dataset with multiple motif focus bases: {focus}")

    chunk_width = property(lambda s: sum(s.chunk_context))
    stored_chunk_context = property(lambda s: s.chunk_context if s._stored_chunk_context is None else s._stored_chunk_context)
    stored_chunk_width = property(lambda s: sum(s.stored_chunk_context))
    chunk_context_adjusted = property(lambda s: s.stored_chunk_context != s.chunk_context)
    kmer_len = property(lambda s: sum(s.kmer_context_bases) + 1)
    stored_kmer_context_bases = property(
        lambda s: s.kmer_context_bases if s._stored_kmer_context_bases is None else s._stored_kmer_context_bases)
    kmer_context_bases_adjusted = property(lambda s: s.stored_kmer_context_bases != s.kmer_context_bases)
    size = property(lambda s: s.dataset_end - s.dataset_start)
    labels = property(lambda s: ["control"] + list(s.mod_long_names))
    num_labels = property(lambda s: len(s.mod_long_names) + 1)
    motifs = property(lambda s: list(zip(s.motif_sequences, s.motif_offsets)))
    num_motifs = property(lambda s: len(s.motif_sequences))
    extra_array_names = property(lambda s: [] if s.extra_arrays is None else list(s.extra_arrays))
    sequence_width = property(lambda s: s.max_seq_len + sum(s.stored_kmer_context_bases))
    sequence_to_signal_mapping_width = property(lambda s: s.max_seq_len + 1)
    signal_shape = property(lambda s: (s.allocate_size, 1, s.stored_chunk_width))
    sequence_shape = property(lambda s: (s.allocate_size, s.sequence_width))
    sequence_to_signal_mapping_shape = property(lambda s: (s.allocate_size, s.sequence_to_signal_mapping_width))
    sequence_lengths_shape = property(lambda s: (s.allocate_size,))
    labels_shape = property(lambda s: (s.allocate_size,))
    extras_shape = property(lambda s: (s.allocate_size,))

    @property
    def extra_array_dtypes_and_shapes(self):
        return [] if self.extra_arrays is None else [(n, dt, self.extras_shape) for n, (dt, _) in self.extra_arrays.items()]

    def _refiner_dict(self):
        from .refine_signal_map import SigMapRefiner

        return (self.sig_map_refiner if self.sig_map_refiner is not None else SigMapRefiner()).asdict()

    def asdict(self):
        """Field order of metadata.jsn: the dataclass fields without the refiner, then the refiner's own
        entries (its rough_rescale_method replaces the field's value in place), :847-852."""
        d = {f.name: getattr(self, f.name) for f in dataclasses.fields(self) if f.name != "sig_map_refiner"}
        d = deepcopy(d)
        if self.sig_map_refiner is not None:
            d.update(self.sig_map_refiner.asdict())
        return d

    def copy(self):
        return deepcopy(self)

    def write(self, metadata_path, kmer_table_path=None):
        """metadata.jsn (+ kmer_table.npy when a level table is attached), :857-888."""
        d = self.asdict()
        if d.get("refine_kmer_levels") is not None:
            if kmer_table_path is not None:
                np.save(kmer_table_path, np.asarray(d["refine_kmer_levels"]), allow_pickle=False)
            del d["refine_kmer_levels"]

        def enc(o):
            if isinstance(o, np.integer):
                return int(o)
            if isinstance(o, np.floating):
                return float(o)
            if isinstance(o, np.bool_):
                return bool(o)
            if isinstance(o, np.ndarray):
                return o.tolist()
            raise TypeError(f"{type(o)} is not JSON serialisable")

        with open(metadata_path, "w") as fh:
            json.dump(d, fh, default=enc)


def dataset_metadata(allocate_size, max_seq_len, mod_bases, mod_long_names, motif_sequences, motif_offsets,
                     chunk_context=(50, 50), kmer_context_bases=(4, 4), base_start_justify=False, offset=0,
                     reverse_signal=False, pa_scaling=None, extra_arrays=None, sig_map_refiner=None,
                     modified_base_labels=True, rough_rescale_method="least_squares"):
    """DatasetMetadata for a new dataset; a refiner is always attached (an unloaded one by default), because a
    dataset without the refine_* keys in metadata.jsn cannot be read back."""
    from .refine_signal_map import SigMapRefiner

    return DatasetMetadata(
        allocate_size=int(allocate_size), max_seq_len=int(max_seq_len), mod_bases=mod_bases, mod_long_names=mod_long_names,
        motif_sequences=list(motif_sequences), motif_offsets=[int(x) for x in motif_offsets],
        modified_base_labels=bool(modified_base_labels), extra_arrays=extra_arrays, chunk_context=chunk_context,
        base_start_justify=bool(base_start_justify), offset=int(offset), kmer_context_bases=kmer_context_bases,
        reverse_signal=bool(reverse_signal), pa_scaling=pa_scaling,
        sig_map_refiner=sig_map_refiner if sig_map_refiner is not None else SigMapRefiner(),
        rough_rescale_method=rough_rescale_method)


def check_super_batch(super_batch, chunk_width):
    """Row sanity of a super batch (:891-923): positive lengths, mappings inside [0, chunk_width] ending at
    chunk_width, monotonic inside each chunk, bases in [-1, 3]."""
    lens = super_batch["sequence_lengths"].astype(np.int64)
    if not np.all(lens) > 0:  # (sic) the reference compares the all() result, not the elements
        raise RemoraError("Sequence lengths must all be positive.")
    maps = super_batch["sequence_to_signal_mapping"]
    valid = np.arange(maps.shape[1]) < (lens[:, None] + 1)
    flat = maps[valid]
    if flat.max() > chunk_width:
        raise RemoraError("Signal mapping extend beyond chunk width")
    if flat.min() < 0:
        raise RemoraError("Signal mapping cannot contain negative values")
    if not np.all(maps[np.arange(lens.size), lens] == chunk_width):
        raise RemoraError("Chunk does not end at chunk_width")
    inner = np.ones(flat.size - 1, dtype=bool)
    ends = np.cumsum(lens)
    inner[ends[:-1] + np.arange(ends.size)[:-1]] = False  # steps from one chunk's end to the next chunk's start
    if np.diff(flat)[inner].min() < 0:
        raise RemoraError("Sequence to signal mappings are not monotonic")
    bases = super_batch["sequence"][np.arange(super_batch["sequence"].shape[1]) < lens[:, None]]
    if bases.max() > 3:
        raise RemoraError("Sequence max must be less than 4")
    if bases.min() < -1:
        raise RemoraError("Sequence min must greater tha -2")


class CoreRemoraDataset:
    """On-disk chunk dataset in the reference's format (src/remora/data_chunks.py:926-1702): five raw
    memmapped core arrays `<array>.npy` (no npy header), optional `extra_<name>.npy`, `metadata.jsn`
    (+ `kmer_table.npy`).  Same constructor arguments and iteration scheme (super batches, wrap-around for
    infinite iteration, random sub-sampling, label conversion, dynamic chunk / k-mer context trimming) as
    the reference; batches carry the core rows themselves (the fused GPU path consumes them directly) and
    `enc_kmers` on request through the encode kernel.  mode "w" appends what the extraction kernels produce."""

    _core_dtypes = {"signal": np.float32, "sequence": np.int8, "sequence_to_signal_mapping": np.int16,
                    "sequence_lengths": np.int16, "labels": np.int64}
    _core_arrays = list(_core_dtypes)

    def __init__(self, data_path=None, mode="r", metadata=None, override_metadata=None,
                 batch_size=constants.DEFAULT_BATCH_SIZE, super_batch_size=constants.DEFAULT_SUPER_BATCH_SIZE,
                 super_batch_sample_frac=None, super_batch_offset=0, infinite_iter=True, do_check_super_batches=False):
        self.data_path, self.mode, self.metadata = data_path, mode, metadata
        self.override_metadata = override_metadata
        self.batch_size, self.super_batch_size = int(batch_size), int(super_batch_size)
        self.super_batch_sample_frac, self.super_batch_offset = super_batch_sample_frac, int(super_batch_offset)
        self.infinite_iter, self.do_check_super_batches = bool(infinite_iter), bool(do_check_super_batches)
        self.label_conv = None
        self.arrays = {}
        self._iter = None
        if mode not in ("r", "w"):
            raise RemoraError("mode must be 'r' or 'w'")
        if data_path is None:
            if mode != "w" or not isinstance(metadata, DatasetMetadata):
                raise RemoraError("In-memory dataset must have mode='w' and metadata")
            self.allocate_arrays()
        elif mode == "r":
            self.data_path = util.resolve_path(data_path)
            self.load_metadata()
        else:
            if not isinstance(metadata, DatasetMetadata):
                raise RemoraError("Must provide metadata for new dataset")
            if override_metadata:
                raise RemoraError("Cannot override metadata of a dataset opened for writing")
            self.data_path = util.resolve_path(data_path)
            os.makedirs(self.data_path, exist_ok=True)
            self.allocate_arrays()
            self.write_metadata()
        self.refresh_memmaps()

    # ---- files ------------------------------------------------------------------------------
    @staticmethod
    def dataset_paths(data_path):
        data_path = util.resolve_path(data_path)
        paths = [os.path.join(data_path, n) for n in ["metadata.jsn"] + [f"{a}.npy" for a in CoreRemoraDataset._core_arrays]]
        paths += glob(os.path.join(data_path, "extra_*.npy"))
        if os.path.isfile(os.path.join(data_path, "kmer_table.npy")):
            paths.append(os.path.join(data_path, "kmer_table.npy"))
        return paths

    @staticmethod
    def check_dataset_dir(data_path):
        return all(os.path.isfile(p) for p in CoreRemoraDataset.dataset_paths(data_path))

    @staticmethod
    def hash(data_path):
        """sha256 over the per-file digests; files of 2 MiB and more are sampled at 8 evenly spaced 256 KiB
        windows (:975-1009), so that configs written by either implementation verify with the other."""
        bufsize, num_buf = 2**18, 8

        def file_digest(path):
            dig = hashlib.sha256()
            size = os.path.getsize(path)
            with open(path, "rb") as fh:
                if size < bufsize * num_buf:
                    for block in iter(lambda: fh.read(bufsize), b""):
                        dig.update(block)
                else:
                    for pos in np.floor(np.linspace(0, size - bufsize, num_buf)).astype(int):
                        fh.seek(int(pos))
                        dig.update(fh.read(bufsize))
            return dig.hexdigest()

        joined = "".join(file_digest(p) for p in CoreRemoraDataset.dataset_paths(data_path))
        return hashlib.sha256(joined.encode("utf-8")).hexdigest()

    @property
    def metadata_path(self):
        if self.data_path is None:
            raise RemoraError("No path available for in-memory dataset")
        return os.path.join(self.data_path, "metadata.jsn")

    @property
    def kmer_table_path(self):
        if self.data_path is None:
            raise RemoraError("No path available for in-memory dataset")
        return os.path.join(self.data_path, "kmer_table.npy")

    def get_array_path(self, array_name):
        if self.data_path is None:
            raise RemoraError("No path available for in-memory dataset")
        if array_name in self._core_arrays:
            return os.path.join(self.data_path, f"{array_name}.npy")
        if array_name in (self.metadata.extra_arrays or {}):
            return os.path.join(self.data_path, f"extra_{array_name}.npy")
        raise RemoraError(f"Invalid extra array name: {array_name}")

    @property
    def array_names(self):
        return self._core_arrays + self.metadata.extra_array_names

    @property
    def arrays_info(self):
        md = self.metadata
        return [(n, dt, getattr(md, f"{n}_shape")) for n, dt in self._core_dtypes.items()] + md.extra_array_dtypes_and_shapes

    def __getattr__(self, name):  # ds.signal, ds.labels, ds.read_ids ... as in the reference
        arrays = self.__dict__.get("arrays") or {}
        if name in arrays:
            return arrays[name]
        raise AttributeError(name)

    def allocate_arrays(self):
        if self.mode != "w":
            raise RemoraError("Cannot write when mode is not 'w'")
        for name, dt, shape in self.arrays_info:
            if self.data_path is None:
                self.arrays[name] = np.empty(shape, dtype=dt)
            else:
                self.arrays[name] = np.memmap(self.get_array_path(name), dt, mode="w+", shape=shape)

    def refresh_memmaps(self):
        if self.data_path is None:
            return
        self.arrays = {}
        for name, dt, shape in self.arrays_info:
            path = self.get_array_path(name)
            if os.path.getsize(path) != int(np.prod(shape)) * np.dtype(dt).itemsize:
                raise RemoraError(f"{path} does not have the size metadata.jsn implies")
            self.arrays[name] = np.memmap(path, dt, mode="r" if self.mode == "r" else "r+", shape=shape)

    def close_memmaps(self):
        if self.data_path is not None:
            self.arrays = {}

    # ---- metadata ---------------------------------------------------------------------------
    def load_metadata(self):
        """metadata.jsn with `override_metadata` applied (:1078-1216).  Overridable: dataset_start / dataset_end
        (slicing), mod_bases + mod_long_names (more labels; stored labels are converted), extra_arrays (a
        subset), kmer_context_bases / chunk_context (not larger than stored)."""
        from .refine_signal_map import SigMapRefiner

        with open(self.metadata_path) as fh:
            loaded = json.load(fh)
        if loaded.get("version") != DATASET_VERSION:
            raise RemoraError(f"Remora dataset version ({loaded.get('version')}) does not match current "
                              f"distribution ({DATASET_VERSION})")
        if os.path.exists(self.kmer_table_path):
            loaded["refine_kmer_levels"] = np.load(self.kmer_table_path)
        loaded["refine_sd_arr"] = np.asarray(loaded["refine_sd_arr"], np.float32)
        loaded["sig_map_refiner"] = SigMapRefiner.load_from_metadata(loaded)
        for key in [k for k in loaded if k.startswith("refine_")]:
            del loaded[key]
        stored_mods = [str(mb) for mb in loaded["mod_bases"]]
        invalid = []
        for key, val in (self.override_metadata or {}).items():
            if key == "dataset_start":
                if val < 0:
                    raise RemoraError("Dataset start must be positive")
            elif key == "dataset_end":
                if val > loaded["dataset_end"]:
                    raise RemoraError("Cannot set dataset end past loaded end")
            elif key == "mod_bases":
                val = [str(mb) for mb in val]
                if "mod_long_names" not in self.override_metadata or len(self.override_metadata["mod_long_names"]) != len(val):
                    raise RemoraError("mod_bases and mod_long_names must be overridden together")
                if not all(mb in val for mb in stored_mods):
                    raise RemoraError("Cannot remove modified base")
                if stored_mods != val[: len(stored_mods)]:
                    self.label_conv = np.zeros(len(stored_mods) + 1, dtype=np.int64)
                    for lab, mb in enumerate(stored_mods):
                        self.label_conv[lab + 1] = val.index(mb) + 1
            elif key == "mod_long_names":
                if "mod_bases" not in self.override_metadata:
                    raise RemoraError("mod_bases and mod_long_names must be overridden together")
            elif key == "extra_arrays":
                val = {} if val is None else val  # (the reference needs a dict here; None = no extra arrays)
                missing = set(val).difference(loaded["extra_arrays"] or {})
                if missing:
                    raise RemoraError(f"Cannot load missing arrays: {', '.join(sorted(missing))}\nAvailable extra "
                                      f"arrays: {', '.join((loaded['extra_arrays'] or {}).keys())}")
                val = {k: loaded["extra_arrays"][k] for k in val}
            elif key in ("chunk_context", "kmer_context_bases"):
                val = tuple(int(x) for x in val)
                stored = loaded[key] = tuple(loaded[key])
                if val[0] > stored[0] or val[1] > stored[1]:
                    what = "chunk context" if key == "chunk_context" else "kmer context"
                    raise RemoraError(f"Cannot expand {what} (stored:{stored} ; requested:{val})")
                loaded["_stored_" + key] = stored
            else:
                invalid.append(key)
                continue
            loaded[key] = val
        if self.override_metadata is not None:
            if loaded["dataset_start"] >= loaded["dataset_end"]:
                raise RemoraError("Loaded dataset is empty")
            if invalid:
                raise RemoraError(f"Cannot change metadata values: {', '.join(invalid)}")
        self.metadata = DatasetMetadata(**loaded)

    def update_metadata(self, other):
        """Take labels, extra arrays and contexts from another dataset's metadata (:1218-1247)."""
        md = {k: getattr(other.metadata, k) for k in ("mod_bases", "mod_long_names", "extra_arrays",
                                                     "kmer_context_bases", "chunk_context")}
        md.update(dataset_start=self.metadata.dataset_start, dataset_end=self.metadata.dataset_end)
        self.override_metadata = md
        self.load_metadata()
        self.refresh_memmaps()

    def write_metadata(self):
        self.metadata.write(self.metadata_path, self.kmer_table_path)

    # convenience views of the loaded (possibly overridden) contexts
    chunk_context = property(lambda s: s.metadata.chunk_context)
    kmer_context_bases = property(lambda s: s.metadata.kmer_context_bases)
    stored_chunk_context = property(lambda s: s.metadata.stored_chunk_context)
    stored_kmer_context_bases = property(lambda s: s.metadata.stored_kmer_context_bases)
    chunk_len = property(lambda s: s.metadata.chunk_width)
    size = property(lambda s: s.metadata.dataset_end - s.metadata.dataset_start)

    # ---- writing (:1345-1469) ---------------------------------------------------------------
    def write_batch(self, arrays):
        """Append rows given as {array name: array[n, ...]} (:1345-1374).  2-D rows narrower than the stored
        width are padded (sequence with -1, mapping with 0)."""
        if self.mode != "w":
            raise RemoraError("Cannot write when mode is not 'w'")
        n = next(iter(arrays.values())).shape[0]
        if any(a.shape[0] != n for a in arrays.values()):
            raise RemoraError("All arrays in a batch must be the same size")
        end = int(self.metadata.dataset_end)
        if end + n > int(self.metadata.allocate_size):
            self.write_metadata()
            raise RemoraError("Batch write greater than allocated memory")
        missing = set(self.array_names).difference(arrays)
        if missing:
            raise RemoraError(f"Batch write must include all arrays. Missing: {', '.join(sorted(missing))}")
        extra = set(arrays).difference(self.array_names)
        if extra:
            raise RemoraError(f"Batch write must only include spcified arrays. Found: {', '.join(sorted(extra))}")
        for name, a in arrays.items():
            out = self.arrays[name]
            a = np.asarray(a)
            if a.ndim == 2 and a.shape[1] < out.shape[1]:
                out[end : end + n, : a.shape[1]] = a
                out[end : end + n, a.shape[1] :] = -1 if name == "sequence" else 0
            else:
                out[end : end + n] = a
        self.metadata.dataset_end = end + n

    def write_chunk(self, chunk):
        """One `Chunk` (:1376-1418), with the read_ids / read_focus_bases extras when the dataset has them."""
        row = {
            "signal": np.asarray(chunk.signal, np.float32)[None, None, :],
            "sequence": np.asarray(chunk.seq_w_context, np.int8)[None, :],
            "sequence_to_signal_mapping": np.asarray(chunk.seq_to_sig_map, np.int16)[None, :],
            "sequence_lengths": np.asarray([chunk.seq_len], np.int16),
            "labels": np.asarray([chunk.label], np.int64),
        }
        extras = self.metadata.extra_arrays or {}
        if "read_ids" in extras:
            row["read_ids"] = np.array([chunk.read_id], dtype=extras["read_ids"][0])
        if "read_focus_bases" in extras:
            row["read_focus_bases"] = np.array([chunk.read_focus_base], dtype=extras["read_focus_bases"][0])
        self.write_batch(row)

    def write_chunk_arrays(self, arrs, keep=None, read_ids=None):
        """Append GPU-extracted `ChunkArrays`; chunks longer than max_seq_len are dropped, as `remora dataset
        prepare` does (src/remora/prepare_train_data.py:213-221).  `read_ids` (one per chunk) feeds the
        read_ids extra array.  Returns the number of chunks written."""
        lens = arrs.lengths.cpu().numpy()
        ok = lens <= int(self.metadata.max_seq_len)
        if keep is not None:
            ok &= np.asarray(keep, bool)
        if not ok.any():
            return 0
        md = self.metadata
        rows = {
            "signal": arrs.signal.cpu().numpy()[ok],
            "sequence": arrs.sequence.cpu().numpy()[ok][:, : md.sequence_width],
            "sequence_to_signal_mapping": arrs.mapping.cpu().numpy()[ok][:, : md.sequence_to_signal_mapping_width],
            "sequence_lengths": lens[ok],
            "labels": np.asarray(arrs.labels, np.int64)[ok],
        }
        extras = md.extra_arrays or {}
        if "read_ids" in extras:
            if read_ids is None:
                raise RemoraError("read_ids are needed for the read_ids extra array")
            rows["read_ids"] = np.asarray(read_ids, dtype=extras["read_ids"][0])[ok]
        if "read_focus_bases" in extras:
            rows["read_focus_bases"] = arrs.read_focus_bases.cpu().numpy().astype(extras["read_focus_bases"][0])[ok]
        self.write_batch(rows)
        return int(ok.sum())

    def shuffle(self, batch_size=100_000, show_prog=False):
        """One np.random permutation applied to the written rows of every array (:1420-1469)."""
        if self.mode != "w":
            raise RemoraError("Cannot write when mode is not 'w'")
        a0, a1 = int(self.metadata.dataset_start), int(self.metadata.dataset_end)
        perm = np.random.permutation(a1 - a0)
        for name in self.array_names:
            view = self.arrays[name][a0:a1]
            src = view.copy()
            for st in range(0, a1 - a0, batch_size):
                view[st : st + batch_size] = src[perm[st : st + batch_size]]
            if hasattr(self.arrays[name], "flush"):
                self.arrays[name].flush()

    def flush(self):
        if self.data_path is None:
            return
        for a in self.arrays.values():
            a.flush()
        if self.mode == "w":
            self.write_metadata()

    # ---- reading (:1470-1702) ---------------------------------------------------------------
    def get_label_counts(self):
        labs = np.asarray(self.arrays["labels"][self.metadata.dataset_start : self.metadata.dataset_end])
        if self.label_conv is not None:
            labs = self.label_conv[labs]
        return np.bincount(labs, minlength=self.metadata.num_labels)

    @property
    def label_summary(self):
        return "; ".join(f"{self.metadata.labels[i]}:{c:,}" for i, c in enumerate(self.get_label_counts()))

    @property
    def summary(self):
        md = self.metadata
        return (f"                data_path : {self.data_path}\n"
                f"                     size : {self.size:,}\n"
                f"            dataset_start : {md.dataset_start:,}\n"
                f"              dataset_end : {md.dataset_end:,}\n"
                f"       label distribution : {self.label_summary}\n"
                f"     modified_base_labels : {md.modified_base_labels}\n"
                f"                mod_bases : {md.mod_bases}\n"
                f"           mod_long_names : {md.mod_long_names}\n"
                f"       kmer_context_bases : {md.kmer_context_bases}\n"
                f"            chunk_context : {md.chunk_context}\n"
                f"                   motifs : {md.motifs}\n"
                f"           reverse_signal : {md.reverse_signal}\n"
                f" chunk_extract_base_start : {md.base_start_justify}\n"
                f"     chunk_extract_offset : {md.offset}\n"
                f"          sig_map_refiner : {md.sig_map_refiner}\n")

    def adjust_batch_params(self):
        """Clamp the super-batch size to the dataset and derive the number of chunks drawn from each super
        batch for `super_batch_sample_frac` (:1471-1510).  -> (chunks_per_super_batch, select_num_chunks|None)"""
        self.super_batch_size = min(self.super_batch_size, self.size)
        frac = self.super_batch_sample_frac
        if frac is None:
            return self.super_batch_size, None
        select = int(np.ceil(self.super_batch_size * frac / self.batch_size) * self.batch_size)
        if select > self.super_batch_size:
            select -= self.batch_size
        if select == 0:
            self.batch_size = int(self.super_batch_size * frac)
            select = self.batch_size
        if frac == 1.0:
            self.super_batch_size = select
        return select, select

    def _trim(self, b):
        """Dynamic k-mer / chunk context trimming of a super batch (:1512-1576; T1 kernel)."""
        from .data_chunks_core import trim_sb_chunk_context_core

        md = self.metadata
        seq_diff = md.stored_kmer_context_bases[0] - md.kmer_context_bases[0]
        if seq_diff > 0:  # the trailing trim happens in the encode via the smaller ka
            b["sequence"][:, :-seq_diff] = b["sequence"][:, seq_diff:].copy()
        if md.chunk_context_adjusted:
            st_diff = md.stored_chunk_context[0] - md.chunk_context[0]
            new_en = md.stored_chunk_context[0] + md.chunk_context[1]
            b["signal"] = np.ascontiguousarray(b["signal"][:, :, st_diff:new_en])
            b["sequence_to_signal_mapping"] = (b["sequence_to_signal_mapping"] - st_diff).astype(np.int16)
            trim_sb_chunk_context_core(*md.stored_chunk_context, *md.chunk_context, sum(md.kmer_context_bases),
                                       b["sequence"], b["sequence_to_signal_mapping"], b["sequence_lengths"])
        return b

    def load_super_batch(self, offset=0, size=None, select_num_chunks=None, copy=True):
        """`size` rows from `offset` (relative to dataset_start): wraps around the end for infinite iteration,
        returns a short last batch otherwise, None past the end (:1578-1633).  With copy=False arrays that need
        no rewriting (trimming, label conversion) stay read-only views of the memmaps - the fused GPU path
        uploads straight from them."""
        md = self.metadata
        if self.infinite_iter:
            offset %= self.size
        elif offset >= self.size:
            return None
        st = md.dataset_start + offset
        if size is None:
            if self.infinite_iter:
                raise RemoraError("Must specify size of super batch for infinite iter dataset")
            size = md.dataset_end - st
        if size > self.size:
            raise RemoraError("Super batch larger than dataset requested")
        en = st + size
        rewritten = set()
        if md.kmer_context_bases_adjusted and md.stored_kmer_context_bases[0] > md.kmer_context_bases[0]:
            rewritten.add("sequence")
        if md.chunk_context_adjusted:
            rewritten.update(("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths"))
        take = (lambda n, a: np.array(a)) if copy else (lambda n, a: np.array(a) if n in rewritten else a)
        if en <= md.dataset_end:
            sb = {n: take(n, self.arrays[n][st:en]) for n in self.array_names}
        elif self.infinite_iter:
            wrap = en - self.size
            sb = {n: np.concatenate([self.arrays[n][st : md.dataset_end], self.arrays[n][md.dataset_start : wrap]])
                  for n in self.array_names}
        else:
            sb = {n: take(n, self.arrays[n][st : md.dataset_end]) for n in self.array_names}
        if select_num_chunks is not None:
            pick = np.random.choice(sb["labels"].size, min(select_num_chunks, sb["labels"].size), replace=False)
            sb = {n: a[pick] for n, a in sb.items()}
        if self.label_conv is not None:
            sb["labels"] = self.label_conv[sb["labels"]]
        return self._trim(sb)

    def load_batch(self, st, en):
        """Rows [st, en) of the dataset (relative to dataset_start), trimmed to the loaded contexts."""
        keep, self.infinite_iter = self.infinite_iter, False
        try:
            return self.load_super_batch(st, min(en, self.size) - st)
        finally:
            self.infinite_iter = keep

    def iter_super_batches(self, select_num_chunks=None, copy=True):
        num = 0
        while True:
            self.refresh_memmaps()
            sb = self.load_super_batch(self.super_batch_offset + num * self.super_batch_size, self.super_batch_size,
                                       select_num_chunks=select_num_chunks, copy=copy)
            if sb is None:
                return
            if self.do_check_super_batches:
                check_super_batch(sb, self.metadata.chunk_width)
            num += 1
            yield sb

    def extract_batch(self, super_batch, batch_st, enc_kmers=False):
        """One batch of a super batch (:1652-1676): the core rows, plus `enc_kmers` from the encode kernel when
        asked for (the fused path does not need it)."""
        en = min(batch_st + self.batch_size, super_batch["sequence"].shape[0])
        batch = {n: a[batch_st:en] for n, a in super_batch.items()}
        if enc_kmers and en <= batch_st:
            md = self.metadata
            batch["enc_kmers"] = np.zeros((0, 4 * md.kmer_len, md.chunk_width), np.float32)
        elif enc_kmers:
            from .encoded_kmers import compute_encoded_kmer_batch

            batch["enc_kmers"] = compute_encoded_kmer_batch(
                *self.metadata.kmer_context_bases, batch["sequence"], batch["sequence_to_signal_mapping"],
                batch["sequence_lengths"])
        return batch

    def iter_batches(self, max_batches=None, enc_kmers=False, copy=True):
        chunks_per_sb, select = self.adjust_batch_params()
        num = 0
        for sb in self.iter_super_batches(select, copy=copy):
            for st in range(0, chunks_per_sb, self.batch_size):
                # (a short last super batch of a finite dataset yields short and then EMPTY batches, as in the
                # reference; consumers that cannot take an empty batch skip it)
                yield self.extract_batch(sb, st, enc_kmers)
                num += 1
                if max_batches is not None and num >= max_batches:
                    return

    def __iter__(self):
        if self._iter is None or not self.infinite_iter:
            self._iter = self.iter_batches()
        return self._iter

    def __next__(self):
        return next(self._iter)


def parse_dataset_config(config_path, used_configs=None):
    """Dataset config = JSON list of [path, weight] or [path, weight, hash]; a path may itself be a config
    (weights multiply).  -> (core paths, proportions summing to 1, hashes) (:1705-1760)."""
    config_path = util.resolve_path(config_path)
    used_configs = {config_path: config_path} if used_configs is None else used_configs
    paths, weights, hashes = [], [], []
    with open(config_path) as fh:
        entries = json.load(fh)
    for entry in entries:
        ds_path, weight = entry[0], entry[1]
        ds_hash = entry[2] if len(entry) == 3 else None
        if not weight > 0:
            raise RemoraError("dataset config weight must be positive")
        ds_path = util.resolve_path(ds_path)
        if not os.path.exists(ds_path):
            raise RemoraError(f"Core dataset path does not exist. {ds_path}")
        if os.path.isdir(ds_path):
            computed = CoreRemoraDataset.hash(ds_path)
            if ds_hash is not None and ds_hash != computed:
                raise RemoraError(f"Dataset hash does not match value from config for dataset at {ds_path}")
            paths.append(ds_path)
            weights.append(weight)
            hashes.append(computed)
            continue
        if ds_path in used_configs:
            raise RemoraError(f"Circular or repeated dataset config refrence. {ds_path} found in {config_path} and "
                              f"previously found in {used_configs[ds_path]}")
        used_configs[ds_path] = config_path
        sub_paths, sub_props, sub_hashes = parse_dataset_config(ds_path, used_configs=used_configs)
        paths.extend(sub_paths)
        weights.extend(sub_props * weight)
        hashes.extend(sub_hashes)
    weights = np.array(weights, dtype=float)
    return paths, weights / weights.sum(), hashes


def load_dataset(ds_path):
    """A core dataset directory or a dataset config (:1763-1770)."""
    ds_path = util.resolve_path(ds_path)
    if not os.path.exists(ds_path):
        raise RemoraError(f"Dataset path does not exist. {ds_path}")
    if os.path.isdir(ds_path):
        return [ds_path], np.ones(1, dtype=float), None
    return parse_dataset_config(ds_path)


def compute_best_split(total_size, props):
    """Integer batch shares: len(props) positive counts adding up to `total_size`, as near to the proportions as
    integers allow.  Same outcome as the reference's function of this name (src/remora/data_chunks.py:1767-1786;
    pinned on reference-generated cases in tests/test_host_cpu.py): floor quotas of at least one row each, an
    over-commitment taken back from the largest share, the remainder handed out one row at a time to the share
    that lags its proportion most (first index on ties in both loops)."""
    target = [float(p) for p in props]
    k = len(target)
    if total_size < k:
        raise RemoraError(f"total_size ({total_size}) smaller than number of proportions {k}")
    quota = [max(1, int(np.floor(total_size * p))) for p in target]
    used = sum(quota)
    while used > total_size:
        j = max(range(k), key=lambda i: (quota[i], -i))
        quota[j] -= 1
        used -= 1
    while used < total_size:
        j = min(range(k), key=lambda i: (quota[i] / used - target[i], i))
        quota[j] += 1
        used += 1
    return np.asarray(quota, dtype=int)


class RemoraDataset:
    """Several core datasets drawn from at fixed proportions in every batch (src/remora/data_chunks.py:1806-2276):
    labels, motifs and contexts are reconciled across the datasets, each contributes `batch_sizes[i]` rows per
    batch.  Batches are lists of arrays named by `return_arrays`; besides the reference's "enc_kmers", "signal",
    "labels" the raw rows ("sequence", "sequence_to_signal_mapping", "sequence_lengths", padded to a common
    width) can be requested — those feed the fused GPU path."""

    def __init__(self, datasets, proportions, hashes=None, batch_size=constants.DEFAULT_BATCH_SIZE,
                 super_batch_size=constants.DEFAULT_SUPER_BATCH_SIZE, super_batch_sample_frac=None, seed=None,
                 return_arrays=("enc_kmers", "signal", "labels")):
        self.datasets, self.props = list(datasets), np.asarray(proportions, dtype=float)
        if not all(0 <= p <= 1 for p in self.props):
            raise RemoraError("Dataset proportions must be between 0 and 1.")
        if len(self.datasets) != len(self.props):
            raise RemoraError("Dataset and proportions must be same length.")
        self._hashes = hashes
        self.set_batch_size(batch_size)
        self.super_batch_size, self.super_batch_sample_frac, self.seed = super_batch_size, super_batch_sample_frac, seed
        self.return_arrays = tuple(return_arrays)
        self.infinite_iter = all(ds.infinite_iter for ds in self.datasets)
        self.set_global_metadata()
        for ds in self.datasets:
            ds.update_metadata(self)
        self.super_batch_offsets = [0] * len(self.datasets)
        self._ds_iters = self._iter = self._all_batches = None

    num_datasets = property(lambda s: len(s.datasets))
    paths = property(lambda s: [ds.data_path for ds in s.datasets])
    size = property(lambda s: sum(ds.size for ds in s.datasets))

    @property
    def hashes(self):
        if self._hashes is None or any(h is None for h in self._hashes):
            self._hashes = [ds.hash(ds.data_path) for ds in self.datasets]
        return self._hashes

    @property
    def init_kwargs(self):
        return dict(proportions=self.props, hashes=self._hashes, batch_size=self.batch_size,
                    super_batch_size=self.super_batch_size, super_batch_sample_frac=self.super_batch_sample_frac,
                    seed=self.seed, return_arrays=self.return_arrays)

    @property
    def summary(self):
        md = self.metadata
        return (f"                     size : {self.size:,}\n"
                f"     modified_base_labels : {md.modified_base_labels}\n"
                f"                mod_bases : {md.mod_bases}\n"
                f"           mod_long_names : {md.mod_long_names}\n"
                f"       kmer_context_bases : {md.kmer_context_bases}\n"
                f"            chunk_context : {md.chunk_context}\n"
                f"                   motifs : {md.motifs}\n"
                f"           reverse_signal : {md.reverse_signal}\n"
                f" chunk_extract_base_start : {md.base_start_justify}\n"
                f"     chunk_extract_offset : {md.offset}\n"
                f"               pa_scaling : {md.pa_scaling}\n"
                f"          sig_map_refiner : {md.sig_map_refiner}\n")

    def set_global_metadata(self):
        """Metadata of the union (:1862-1992): exact-match attributes checked, mod bases unioned and sorted by
        short name, k-mer context reduced to the common minimum, motif sets merged."""
        md = self.metadata = self.datasets[0].metadata.copy()
        for name in ("allocate_size", "max_seq_len", "dataset_start", "dataset_end"):
            setattr(md, name, None)

        def set_motifs(motifs):
            md.motif_sequences, md.motif_offsets = zip(*[m.to_tuple() for m in util.merge_motifs(motifs)])
            md.check_motifs()

        set_motifs(md.motifs)
        for ds in self.datasets[1:]:
            other = ds.metadata
            for attr in ("modified_base_labels", "base_start_justify", "offset", "reverse_signal", "pa_scaling",
                         "sig_map_refiner"):
                if getattr(other, attr) != getattr(md, attr):
                    raise RemoraError(f"All datasets must have same {attr} {getattr(other, attr)} != {getattr(md, attr)}")
            if set(other.extra_array_names) != set(md.extra_array_names):
                raise RemoraError(f"Extra arrays not equal: {other.extra_array_names} != {md.extra_array_names}")
            for mb, mln in zip(other.mod_bases, other.mod_long_names):
                if mb in md.mod_bases:
                    known = md.mod_long_names[md.mod_bases.index(mb)]
                    if mln != known:
                        raise RemoraError(f"Mismatched modified bases.\n\tPreviously loaded modified bases: "
                                          f"{md.mod_bases} {md.mod_long_names}\n\tNew modified bases: "
                                          f"{other.mod_bases} {other.mod_long_names}")
                else:
                    md.mod_bases.append(mb)
                    md.mod_long_names.append(mln)
            if other.kmer_context_bases != md.kmer_context_bases:
                md.kmer_context_bases = tuple(min(a, b) for a, b in zip(md.kmer_context_bases, other.kmer_context_bases))
            if other.chunk_context != md.chunk_context:
                # (sic) the reference stores the reduced chunk context into kmer_context_bases (:1949-1964);
                # kept, because the datasets are then re-loaded with exactly these values
                md.kmer_context_bases = tuple(min(a, b) for a, b in zip(md.chunk_context, other.chunk_context))
            if set(other.motifs) != set(md.motifs):
                set_motifs(md.motifs + other.motifs)
        order = sorted(range(len(md.mod_bases)), key=md.mod_bases.__getitem__)
        md.mod_bases = [md.mod_bases[i] for i in order]
        md.mod_long_names = [md.mod_long_names[i] for i in order]

    def update_metadata(self, other):
        for key in ("modified_base_labels", "offset", "reverse_signal", "pa_scaling", "sig_map_refiner"):
            if getattr(self.metadata, key) != getattr(other.metadata, key):
                raise RemoraError(f"Cannot update metadata with mismatching '{key}'. ({getattr(self.metadata, key)} != "
                                  f"{getattr(other.metadata, key)})")
        for ds in self.datasets:
            ds.update_metadata(other)
        for key in ("mod_bases", "mod_long_names", "extra_arrays", "kmer_context_bases", "chunk_context"):
            setattr(self.metadata, key, getattr(other.metadata, key))

    def set_batch_size(self, batch_size):
        self.batch_size = int(batch_size)
        self.batch_sizes = compute_best_split(self.batch_size, self.props)

    @classmethod
    def from_config(cls, config_path, override_metadata=None, ds_kwargs=None, **kwargs):
        paths, props, hashes = parse_dataset_config(config_path)
        datasets = [CoreRemoraDataset(p, override_metadata=dict(override_metadata or {}), **(ds_kwargs or {})) for p in paths]
        return cls(datasets, props, hashes, **kwargs)

    def _split(self, sizes, make_md, **ds_kwargs):
        out = []
        for ds, n in zip(self.datasets, sizes):
            if n >= ds.size:
                raise RemoraError("Not enough chunks")
            out.append(CoreRemoraDataset(ds.data_path, override_metadata=make_md(ds, int(n)), **ds_kwargs))
        return out

    def train_test_split(self, num_test_chunks, override_metadata=None):
        """The first rows of every dataset (in proportion) become the finite test set, the rest the training set (:2078-2108)."""
        sizes = compute_best_split(num_test_chunks, self.props)
        base = dict(override_metadata or {})
        train = self._split(sizes, lambda ds, n: {**base, "dataset_start": ds.metadata.dataset_start + n})
        test = self._split(sizes, lambda ds, n: {**base, "dataset_end": ds.metadata.dataset_start + n}, infinite_iter=False)
        return RemoraDataset(train, **self.init_kwargs), RemoraDataset(test, **self.init_kwargs)

    def head(self, num_chunks, override_metadata=None):
        sizes = compute_best_split(num_chunks, self.props)
        base = dict(override_metadata or {})
        heads = self._split(sizes, lambda ds, n: {**base, "dataset_start": ds.metadata.dataset_start,
                                                  "dataset_end": ds.metadata.dataset_start + n}, infinite_iter=False)
        return RemoraDataset(heads, **self.init_kwargs)

    def shard(self, rank, world, override_metadata=None):
        """Rank `rank`'s share when `world` processes (one per GPU) split the dataset: every core dataset is narrowed to a
        contiguous range of its rows (dist.shard_range), proportions and batch scheme unchanged, finite iteration.  The
        shares partition the rows, so per-label tallies summed over the ranks equal the single-process ones whenever a
        single process visits every row (always for one core dataset; mixes stop, as in the reference, when the first
        core dataset runs out — per rank here)."""
        from .dist import shard_range

        base = dict(override_metadata or {})
        parts = []
        for ds in self.datasets:
            a, b = shard_range(ds.size, rank, world)
            if b <= a:
                raise RemoraError(f"dataset {ds.data_path} has fewer rows ({ds.size}) than ranks ({world})")
            st = ds.metadata.dataset_start
            parts.append(CoreRemoraDataset(ds.data_path, override_metadata={**(ds.override_metadata or {}), **base,
                                                                            "dataset_start": st + a, "dataset_end": st + b},
                                           infinite_iter=False))
        return RemoraDataset(parts, **self.init_kwargs)

    def _set_sub_ds_iters(self, enc_kmers, copy=True):
        for ds, bs, off in zip(self.datasets, self.batch_sizes, self.super_batch_offsets):
            ds.batch_size, ds.super_batch_offset = int(bs), int(off)
            ds.super_batch_size, ds.super_batch_sample_frac = self.super_batch_size, self.super_batch_sample_frac
        self._ds_iters = [ds.iter_batches(enc_kmers=enc_kmers, copy=copy) for ds in self.datasets]

    @staticmethod
    def _concat(name, parts):
        if len(parts) == 1:
            return parts[0]
        if name in ("sequence", "sequence_to_signal_mapping"):  # datasets may differ in max_seq_len
            width = max(p.shape[1] for p in parts)
            fill = -1 if name == "sequence" else 0
            parts = [p if p.shape[1] == width else np.pad(p, ((0, 0), (0, width - p.shape[1])), constant_values=fill)
                     for p in parts]
        return np.concatenate(parts)

    def iter_numpy_batches(self, return_arrays=None, copy=False):
        """Batches as lists of numpy arrays until any core dataset runs out.  With copy=False (default here) a
        single-dataset batch whose rows need no rewriting is a read-only view of the memmaps: nothing is copied
        on the host between the page cache and the upload."""
        names = tuple(return_arrays) if return_arrays is not None else self.return_arrays
        if self._ds_iters is None:
            self._set_sub_ds_iters("enc_kmers" in names, copy=copy)
        while True:
            try:
                parts = [next(it) for it in self._ds_iters]
            except StopIteration:
                return
            yield [self._concat(n, [p[n] for p in parts]) for n in names]

    def iter_batches(self, return_arrays=None):
        """Batches as lists of torch tensors, the reference's element type (:2135-2149)."""
        torch = _torch()
        for arrays in self.iter_numpy_batches(return_arrays, copy=True):
            yield [torch.from_numpy(np.ascontiguousarray(a)) for a in arrays]

    def load_all_batches(self):
        if self.infinite_iter:
            raise RemoraError("Cannot save all batches for infinite dataset")
        self._ds_iters = None
        self._all_batches = list(self.iter_batches())
        for ds in self.datasets:
            ds.close_memmaps()

    def __iter__(self):
        if self._all_batches is not None:
            self._iter = iter(self._all_batches)
            return self._iter
        if self._iter is None or not self.infinite_iter:
            self._ds_iters = None
            self._iter = self.iter_batches()
        return self._iter

    def __next__(self):
        return next(self._iter)

    def get_label_counts(self):
        counts = np.zeros(self.metadata.num_labels, dtype=int)
        if self._all_batches is not None and "labels" in self.return_arrays:
            li = self.return_arrays.index("labels")
            for b in self._all_batches:
                c = np.bincount(b[li].numpy())
                counts[: c.size] += c
            return counts
        for ds in self.datasets:
            c = ds.get_label_counts()
            counts[: c.size] += c
        return counts

    @property
    def label_summary(self):
        return "; ".join(f"{self.metadata.labels[i]}:{c:,}" for i, c in enumerate(self.get_label_counts()))

    def get_config(self):
        return [(p, float(w)) if h is None else (p, float(w), h) for p, w, h in zip(self.paths, self.props, self.hashes)]

    def epoch_summary(self, batches_per_epoch):
        """Per-dataset table of how much of each dataset an epoch consumes (:2205-2276)."""
        labs = self.metadata.labels
        lines = []
        for ds, bs in zip(self.datasets, self.batch_sizes):
            counts = dict(zip(ds.metadata.labels, ds.get_label_counts()))
            tot = sum(counts.values())
            per_batch = "\t".join(f"{int(np.ceil(counts.get(lab, 0) / tot * bs)):,}" for lab in labs)
            stored = "\t".join(f"{counts.get(lab, 0):,}" for lab in labs)
            per_epoch = batches_per_epoch * int(bs)
            lines.append(f"{per_epoch / ds.size:10.4%}\t{per_batch}\t{per_epoch:,}\t{ds.size:,}\t{stored}\t{ds.data_path}")
        header = ("percent_of_dataset_per_epoch\t" + "\t".join(f"batch_{lab}" for lab in labs) +
                  "\tdataset_chunks_per_epoch\tdataset_size\t" + "\t".join(f"dataset_{lab}" for lab in labs) + "\tpath\n")
        return header + "\n".join(lines)


def validate_dataset(dataset, model):
    """Per-chunk calls for an on-disk dataset: logits through the fused path, argmax tally on the GPU
    (label counts) and the confusion matrix against the stored labels — the numbers
    `remora validate from_remora_dataset` derives (src/remora/validate.py:42-66, 190-259)."""
    num_out = model.num_out
    counts = np.zeros(num_out, np.int64)
    conf = np.zeros((num_out, num_out), np.int64)
    logits = []
    if dataset.infinite_iter:
        raise RemoraError("validate_dataset needs a finite dataset (infinite_iter=False)")
    for b in dataset.iter_batches():
        out = model.infer_chunks(b["signal"], b["sequence"], b["sequence_to_signal_mapping"], b["sequence_lengths"],
                                 dataset.kmer_context_bases, label_counts=counts)
        pred = out.argmax(axis=1)
        ok = (b["labels"] >= 0) & (b["labels"] < num_out)
        np.add.at(conf, (b["labels"][ok], pred[ok]), 1)
        logits.append(out)
    logits = np.concatenate(logits) if logits else np.zeros((0, num_out), np.float32)
    total = int(conf.sum())
    return dict(logits=logits, pred_counts=counts, confusion=conf, acc=(np.trace(conf) / total) if total else float("nan"))
