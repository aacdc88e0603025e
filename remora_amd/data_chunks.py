"""Read model and chunk extraction of the hot path — the inference half of
`remora.data_chunks` (src/remora/data_chunks.py): RemoraRead :126-540, Chunk :544-641.

The per-chunk Python of the reference (extract_chunk :331-423 called once per focus base,
write_chunk :1376-1418) is replaced by two kernel launches for the whole read
(rmr_chunk_geometry + rmr_chunk_fill); the arrays stay on the GPU in the CoreRemoraDataset
layout (:786-816) and feed `HipModel.infer_chunks` without ever materialising the one-hot.
"""
import ctypes
import dataclasses

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
    match runs, linearly interpolated across insertions / deletions (src/remora/data_chunks.py:77-115)."""
    cigar = list(cigar)
    while cigar and not MATCH_OPS[cigar[-1][0]]:
        cigar.pop()
    if not cigar:
        raise RemoraError("No match operations found in alignment cigar")
    ops, lens = (np.array(x) for x in zip(*cigar))
    if ops.min() < 0 or ops.max() > 8:
        raise RemoraError("Invalid cigar op(s)")
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


_PINNED = {}


def _pinned(key, dt, count):
    """Grow-only pinned host staging buffers, one per array kind (pinned allocation is too slow to do per batch)."""
    torch = _torch()
    buf = _PINNED.get(key)
    if buf is None or buf.numel() < count:
        buf = torch.empty(max(int(count * 1.25), 1 << 16), dtype=getattr(torch, np.dtype(dt).name), pin_memory=True)
        _PINNED[key] = buf
    return buf


class DeviceReads:
    """The arrays of a batch of reads, concatenated and resident in HBM (the rmr_reads layout of
    include/remora_hip.h) - uploaded once and shared by the motif scan, the signal-mapping refinement
    and the chunk extraction."""

    def __init__(self, reads, engine=None):
        torch = _torch()
        self.engine = engine if engine is not None else get_engine()
        dev = self.engine.torch_device
        nr = len(reads)
        self.n_reads = nr
        self.sig_off = np.zeros(nr + 1, np.int64)
        self.seq_off = np.zeros(nr + 1, np.int64)
        for i, r in enumerate(reads):
            self.sig_off[i + 1] = self.sig_off[i] + r.dacs.size
            self.seq_off[i + 1] = self.seq_off[i] + r.int_seq.size
            if r.seq_to_sig_map.size != r.int_seq.size + 1:
                raise RemoraError(f"Invalid read: seq ({r.int_seq.size}) and mapping ({r.seq_to_sig_map.size}) sizes incompatible")
        to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

        def cat_to_dev(arrs, dt, total, key):
            """Concatenate straight into a cached pinned staging buffer (no intermediate array, no bounce copy in the
            driver) and upload from there."""
            if total == 0:
                return torch.zeros(0, dtype=getattr(torch, np.dtype(dt).name), device=dev)
            buf = _pinned(key, dt, total)
            host = buf.numpy()[:total]
            o = 0
            for a in arrs:
                a = np.asarray(a).ravel()
                host[o : o + a.size] = a  # casts to `dt` on the fly
                o += a.size
            out = buf[:total].to(dev, non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()  # the staging buffer is reused by the next batch
            return out

        self.dacs = cat_to_dev([r.dacs for r in reads], np.int16, int(self.sig_off[-1]), "dacs")
        self.s2s = cat_to_dev([r.seq_to_sig_map for r in reads], np.int64, int(self.seq_off[-1]) + nr, "s2s")
        self.iseq = cat_to_dev([r.int_seq for r in reads], np.int8, int(self.seq_off[-1]), "iseq")
        self.d_sig_off, self.d_seq_off = to_dev(self.sig_off), to_dev(self.seq_off)
        self.set_scaling([float(r.shift) for r in reads], [float(r.scale) for r in reads])

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
        flags = torch.zeros(max(total, 1), dtype=torch.uint8, device=self.engine.torch_device)
        if total:
            L.check(L.lib().rmr_motif_flags(self.engine.handle, self.iseq.data_ptr(), self.d_seq_off.data_ptr(),
                                            self.n_reads, ctypes.byref(ms), flags.data_ptr(), L.MEM_DEVICE))
        pos = torch.nonzero(flags[:total]).flatten()  # ascending
        read_of = torch.searchsorted(self.d_seq_off, pos, right=True) - 1
        counts = torch.bincount(read_of, minlength=self.n_reads)[: self.n_reads]
        foc_off = np.zeros(self.n_reads + 1, np.int64)
        np.cumsum(counts.cpu().numpy(), out=foc_off[1:])
        return pos - self.d_seq_off[read_of], foc_off


def _extract_device(dr, focus, foc_off, chunk_context, kmer_context_bases, base_start_justify, offset, labels=None):
    """Chunk extraction for device-resident reads and focus bases (see extract_chunk_arrays)."""
    torch = _torch()
    eng, lib = dr.engine, L.lib()
    dev = eng.torch_device
    n_chunks = int(foc_off[-1])
    d_foc_off = torch.from_numpy(np.ascontiguousarray(foc_off)).to(dev)
    if focus.numel() == 0:
        focus = torch.zeros(1, dtype=torch.int64, device=dev)
    rs = L.Reads(dr.n_reads, dr.dacs.data_ptr(), dr.d_sig_off.data_ptr(), dr.s2s.data_ptr(), dr.iseq.data_ptr(),
                 dr.d_seq_off.data_ptr(), dr.shift.data_ptr(), dr.scale.data_ptr(), focus.data_ptr(),
                 d_foc_off.data_ptr(), int(chunk_context[0]), int(chunk_context[1]),
                 int(kmer_context_bases[0]), int(kmer_context_bases[1]), int(bool(base_start_justify)), int(offset))
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
    # the kernels above read the staged inputs: keep them alive until the stream drains
    eng.synchronize()
    if labels is None:
        labels = np.full(n_chunks, -1, np.int64)
    return ChunkArrays(signal, sequence, mapping, lengths, rfb, labels, geo, kmer_context_bases, chunk_context), sig


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
    d_focus = torch.from_numpy(np.ascontiguousarray(focus)).to(dr.engine.torch_device)
    return _extract_device(dr, d_focus, foc_off, chunk_context, kmer_context_bases, base_start_justify, offset, labels)


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
            keep = [fb for fb in fbs if any(m.match(self.int_seq, fb) for m in motifs)]
            saved, self.focus_bases = fbs, np.asarray(keep, dtype=np.int64)
            try:
                arrs, _ = extract_chunk_arrays([self], chunk_context, kmer_context_bases, base_start_justify, offset)
            finally:
                self.focus_bases = saved
            return arrs
        arrs, _ = extract_chunk_arrays([self], chunk_context, kmer_context_bases, base_start_justify, offset)
        return arrs

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
            poss.append(arrs.read_focus_bases.cpu().numpy())
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
import json
import os

DATASET_VERSION = 3


def dataset_metadata(allocate_size, max_seq_len, mod_bases, mod_long_names, motif_sequences, motif_offsets,
                     chunk_context=(50, 50), kmer_context_bases=(4, 4), base_start_justify=False, offset=0,
                     reverse_signal=False, pa_scaling=None, extra_arrays=None, sig_map_refiner=None,
                     modified_base_labels=True, rough_rescale_method="least_squares"):
    """The JSON-able dict the reference's DatasetMetadata.write stores in metadata.jsn
    (src/remora/data_chunks.py:786-888): same keys, same order, refiner fields flattened in."""
    from .refine_signal_map import SigMapRefiner

    ref = sig_map_refiner if sig_map_refiner is not None else SigMapRefiner()
    md = {
        "allocate_size": int(allocate_size), "max_seq_len": int(max_seq_len), "mod_bases": list(mod_bases),
        "mod_long_names": list(mod_long_names), "motif_sequences": list(motif_sequences),
        "motif_offsets": [int(x) for x in motif_offsets], "dataset_start": 0, "dataset_end": 0,
        "version": DATASET_VERSION, "modified_base_labels": bool(modified_base_labels), "extra_arrays": extra_arrays,
        "chunk_context": [int(x) for x in chunk_context], "base_start_justify": bool(base_start_justify),
        "offset": int(offset), "kmer_context_bases": [int(x) for x in kmer_context_bases],
        "reverse_signal": bool(reverse_signal), "pa_scaling": pa_scaling, "rough_rescale_method": rough_rescale_method,
        "_stored_kmer_context_bases": None, "_stored_chunk_context": None,
    }
    md.update(ref.asdict())
    md.pop("rough_rescale_method")
    md["rough_rescale_method"] = ref.rough_rescale_method if ref.is_loaded else rough_rescale_method
    # keep the reference's key order: ..., pa_scaling, rough_rescale_method, _stored_*, refine_*
    order = ["allocate_size", "max_seq_len", "mod_bases", "mod_long_names", "motif_sequences", "motif_offsets",
             "dataset_start", "dataset_end", "version", "modified_base_labels", "extra_arrays", "chunk_context",
             "base_start_justify", "offset", "kmer_context_bases", "reverse_signal", "pa_scaling",
             "rough_rescale_method", "_stored_kmer_context_bases", "_stored_chunk_context", "refine_kmer_levels",
             "refine_kmer_center_idx", "refine_do_rough_rescale", "refine_scale_iters", "refine_algo",
             "refine_half_bandwidth", "refine_sd_arr"]
    return {k: md[k] for k in order}


class CoreRemoraDataset:
    """On-disk chunk dataset in the reference's format (src/remora/data_chunks.py:926-1702): five raw
    memmapped core arrays named `<array>.npy` (no npy header) + `metadata.jsn` (+ `kmer_table.npy`).
    mode "r" reads a directory (optionally with smaller chunk / k-mer contexts than stored);
    mode "w" creates one from `metadata` (see `dataset_metadata`) and appends chunk arrays that the
    extraction kernels already produce in this layout."""

    _core_dtypes = {"signal": np.float32, "sequence": np.int8, "sequence_to_signal_mapping": np.int16,
                    "sequence_lengths": np.int16, "labels": np.int64}

    def __init__(self, data_path, override_metadata=None, batch_size=2048, mode="r", metadata=None):
        self.data_path = data_path
        self.batch_size = int(batch_size)
        self.mode = mode
        if mode not in ("r", "w"):
            raise RemoraError("mode must be 'r' or 'w'")
        if mode == "w":
            if not isinstance(metadata, dict) or "allocate_size" not in metadata:
                raise RemoraError("Must provide metadata for new dataset")
            if override_metadata:
                raise RemoraError("Cannot override metadata of a dataset opened for writing")
            os.makedirs(data_path, exist_ok=True)
            md = dict(metadata)
        else:
            with open(os.path.join(data_path, "metadata.jsn")) as fh:
                md = json.load(fh)
        if md.get("version") != DATASET_VERSION:
            raise RemoraError(f"Remora dataset version ({md.get('version')}) does not match current "
                              f"distribution ({DATASET_VERSION})")
        self.metadata = md
        self.stored_chunk_context = tuple(md.get("_stored_chunk_context") or md["chunk_context"])
        self.stored_kmer_context_bases = tuple(md.get("_stored_kmer_context_bases") or md["kmer_context_bases"])
        self.chunk_context = tuple(md["chunk_context"])
        self.kmer_context_bases = tuple(md["kmer_context_bases"])
        for k, v in (override_metadata or {}).items():
            if k == "chunk_context":
                v = tuple(int(x) for x in v)
                if v[0] > self.stored_chunk_context[0] or v[1] > self.stored_chunk_context[1]:
                    raise RemoraError("Cannot expand chunk context beyond stored chunk context")
                self.chunk_context = v
            elif k == "kmer_context_bases":
                v = tuple(int(x) for x in v)
                if v[0] > self.stored_kmer_context_bases[0] or v[1] > self.stored_kmer_context_bases[1]:
                    raise RemoraError("Cannot expand kmer context beyond stored kmer context")
                self.kmer_context_bases = v
            elif k in ("dataset_start", "dataset_end"):
                md[k] = int(v)
            else:
                raise RemoraError(f"cannot override dataset metadata attribute {k!r}")
        n, msl = int(md["allocate_size"]), int(md["max_seq_len"])
        L = sum(self.stored_chunk_context)
        shapes = {"signal": (n, 1, L), "sequence": (n, msl + sum(self.stored_kmer_context_bases)),
                  "sequence_to_signal_mapping": (n, msl + 1), "sequence_lengths": (n,), "labels": (n,)}
        self.arrays = {}
        for name, dt in self._core_dtypes.items():
            path = os.path.join(data_path, f"{name}.npy")
            if mode == "w":
                self.arrays[name] = np.memmap(path, dt, mode="w+", shape=shapes[name])
                continue
            if os.path.getsize(path) != int(np.prod(shapes[name])) * np.dtype(dt).itemsize:
                raise RemoraError(f"{path} does not have the size metadata.jsn implies")
            self.arrays[name] = np.memmap(path, dt, mode="r", shape=shapes[name])
        if mode == "w":
            self.write_metadata()

    # ---- writing (:1268-1469) ----------------------------------------------------------------
    def write_metadata(self):
        """metadata.jsn (+ kmer_table.npy when a level table is attached), as DatasetMetadata.write (:865-888)."""
        md = dict(self.metadata)
        levels = md.get("refine_kmer_levels")
        if levels is not None:
            np.save(os.path.join(self.data_path, "kmer_table.npy"), np.asarray(levels), allow_pickle=False)
            del md["refine_kmer_levels"]

        def enc(o):
            if isinstance(o, np.integer):
                return int(o)
            if isinstance(o, np.floating):
                return float(o)
            if isinstance(o, np.bool_):
                return bool(o)
            if isinstance(o, np.ndarray):
                return o.tolist()
            raise TypeError(type(o))

        with open(os.path.join(self.data_path, "metadata.jsn"), "w") as fh:
            json.dump(md, fh, default=enc)

    def write_batch(self, arrays):
        """Append rows given as {array name: array[n, ...]} in the core dtypes (:1345-1374)."""
        if self.mode != "w":
            raise RemoraError("Cannot write when mode is not 'w'")
        n = next(iter(arrays.values())).shape[0]
        if any(a.shape[0] != n for a in arrays.values()):
            raise RemoraError("All arrays in a batch must be the same size")
        end = int(self.metadata["dataset_end"])
        if end + n > int(self.metadata["allocate_size"]):
            self.write_metadata()
            raise RemoraError("Batch write greater than allocated memory")
        missing = set(self.arrays).difference(arrays)
        if missing:
            raise RemoraError(f"Batch write must include all arrays. Missing: {', '.join(sorted(missing))}")
        extra = set(arrays).difference(self.arrays)
        if extra:
            raise RemoraError(f"Batch write must only include spcified arrays. Found: {', '.join(sorted(extra))}")
        for name, a in arrays.items():
            out = self.arrays[name]
            a = np.asarray(a)
            if a.ndim == 2 and a.shape[1] < out.shape[1]:  # narrower rows: the tail columns are padding
                out[end : end + n, : a.shape[1]] = a
                out[end : end + n, a.shape[1] :] = -1 if name == "sequence" else 0
            else:
                out[end : end + n] = a
        self.metadata["dataset_end"] = end + n

    def write_chunk(self, chunk):
        """One `Chunk` (:1376-1418)."""
        self.write_batch({
            "signal": np.asarray(chunk.signal, np.float32)[None, None, :],
            "sequence": np.asarray(chunk.seq_w_context, np.int8)[None, :],
            "sequence_to_signal_mapping": np.asarray(chunk.seq_to_sig_map, np.int16)[None, :],
            "sequence_lengths": np.asarray([chunk.seq_len], np.int16),
            "labels": np.asarray([chunk.label], np.int64),
        })

    def write_chunk_arrays(self, arrs, keep=None):
        """Append GPU-extracted `ChunkArrays` (extract_chunk_arrays); chunks longer than max_seq_len are
        dropped, as `remora dataset prepare` does (src/remora/prepare_train_data.py:213-221).  Returns
        the number of chunks written."""
        lens = arrs.lengths.cpu().numpy()
        ok = lens <= int(self.metadata["max_seq_len"])
        if keep is not None:
            ok &= np.asarray(keep, bool)
        if not ok.any():
            return 0
        msl = int(self.metadata["max_seq_len"])
        sw, mw = msl + sum(self.stored_kmer_context_bases), msl + 1
        seq = arrs.sequence.cpu().numpy()[ok]
        mp = arrs.mapping.cpu().numpy()[ok]
        self.write_batch({
            "signal": arrs.signal.cpu().numpy()[ok],
            "sequence": seq[:, :sw],
            "sequence_to_signal_mapping": mp[:, :mw],
            "sequence_lengths": lens[ok],
            "labels": np.asarray(arrs.labels, np.int64)[ok],
        })
        return int(ok.sum())

    def shuffle(self, batch_size=100_000):
        """Random permutation of the written rows, every array with the same permutation (:1420-1469)."""
        if self.mode != "w":
            raise RemoraError("Cannot write when mode is not 'w'")
        a0, a1 = int(self.metadata["dataset_start"]), int(self.metadata["dataset_end"])
        perm = np.random.permutation(a1 - a0)
        for a in self.arrays.values():
            view = a[a0:a1]
            src = view.copy()
            for st in range(0, a1 - a0, batch_size):
                view[st : st + batch_size] = src[perm[st : st + batch_size]]
            a.flush()

    def flush(self):
        for a in self.arrays.values():
            if hasattr(a, "flush"):
                a.flush()
        if self.mode == "w":
            self.write_metadata()

    @property
    def size(self):
        return int(self.metadata["dataset_end"]) - int(self.metadata["dataset_start"])

    @property
    def chunk_len(self):
        return sum(self.chunk_context)

    def load_batch(self, st, en):
        """Rows [st, en) of the dataset as writable numpy arrays, trimmed to the loaded contexts."""
        from .data_chunks_core import trim_sb_chunk_context_core

        a0 = int(self.metadata["dataset_start"])
        b = {k: np.array(v[a0 + st : a0 + en]) for k, v in self.arrays.items()}
        seq_diff = self.stored_kmer_context_bases[0] - self.kmer_context_bases[0]
        if seq_diff > 0:  # :1512-1534 (the trailing trim happens in the encode via the smaller ka)
            b["sequence"][:, :-seq_diff] = b["sequence"][:, seq_diff:].copy()
        if self.chunk_context != self.stored_chunk_context:  # :1536-1576
            st_diff = self.stored_chunk_context[0] - self.chunk_context[0]
            new_en = self.stored_chunk_context[0] + self.chunk_context[1]
            b["signal"] = np.ascontiguousarray(b["signal"][:, :, st_diff:new_en])
            b["sequence_to_signal_mapping"] = (b["sequence_to_signal_mapping"] - st_diff).astype(np.int16)
            trim_sb_chunk_context_core(*self.stored_chunk_context, *self.chunk_context,
                                       sum(self.kmer_context_bases), b["sequence"],
                                       b["sequence_to_signal_mapping"], b["sequence_lengths"])
        return b

    def iter_batches(self):
        for st in range(0, self.size, self.batch_size):
            yield self.load_batch(st, min(st + self.batch_size, self.size))

    def get_label_counts(self):
        a0, a1 = int(self.metadata["dataset_start"]), int(self.metadata["dataset_end"])
        return np.bincount(self.arrays["labels"][a0:a1], minlength=len(self.metadata["mod_bases"]) + 1)


def validate_dataset(dataset, model):
    """Per-chunk calls for an on-disk dataset: logits through the fused path, argmax tally on the GPU
    (label counts) and the confusion matrix against the stored labels — the numbers
    `remora validate from_remora_dataset` derives (src/remora/validate.py:42-66, 190-259)."""
    num_out = model.num_out
    counts = np.zeros(num_out, np.int64)
    conf = np.zeros((num_out, num_out), np.int64)
    logits = []
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
