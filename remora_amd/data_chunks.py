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


def extract_chunk_arrays(reads, chunk_context, kmer_context_bases, base_start_justify=False, offset=0,
                         engine=None):
    """Chunk arrays for a list of RemoraRead objects whose `focus_bases` are set.
    GPU counterpart of iter_chunks + extract_chunk + write_chunk for every focus base
    (src/remora/data_chunks.py:425-466, :331-423, :1376-1418).  Returns (ChunkArrays, sig)
    where sig is the normalised signal of all reads (float32, CUDA)."""
    torch = _torch()
    eng = engine if engine is not None else get_engine()
    dev = eng.torch_device
    lib = L.lib()
    nr = len(reads)
    sig_off = np.zeros(nr + 1, np.int64)
    seq_off = np.zeros(nr + 1, np.int64)
    foc_off = np.zeros(nr + 1, np.int64)
    for i, r in enumerate(reads):
        sig_off[i + 1] = sig_off[i] + r.dacs.size
        seq_off[i + 1] = seq_off[i] + r.int_seq.size
        foc_off[i + 1] = foc_off[i] + (0 if r.focus_bases is None else len(r.focus_bases))
        if r.seq_to_sig_map.size != r.int_seq.size + 1:
            raise RemoraError(f"Invalid read: seq ({r.int_seq.size}) and mapping ({r.seq_to_sig_map.size}) sizes incompatible")
    n_chunks = int(foc_off[-1])
    cat = lambda arrs, dt: (np.concatenate([np.asarray(a).ravel() for a in arrs]).astype(dt, copy=False)
                            if arrs else np.zeros(0, dt))
    dacs = cat([r.dacs for r in reads], np.int16)
    s2s = cat([r.seq_to_sig_map for r in reads], np.int64)
    iseq = cat([r.int_seq for r in reads], np.int8)
    focus = cat([r.focus_bases for r in reads if r.focus_bases is not None and len(r.focus_bases)], np.int64)
    shift = np.array([float(r.shift) for r in reads], np.float64)
    scale = np.array([float(r.scale) for r in reads], np.float64)
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = dict(dacs=to_dev(dacs), sig_off=to_dev(sig_off), s2s=to_dev(s2s), iseq=to_dev(iseq),
             seq_off=to_dev(seq_off), shift=to_dev(shift), scale=to_dev(scale),
             focus=to_dev(focus if focus.size else np.zeros(1, np.int64)), foc_off=to_dev(foc_off))
    rs = L.Reads(nr, d["dacs"].data_ptr(), d["sig_off"].data_ptr(), d["s2s"].data_ptr(), d["iseq"].data_ptr(),
                 d["seq_off"].data_ptr(), d["shift"].data_ptr(), d["scale"].data_ptr(), d["focus"].data_ptr(),
                 d["foc_off"].data_ptr(), int(chunk_context[0]), int(chunk_context[1]),
                 int(kmer_context_bases[0]), int(kmer_context_bases[1]), int(bool(base_start_justify)), int(offset))
    L_chunk = int(chunk_context[0]) + int(chunk_context[1])
    sig = torch.empty(max(int(sig_off[-1]), 1), dtype=torch.float32, device=dev)
    geo = torch.empty((max(n_chunks, 1), 6), dtype=torch.int64, device=dev)
    max_sl = ctypes.c_int64(0)
    L.check(lib.rmr_chunk_geometry(eng.handle, ctypes.byref(rs), sig.data_ptr(), geo.data_ptr(),
                                   ctypes.byref(max_sl), L.MEM_DEVICE))
    sig = sig[: int(sig_off[-1])]
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
    # the kernels above read the staged inputs in `d`: keep them alive until the stream drains
    eng.synchronize()
    labels = np.full(n_chunks, -1, np.int64)
    for i, r in enumerate(reads):
        if r.labels is not None and foc_off[i + 1] > foc_off[i]:
            labels[foc_off[i] : foc_off[i + 1]] = np.asarray(r.labels)[np.asarray(r.focus_bases)]
    return ChunkArrays(signal, sequence, mapping, lengths, rfb, labels, geo, kmer_context_bases, chunk_context), sig


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


class CoreRemoraDataset:
    _core_dtypes = {"signal": np.float32, "sequence": np.int8, "sequence_to_signal_mapping": np.int16,
                    "sequence_lengths": np.int16, "labels": np.int64}

    def __init__(self, data_path, override_metadata=None, batch_size=2048):
        self.data_path = data_path
        self.batch_size = int(batch_size)
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
            if os.path.getsize(path) != int(np.prod(shapes[name])) * np.dtype(dt).itemsize:
                raise RemoraError(f"{path} does not have the size metadata.jsn implies")
            self.arrays[name] = np.memmap(path, dt, mode="r", shape=shapes[name])

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
