"""Helpers shared by the tests and tools/gen_golden.py: chunk datasets are kept in the golden npz files as
their written rows + the metadata.jsn text and are turned back into dataset directories on demand."""
import json
import os

import numpy as np

CORE = ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths", "labels")


def dataset_keys(prefix):
    return [f"{prefix}__{n}" for n in CORE]


def materialise_dataset(g, prefix, out_dir):
    """Write the dataset stored under `prefix` in the npz mapping `g` as a CoreRemoraDataset directory:
    raw C-order arrays `<name>.npy` / `extra_<name>.npy` with allocate_size rows (rows past the written
    ones are zero), metadata.jsn verbatim, kmer_table.npy when a level table was stored."""
    os.makedirs(out_dir, exist_ok=True)
    text = str(g[f"{prefix}__metadata_jsn"])
    md = json.loads(text)
    with open(os.path.join(out_dir, "metadata.jsn"), "w") as fh:
        fh.write(text)
    alloc = int(md["allocate_size"])
    names = list(CORE) + list((md.get("extra_arrays") or {}).keys())
    for name in names:
        rows = np.asarray(g[f"{prefix}__{name}"])
        full = np.zeros((alloc,) + rows.shape[1:], dtype=rows.dtype)
        full[: rows.shape[0]] = rows
        fn = f"{name}.npy" if name in CORE else f"extra_{name}.npy"
        with open(os.path.join(out_dir, fn), "wb") as fh:
            fh.write(np.ascontiguousarray(full).tobytes())
    if f"{prefix}__kmer_table" in g:
        np.save(os.path.join(out_dir, "kmer_table.npy"), np.asarray(g[f"{prefix}__kmer_table"]), allow_pickle=False)
    return out_dir


def dataset_rows(ds_dir, md=None):
    """(metadata dict, {array name: written rows}) of a dataset directory, padding columns masked (sequence -> -1,
    mapping -> 0 beyond each chunk's length) so that rows compare independently of uninitialised padding."""
    if md is None:
        with open(os.path.join(ds_dir, "metadata.jsn")) as fh:
            md = json.load(fh)
    alloc, msl = int(md["allocate_size"]), int(md["max_seq_len"])
    n0, n1 = int(md["dataset_start"]), int(md["dataset_end"])
    kb, ka = md["_stored_kmer_context_bases"] or md["kmer_context_bases"]
    L = sum(md["_stored_chunk_context"] or md["chunk_context"])
    spec = {"signal": (np.float32, (alloc, 1, L)), "sequence": (np.int8, (alloc, msl + kb + ka)),
            "sequence_to_signal_mapping": (np.int16, (alloc, msl + 1)), "sequence_lengths": (np.int16, (alloc,)),
            "labels": (np.int64, (alloc,))}
    for name, (dt, _desc) in (md.get("extra_arrays") or {}).items():
        spec[name] = (np.dtype(dt), (alloc,))
    rows = {}
    for name, (dt, shape) in spec.items():
        fn = f"{name}.npy" if name in CORE else f"extra_{name}.npy"
        rows[name] = np.array(np.memmap(os.path.join(ds_dir, fn), dt, mode="r", shape=shape)[n0:n1])
    lens = rows["sequence_lengths"].astype(int)
    cols = np.arange(rows["sequence"].shape[1])[None, :]
    rows["sequence"][cols >= (lens[:, None] + kb + ka)] = -1
    cols = np.arange(rows["sequence_to_signal_mapping"].shape[1])[None, :]
    rows["sequence_to_signal_mapping"][cols > lens[:, None]] = 0
    return md, rows


def pod5_reads_cpu(pod5_path, read_ids=None):
    """The reads of a POD5 file decoded WITHOUT a GPU: table access and zstd from remora_amd.io, the VBZ layer from
    the C oracle.  For CPU-side tests and for tools/gen_golden.py only (the product decodes on the GPU)."""
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import oracle as O
    from remora_amd import io as rio

    f = rio.Pod5File(pod5_path)
    out = []
    for rid in f.read_ids:
        if read_ids is not None and rid not in read_ids:
            continue
        parts = [O.vbz_decode(bytes(rio._zstd_decompress(blob)), n) for blob, n in f.signal_rows(rid)]
        off, scale = f.calibration(rid)
        out.append(rio.Pod5Read(rid, np.concatenate(parts) if len(parts) > 1 else parts[0], off, scale))
    return out


class RecordWithTags:
    """A BAM record seen through edited tags, for the branches of Read.add_alignment (src/remora/io.py:1972-2044) that the
    reference's test files never take: tags dropped (`sm` / `sd` -> median / MAD scaling), added or replaced (`sp`, `pi`
    -> split reads), another query name (the child of a split read).  Everything else is the wrapped record's.  Used by
    tools/gen_golden.py (fed to the reference's add_alignment) and by the tests (fed to remora_amd's) alike."""

    _HOT = ("mv", "ts", "ns", "sp", "sm", "sd", "pi")

    def __init__(self, rec, drop=(), add=None, query_name=None):
        add = dict(add or {})
        self._rec = rec
        self.tags = [(k, v) for k, v in rec.tags if k not in drop and k not in add] + list(add.items())
        self.query_name = rec.query_name if query_name is None else query_name

    def hot_tags(self):
        hot = dict(self._rec.hot_tags()) if hasattr(self._rec, "hot_tags") else dict(self._rec.tags)
        names = {k for k, _ in self.tags}
        hot = {k: v for k, v in hot.items() if k in names}
        hot.update({k: v for k, v in self.tags if k in self._HOT and k not in hot})
        for k, v in self.tags:  # replaced values win
            if k in self._HOT and k != "mv":
                hot[k] = v
        return hot

    def get_tag(self, name):
        for k, v in self.tags:
            if k == name:
                return v
        raise KeyError(name)

    def __getattr__(self, name):  # flag, is_reverse, reference_name, query_sequence, to_dict, cigartuples, ...
        return getattr(self._rec, name)


REAL_READ_BRANCHES = ("rev", "nosmsd", "split", "pa")
PA_SCALING = (87.5, 14.25)  # picoamp -> zero-centred picoamp (shift, scale) of the "pa" branch
SPLIT_PREFIX = 137          # samples of the parent read in front of the child of the "split" branch


def real_read_branch(variant, pod, rec, rng_seed):
    """(read id, dacs, record, add_alignment keyword arguments, reverse_signal at read creation) of one branch case, built the
    same way for the reference (generator) and for remora_amd (test):
      rev     the signal is read 3'->5' (RNA): reversed when the read is made (io.py:466) and around the sp/ts/ns trims
      nosmsd  no sm / sd tags: the pA -> norm scaling is median / MAD of the pA signal (io.py:1851-1856)
      split   the record is the child of a longer parent read: `sp` samples are cut off the front, `pi` names the parent
      pa      pa_scaling given: into_remora_read composes the zero-centred pA scaling instead of sm / sd (io.py:2159-2167)"""
    if variant == "rev":
        return pod.read_id, pod.signal[::-1], rec, dict(reverse_signal=True), True
    if variant == "nosmsd":
        return pod.read_id, pod.signal, RecordWithTags(rec, drop=("sm", "sd")), {}, False
    if variant == "split":
        front = np.random.default_rng(rng_seed).integers(300, 700, SPLIT_PREFIX).astype(pod.signal.dtype)
        parent = "parent-of-" + pod.read_id
        return parent, np.concatenate([front, pod.signal]), RecordWithTags(rec, add={"sp": SPLIT_PREFIX, "pi": parent}), {}, False
    if variant == "pa":
        return pod.read_id, pod.signal, rec, dict(pa_scaling=PA_SCALING), False
    raise ValueError(variant)
