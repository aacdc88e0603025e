"""Read-side helper of the hot path: move-table expansion (src/remora/io.py:394-407)."""
import ctypes
import os

import numpy as np

from . import RemoraError
from . import _lib as L
from .engine import get_engine


def parse_move_tag(mv_tag, sig_len, seq_len=None, check=True, reverse_signal=False, engine=None):
    """Same signature and return value as remora.io.parse_move_tag:
    (query_to_signal int64[#moves+1], mv_table, stride).  The compaction runs on the GPU
    (rmr_parse_moves); raises RemoraError("Move table discordant with basecalls" / "... with
    signal") exactly where the reference does."""
    eng = engine if engine is not None else get_engine()
    mv = np.ascontiguousarray(mv_tag, dtype=np.int8)
    if mv.size < 1:
        raise RemoraError("empty move tag")
    q2s = np.empty(mv.size + 1, np.int64)
    n_out = ctypes.c_int64(0)
    rc = L.lib().rmr_parse_moves(eng.handle, mv.ctypes.data, mv.size, int(sig_len),
                                 -1 if seq_len is None else int(seq_len), int(bool(check)),
                                 int(bool(reverse_signal)), q2s.ctypes.data, ctypes.byref(n_out), L.MEM_HOST)
    L.check(rc)
    stride = int(mv[0])
    return q2s[: n_out.value].copy(), mv[1:].astype(np.int64), stride


_MOVE_ERRORS = {-4: "Move table discordant with basecalls", -5: "Move table discordant with signal"}


def parse_move_tags(mv_tags, sig_lens, seq_lens=None, check=True, reverse_signal=False, engine=None):
    """parse_move_tag for a batch of reads in ONE launch (rmr_parse_moves_batch): per read either the tuple
    parse_move_tag returns or the RemoraError it would raise (returned, not raised)."""
    eng = engine if engine is not None else get_engine()
    n = len(mv_tags)
    if n == 0:
        return []
    mvs = [np.ascontiguousarray(m, dtype=np.int8) for m in mv_tags]
    off = np.zeros(n + 1, np.int64)
    np.cumsum([m.size for m in mvs], out=off[1:])
    cat = np.concatenate(mvs) if off[-1] else np.zeros(1, np.int8)
    sl = np.ascontiguousarray(sig_lens, np.int64)
    ql = np.full(n, -1, np.int64) if seq_lens is None else np.asarray([-1 if x is None else int(x) for x in seq_lens], np.int64)
    q2s = np.empty(max(int(off[-1]), 1), np.int64)
    counts, status = np.zeros(n, np.int64), np.zeros(n, np.int32)
    L.check(L.lib().rmr_parse_moves_batch(eng.handle, cat.ctypes.data, off.ctypes.data, sl.ctypes.data, ql.ctypes.data, n,
                                          int(bool(check)), int(bool(reverse_signal)), q2s.ctypes.data, counts.ctypes.data,
                                          status.ctypes.data, L.MEM_HOST))
    out = []
    for i, m in enumerate(mvs):
        if status[i] != 0:
            out.append(RemoraError(_MOVE_ERRORS.get(int(status[i]), "empty move tag" if m.size < 1 else
                                                    f"move table stride {int(m[0])}")))
        else:
            out.append((q2s[off[i] : off[i] + counts[i]].copy(), m[1:].astype(np.int64), int(m[0])))
    return out


# =======================================================================================
# POD5 + BAM ingest without pysam / pod5: the reference reads these with
# pod5.DatasetReader (src/remora/io.py:441-474) and pysam (:184-358); both formats are simple
# enough to parse with the standard library + pyarrow:
#   BAM  = BGZF (concatenated gzip members) around fixed binary records (SAM spec §4.2)
#   POD5 = a container of three Arrow IPC files (signal / run info / reads tables); signal
#          rows are "VBZ": zstd( streamvbyte-16( zigzag( delta( int16 samples ))))
# =======================================================================================
import array
import dataclasses
import gzip
import struct
import threading

import re

_MD_TOKEN = re.compile(r"(\d+)|(\^[A-Za-z]+)|([A-Za-z])")
_MD_VALID = re.compile(r"(?:\d+|\^[A-Za-z]+|[A-Za-z])*")
_SEQ_NT16 = "=ACMGRSVTWYHKDBN"
_CIGAR_OPS = "MIDNSHP=XB"


@dataclasses.dataclass
class BamRecord:
    """The pysam.AlignedSegment attributes the hot path touches (src/remora/io.py:1972-2084)."""

    query_name: str
    flag: int
    reference_id: int
    reference_name: str
    reference_start: int
    mapping_quality: int
    cigartuples: list
    query_sequence: str
    query_qualities: bytes
    tags: list  # [(name, value)] in file order
    raw: bytes = b""          # the record as stored (without the leading block_size)
    tags_offset: int = 0      # byte offset of the tag region inside `raw`
    tag_spans: list = None    # [(name, start, end)] byte spans inside the tag region

    @property
    def is_reverse(self):
        return bool(self.flag & 0x10)

    @property
    def is_unmapped(self):
        return bool(self.flag & 0x4)

    @property
    def is_secondary(self):
        return bool(self.flag & 0x100)

    @property
    def is_supplementary(self):
        return bool(self.flag & 0x800)

    def get_tag(self, name):
        for k, v in self.tags:
            if k == name:
                return v
        raise KeyError(name)

    def get_reference_sequence(self):
        """Reference bases spanned by the alignment, rebuilt from the query, the CIGAR and the MD tag as
        pysam.AlignedSegment.get_reference_sequence does (mismatched bases in lower case); ValueError
        without an MD tag."""
        try:
            md = self.get_tag("MD")
        except KeyError:
            raise ValueError("MD tag not present")
        # reference-consuming columns of the alignment: query base for M/=/X, placeholder for D/N
        parts, q = [], 0
        for op, ln in self.cigartuples:
            if op in (0, 7, 8):
                parts.append(self.query_sequence[q : q + ln])
                q += ln
            elif op in (1, 4):
                q += ln
            elif op in (2, 3):
                parts.append("-" * ln)
        cols = "".join(parts)
        out, i = [], 0
        tokens = _MD_TOKEN.findall(md)
        for run, deleted, mismatch in tokens:
            if run:
                n = int(run)
                out.append(cols[i : i + n])
                i += n
            elif deleted:
                out.append(deleted[1:].upper())
                i += len(deleted) - 1
            else:
                out.append(mismatch.lower())
                i += 1
        if i != len(cols) or _MD_VALID.fullmatch(md) is None:
            raise ValueError("MD tag and CIGAR disagree about the reference span")
        return "".join(out)

    def to_dict(self):
        return {"name": self.query_name, "flag": str(self.flag), "ref_name": self.reference_name or "*",
                "ref_pos": str(self.reference_start + 1), "map_quality": str(self.mapping_quality),
                "seq": self.query_sequence}


def _parse_tags(buf, spans=None):
    tags, p, n = [], 0, len(buf)
    scalar = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}
    while p < n:
        p0 = p
        name = buf[p : p + 2].decode()
        t = chr(buf[p + 2])
        p += 3
        if t == "A":
            val = chr(buf[p]); p += 1
        elif t in scalar:
            sz = struct.calcsize(scalar[t])
            val = struct.unpack_from(scalar[t], buf, p)[0]; p += sz
        elif t in "ZH":
            e = buf.index(b"\x00", p)
            val = buf[p:e].decode(); p = e + 1
        elif t == "B":
            sub = chr(buf[p]); cnt = struct.unpack_from("<i", buf, p + 1)[0]; p += 5
            dt = {"c": np.int8, "C": np.uint8, "s": np.int16, "S": np.uint16, "i": np.int32, "I": np.uint32,
                  "f": np.float32}[sub]
            nbytes = cnt * np.dtype(dt).itemsize
            # pysam hands B tags over as array.array: same here (BAM is little-endian, as are the hosts this runs on)
            val = array.array({"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub])
            val.frombytes(bytes(buf[p : p + nbytes]))
            p += nbytes
        else:
            raise RemoraError(f"unknown BAM tag type {t!r}")
        tags.append((name, val))
        if spans is not None:
            spans.append((name, p0, p))
    return tags


_NT16 = np.frombuffer(_SEQ_NT16.encode(), dtype="S1")


def _resolve_long_cigar(cig, l_seq, get_tags):
    """SAM spec 4.2.2: a CIGAR with more than 65535 operations lives in the CG:B,I tag and the record itself holds
    the placeholder <l_seq>S<ref_len>N; htslib (hence pysam, which the reference reads alignments with) resolves it
    when the record is read.  `cig`: uint32 operations as stored; returns the operations to use."""
    # htslib's bam_tag2cigar: a mapped record whose FIRST operation is <l_seq>S carries the real CIGAR in CG (it does not
    # look at the second operation)
    if cig.size >= 1 and (int(cig[0]) & 0xF) == 4 and (int(cig[0]) >> 4) == l_seq:
        for k, v in get_tags():
            if k == "CG" and isinstance(v, array.array) and v.typecode == "I" and len(v):
                return np.frombuffer(v.tobytes(), dtype="<u4")
    return cig


def _read_exact(fh, n):
    buf = fh.read(n)
    if len(buf) != n:
        raise RemoraError("truncated BAM file")
    return buf


class _NativeBamRecord(BamRecord):
    """BamRecord filled from a native batch (rmr_bam_read_batch): the hot fields are there at once, everything else
    (full tag list, tag byte spans, CIGAR tuples, qualities) is decoded from the record bytes on first use."""

    def __init__(self, query_name, flag, reference_id, reference_name, reference_start, mapping_quality, query_sequence,
                 raw, tags_offset, n_cigar, hot, ref_seq, voffset=-1, cigar=None):
        self.query_name, self.flag, self.reference_id, self.reference_name = query_name, flag, reference_id, reference_name
        self.reference_start, self.mapping_quality, self.query_sequence = reference_start, mapping_quality, query_sequence
        self.raw, self.tags_offset, self._n_cigar, self._hot, self._ref_seq = raw, tags_offset, n_cigar, hot, ref_seq
        self.voffset = voffset  # BGZF virtual offset of the record in its file
        self._cigar = cigar  # uint32 operations from the native reader (the CG tag's when the record holds a placeholder)

    def _parse_all_tags(self):
        spans = []
        self._tags = _parse_tags(self.raw[self.tags_offset :], spans)
        self._tag_spans = spans  # byte spans of ALL stored tags (records are re-emitted as stored)
        if self._cigar is not None and len(self._cigar) != self._n_cigar:
            # the long CIGAR was taken from CG: htslib removes the tag when it does that, so pysam callers never see it
            self._tags = [(k, v) for k, v in self._tags if k != "CG"]

    @property
    def tags(self):
        if "_tags" not in self.__dict__:
            self._parse_all_tags()
        return self._tags

    @property
    def tag_spans(self):
        if "_tag_spans" not in self.__dict__:
            self._parse_all_tags()
        return self._tag_spans

    @property
    def cigartuples(self):
        if "_cigartuples" not in self.__dict__:
            cig = self._cigar
            if cig is None:
                cig = _resolve_long_cigar(np.frombuffer(self.raw, dtype="<u4", count=self._n_cigar, offset=32 + self.raw[8]),
                                          len(self.query_sequence), lambda: self.tags)
            self._cigartuples = list(zip((cig & 0xF).tolist(), (cig >> 4).tolist()))
        return self._cigartuples

    @property
    def query_qualities(self):
        l_seq = len(self.query_sequence)
        q = 32 + self.raw[8] + 4 * self._n_cigar + (l_seq + 1) // 2
        return bytes(self.raw[q : q + l_seq])

    def hot_tags(self):
        """{name: value} of the tags Read.add_alignment looks at (mv, ts, ns, sp, sm, sd, pi), without parsing the
        whole tag region; mv as an int8 numpy array."""
        return self._hot

    def get_reference_sequence(self):
        if self._ref_seq is None:  # no MD tag, unmapped, or not requested from the native reader: python path decides
            return BamRecord.get_reference_sequence(self)
        return self._ref_seq


def _iter_bam_records_native(bam_path, want_ref, batch, voffsets=None, start_voffset=None, max_records=None, end_voffset=None):
    """Records of a BAM file from the native reader; with `voffsets` only the records at those virtual offsets
    (one seek + one record each); with `start_voffset` and `max_records` / `end_voffset` the contiguous run of records
    that starts there (a rank's share of the file: `bam_shard` counts records, `bam_byte_shard` names the record the
    next share begins with - a run that passes that record without meeting it is an error); otherwise the whole file."""
    lib = L.lib()
    h = ctypes.c_void_p()
    L.check(lib.rmr_bam_open(str(bam_path).encode(), ctypes.byref(h)))
    try:
        if voffsets is not None:
            for vo in voffsets:
                L.check(lib.rmr_bam_seek(h, int(vo)))
                yield from _native_batches(lib, h, want_ref, 1, once=True)
            return
        if start_voffset is not None:
            L.check(lib.rmr_bam_seek(h, int(start_voffset)))
        if end_voffset is None:
            yield from _native_batches(lib, h, want_ref, batch, limit=max_records)
            return
        end_voffset = int(end_voffset)
        for rec in _native_batches(lib, h, want_ref, batch):
            if rec.voffset >= end_voffset:
                if rec.voffset != end_voffset:
                    raise RemoraError(f"{bam_path}: the records of this share run past virtual offset {end_voffset} without one "
                                      f"starting there - the next share's start was guessed wrong (REMORA_AMD_BAM_SHARD=scan "
                                      f"splits by an exact pass over the file instead)")
                return
            yield rec
        raise RemoraError(f"{bam_path}: end of file before the record at virtual offset {end_voffset} where the next share begins")
    finally:
        lib.rmr_bam_close(h)


def bam_guess_start(bam_path, file_offset):
    """Virtual offset of the first record that starts in a BGZF member at or behind byte `file_offset` of the file, or
    None (rmr_bam_guess_start: found from the bytes there, not from the records in front)."""
    lib = L.lib()
    h = ctypes.c_void_p()
    L.check(lib.rmr_bam_open(str(bam_path).encode(), ctypes.byref(h)))
    try:
        v = ctypes.c_int64()
        L.check(lib.rmr_bam_guess_start(h, int(file_offset), ctypes.byref(v)))
        return None if v.value < 0 else int(v.value)
    finally:
        lib.rmr_bam_close(h)


def shard_of(bam_path, rank, world):
    """What `iter_bam_records(shard=...)` reads for (rank, world): ("bytes", start, end) - a share by byte range
    (`bam_byte_shard`, the default) - or, with REMORA_AMD_BAM_SHARD=scan, (start, n_records) from one exact pass over
    the file (`bam_shard`)."""
    if os.environ.get("REMORA_AMD_BAM_SHARD", "bytes") == "scan":
        return bam_shard(bam_path, rank, world)
    return ("bytes",) + bam_byte_shard(bam_path, rank, world)


def bam_byte_shard(bam_path, rank, world):
    """(start_voffset, end_voffset) of rank `rank`'s share when `world` workers split `bam_path` by BYTE ranges: the
    records that start in BGZF members at or behind byte size * rank / world and in front of size * (rank + 1) / world.
    No pass over the file and no coordinator: both ends come from `bam_guess_start`, a pure function of the file, so
    rank r's end IS rank r + 1's start.  end None = to the end of the file; start None = nothing for this rank.  Both
    marks are verified before the rank starts (`_verify_share_mark`: an independent guess 1 MiB in front, chained forward,
    has to land on the mark) and again by the chain of the worker in front, which has to end on it
    (`_iter_bam_records_native`)."""
    if world <= 1:
        return bam_guess_start(bam_path, 0), None
    size = os.path.getsize(bam_path)
    cut = lambda r: size * int(r) // int(world)  # noqa: E731
    start = bam_guess_start(bam_path, cut(rank))
    if start is None:
        return None, None
    end = None if rank == world - 1 else bam_guess_start(bam_path, cut(rank + 1))
    if end is not None and end == start:
        return None, None
    if os.environ.get("REMORA_AMD_BAM_SHARD_VERIFY", "1") != "0":
        # both ends checked BEFORE any work is done (the chain of the worker in front would only find a wrong guess at the
        # very end of its share, behind all of its GPU work): a second, independent guess a little in front of the mark is
        # chained forward record by record and has to arrive exactly on it
        for mark, at in ((start, cut(rank)), (end, cut(rank + 1))):
            if mark is not None and at > 0:
                _verify_share_mark(bam_path, mark, at)
    return start, end


def _verify_share_mark(bam_path, mark, file_offset, back=1 << 20):
    """Raise unless the records chained from an independent starting point in front of byte `file_offset` (the guess `back`
    bytes earlier, or the file's first record) meet virtual offset `mark` exactly."""
    first = bam_guess_start(bam_path, 0)
    probe = None
    # a starting point in front of the mark: 1, 4, 16, 64 MiB back (records are tens of KiB: the first window almost always
    # holds one); only a file smaller than the window is chained from its first record - never a whole large BAM per rank
    # and mark (the rank in front verifies the boundary once more when it gets there, which stays the authoritative check)
    for k in range(4):
        window = back << (2 * k)
        if file_offset <= window:
            probe = first
            break
        probe = bam_guess_start(bam_path, max(int(file_offset) - window, 0))
        if probe is not None and probe < mark:
            break
        probe = None
    if probe is None:
        if file_offset <= (back << 6):
            return  # a small file whose first record cannot be guessed either: nothing independent to chain from (the rank in
            # front verifies the boundary when it gets there, which stays the authoritative check)
        raise RemoraError(f"{bam_path}: no record start found within {back << 6} bytes in front of the share boundary at byte "
                          f"{file_offset} - REMORA_AMD_BAM_SHARD=scan splits by an exact pass over the file instead")
    if probe == mark:
        return
    lib = L.lib()
    h = ctypes.c_void_p()
    L.check(lib.rmr_bam_open(str(bam_path).encode(), ctypes.byref(h)))
    try:
        L.check(lib.rmr_bam_seek(h, int(probe)))
        for rb in _native_raw_batches(lib, h, False, 64):
            past = np.nonzero(rb.voffset >= mark)[0]
            if past.size:
                if int(rb.voffset[int(past[0])]) == mark:
                    return
                break
        raise RemoraError(f"{bam_path}: the share boundary guessed at byte {file_offset} (virtual offset {mark}) is not a record start "
                          f"(records chained from virtual offset {probe} pass it) - REMORA_AMD_BAM_SHARD=scan splits by an exact "
                          f"pass over the file instead")
    finally:
        lib.rmr_bam_close(h)


def bam_scan(bam_path, every=64, inflate_threads=0):
    """(marks, n_records): the BGZF virtual offset of records 0, every, 2 every, ... of `bam_path` and the number of
    records.  One light pass over the file (rmr_bam_scan: BGZF inflate + the block_size fields).  `inflate_threads`: BGZF
    inflate workers of the pass (0 = the reader's default)."""
    lib = L.lib()
    cap = 1 << 16
    while True:  # rmr_bam_open leaves the handle at the first record
        h = ctypes.c_void_p()
        L.check(lib.rmr_bam_open_threads(str(bam_path).encode(), int(inflate_threads), ctypes.byref(h)))
        try:
            marks = np.empty(cap, np.int64)
            n = ctypes.c_int64()
            L.check(lib.rmr_bam_scan(h, int(every), marks.ctypes.data, cap, ctypes.byref(n)))
        finally:
            lib.rmr_bam_close(h)
        n_marks = (n.value + every - 1) // every
        if n_marks <= cap:
            return marks[:n_marks].copy(), int(n.value)
        cap = int(n_marks)  # more marks than the first guess: scan again with room for all of them


SCAN_ENV = "REMORA_AMD_BAM_SCAN"  # set by dist.launch_ranks: the launcher scans the BAM once for all of its ranks


def _scan_key(bam_path, every):
    st = os.stat(bam_path)
    return np.array([st.st_size, st.st_mtime_ns, int(every)], np.int64)


def write_bam_scan(bam_path, out_path, every=64):
    """Scan `bam_path` and leave (marks, n_records) in `out_path` for the ranks of this launch (`bam_shard` picks it up
    through REMORA_AMD_BAM_SCAN).  The file appears atomically; a scan that fails leaves a marker instead, and the ranks
    scan for themselves (and report the error in their own words)."""
    tmp = f"{out_path}.tmp{os.getpid()}"
    try:
        # this process has nothing else to do while its ranks start: every core it may use inflates (unless the user said
        # how many: RMR_BAM_INFLATE_THREADS, which the reader reads for a count of 0)
        threads = 0 if os.environ.get("RMR_BAM_INFLATE_THREADS") else max(8, min(32, _eff_cpus()))
        marks, n = bam_scan(bam_path, every, inflate_threads=threads)
        payload = dict(key=_scan_key(bam_path, every), marks=marks, n=np.int64(n))
    except Exception:  # noqa: BLE001 - whatever it is, the ranks will meet it themselves
        payload = dict(key=np.zeros(3, np.int64), marks=np.zeros(0, np.int64), n=np.int64(-1))
    with open(tmp, "wb") as fh:
        np.savez(fh, **payload)
    os.replace(tmp, out_path)


def _launcher_bam_scan(bam_path, every, wait_s=120.0):
    """(marks, n_records) from the launcher's scan of this very file, or None (no launcher scan announced, it failed,
    it is about another file, or it did not appear in `wait_s`)."""
    import time

    path = os.environ.get(SCAN_ENV)
    if not path:
        return None
    deadline = time.monotonic() + wait_s
    while not os.path.exists(path):
        if time.monotonic() > deadline:
            return None
        time.sleep(0.005)
    try:
        with np.load(path) as z:
            if int(z["n"]) < 0 or not np.array_equal(z["key"], _scan_key(bam_path, every)):
                return None
            return z["marks"].copy(), int(z["n"])
    except (OSError, ValueError, KeyError):
        return None


def bam_shard(bam_path, rank, world, every=64):
    """(start_voffset, n_records) of rank `rank`'s contiguous share of the alignments of `bam_path` when `world` workers
    split the file between them (dist.shard_range over marks set every `every` records; shares differ by less than
    `every` records unless the file has fewer than `world` marks).  The marks come from one light pass over the file
    (`bam_scan`): the launcher of the ranks makes it once for all of them (dist.launch_ranks); under a foreign launcher
    (torchrun) every worker makes it by itself - no coordinator, no index file.  n_records is 0 (and the offset None)
    for a rank that gets nothing."""
    from .dist import shard_range

    if world <= 1:
        return None, None
    import time

    t_scan = time.perf_counter()
    got = _launcher_bam_scan(bam_path, every)
    marks, n = got if got is not None else bam_scan(bam_path, every)
    if os.environ.get("RMR_INFER_TIMING"):
        import sys

        print(f"[bam_shard rank {rank}/{world}] {n} records, marks from {'the launcher' if got is not None else 'its own scan'}: "
              f"{time.perf_counter() - t_scan:.2f}s", file=sys.stderr, flush=True)
    m0, m1 = shard_range(len(marks), rank, world)
    if m1 <= m0:
        return None, 0
    return int(marks[m0]), int(min(m1 * every, n) - m0 * every)


class RawBamBatch:
    """One rmr_bam_read_batch result as flat arrays (copies: the native buffers are reused by the next call): per record
    flag, ref_id, pos, mapq, n_cigar, ts, ns, sp (int32), sm, sd (float32), has (bit set of the hot tags present), ref_ok,
    tags_off, voffset (int64) and the offset arrays (int64[n + 1]) into the blobs raw (record bytes without block_size),
    names, seq (ASCII bases), mv (int8 move tables), pi, refseq, cigar."""

    __slots__ = ("n", "flag", "ref_id", "pos", "mapq", "n_cigar", "ts", "ns", "sp", "sm", "sd", "has", "ref_ok", "tags_off", "voffset",
                 "raw_off", "name_off", "seq_off", "mv_off", "pi_off", "refseq_off", "cigar_off", "raw", "names", "seq", "mv", "pi",
                 "refseq", "cigar", "want_ref")

    def head(self, k):
        """The first k records (offset arrays cut, blobs shared)."""
        out = RawBamBatch()
        for f in self.__slots__:
            v = getattr(self, f)
            if f == "n":
                v = int(k)
            elif f.endswith("_off") and f != "tags_off":
                v = v[: k + 1]
            elif isinstance(v, np.ndarray) and f not in ("mv", "cigar"):
                v = v[:k]
            setattr(out, f, v)
        return out


def _native_raw_batches(lib, h, want_ref, batch, once=False, limit=None, light=False):
    """`light`: flags, names and pi tags only - the blobs nobody reads when records are merely counted (record bytes, bases,
    move tables, reference bases, CIGARs: 16 KB a record) stay where they are."""
    bb = L.BamBatch()
    arr = lambda ptr, dt, count: (np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(dt)), shape=(count,)).copy()
                                  if count else np.zeros(0, dt))  # noqa: E731
    blob = lambda ptr, total: ctypes.string_at(ptr, total) if total else b""  # noqa: E731
    left = None if limit is None else int(limit)
    while True:
        if left is not None:
            if left <= 0:
                return
            batch = min(batch, left)
        L.check(lib.rmr_bam_read_batch(h, batch, 2 if light else int(bool(want_ref)), ctypes.byref(bb)))
        n = int(bb.n_records)
        if n == 0:
            return
        if left is not None:
            left -= n
        rb = RawBamBatch()
        rb.n, rb.want_ref = n, bool(want_ref)
        for f in ("flag", "ref_id", "pos", "mapq", "n_cigar", "ts", "ns", "sp"):
            setattr(rb, f, arr(getattr(bb, f), ctypes.c_int32, n))
        rb.sm, rb.sd = arr(bb.sm, ctypes.c_float, n), arr(bb.sd, ctypes.c_float, n)
        rb.has, rb.ref_ok = arr(bb.has, ctypes.c_uint8, n), arr(bb.ref_ok, ctypes.c_uint8, n)
        rb.tags_off, rb.voffset = arr(bb.tags_off, ctypes.c_int64, n), arr(bb.voffset, ctypes.c_int64, n)
        for f in ("raw_off", "name_off", "seq_off", "mv_off", "pi_off", "refseq_off", "cigar_off"):
            setattr(rb, f, arr(getattr(bb, f), ctypes.c_int64, n + 1))
        rb.names, rb.pi = blob(bb.names, int(rb.name_off[n])), blob(bb.pi, int(rb.pi_off[n]))
        if light:
            rb.cigar, rb.mv, rb.raw, rb.seq, rb.refseq = np.zeros(0, np.uint32), np.zeros(0, np.int8), b"", b"", b""
        else:
            rb.cigar = arr(bb.cigar, ctypes.c_uint32, int(rb.cigar_off[n]))
            rb.raw, rb.seq = blob(bb.raw, int(rb.raw_off[n])), blob(bb.seq, int(rb.seq_off[n]))
            rb.refseq = blob(bb.refseq, int(rb.refseq_off[n]))
            rb.mv = arr(bb.mv, ctypes.c_int8, int(rb.mv_off[n]))
        yield rb
        if n < batch or once:
            return


def _records_of(rb, lib, h, refs):
    """The _NativeBamRecord objects of a raw batch."""
    n = rb.n
    i32 = lambda f: getattr(rb, f).tolist()  # noqa: E731
    flag, ref_id, pos, mapq, n_cig = i32("flag"), i32("ref_id"), i32("pos"), i32("mapq"), i32("n_cigar")
    ts, ns, sp = i32("ts"), i32("ns"), i32("sp")
    sm, sd = rb.sm.tolist(), rb.sd.tolist()
    has, ref_ok = rb.has.tolist(), rb.ref_ok.tolist()
    raw_off, name_off, seq_off, mv_off, pi_off, rs_off, cig_off = (i32(f) for f in (
        "raw_off", "name_off", "seq_off", "mv_off", "pi_off", "refseq_off", "cigar_off"))
    cigar, tags_off, voff = rb.cigar, i32("tags_off"), i32("voffset")
    # latin-1 is one character per byte: offsets stay valid whatever the bytes are (one odd record cannot
    # take the whole batch down with a UnicodeDecodeError)
    raw, names, seq = rb.raw, rb.names.decode("latin-1"), rb.seq.decode("latin-1")
    pi, refseq = rb.pi.decode("latin-1"), rb.refseq.decode("latin-1")
    mv, want_ref = rb.mv, rb.want_ref
    for i in range(n):
        h_i = has[i]
        hot = {}
        if h_i & 1:
            hot["mv"] = mv[mv_off[i] : mv_off[i + 1]]
        if h_i & 2:
            hot["ts"] = ts[i]
        if h_i & 4:
            hot["ns"] = ns[i]
        if h_i & 8:
            hot["sp"] = sp[i]
        if h_i & 16:
            hot["sm"] = sm[i]
        if h_i & 32:
            hot["sd"] = sd[i]
        if h_i & 64:
            hot["pi"] = pi[pi_off[i] : pi_off[i + 1]]
        rid = ref_id[i]
        if rid >= 0 and rid not in refs and lib is not None:
            nm = lib.rmr_bam_ref_name(h, rid)
            refs[rid] = nm.decode() if nm is not None else None
        yield _NativeBamRecord(names[name_off[i] : name_off[i + 1]], flag[i], rid, refs.get(rid) if rid >= 0 else None,
                               pos[i], mapq[i], seq[seq_off[i] : seq_off[i + 1]], raw[raw_off[i] : raw_off[i + 1]],
                               tags_off[i], n_cig[i], hot,
                               refseq[rs_off[i] : rs_off[i + 1]] if (want_ref and ref_ok[i]) else None, voff[i],
                               cigar[cig_off[i] : cig_off[i + 1]])


def _native_batches(lib, h, want_ref, batch, once=False, limit=None):
    refs = {}
    for rb in _native_raw_batches(lib, h, want_ref, batch, once=once, limit=limit):
        yield from _records_of(rb, lib, h, refs)


def read_is_primary(read):
    """Not secondary and not supplementary (src/remora/io.py:147-154)."""
    return not (read.is_supplementary or read.is_secondary)


def get_parent_id(bam_read):
    """The `pi` tag of a split read's child record, else the record's own name (src/remora/io.py:176-182)."""
    hot = bam_read.hot_tags() if hasattr(bam_read, "hot_tags") else dict(bam_read.tags)
    return hot.get("pi", bam_read.query_name)


class ReadIndexedBam:
    """BAM file indexed by (parent) read id, the reference's ReadIndexedBam (src/remora/io.py:184-358) without
    pysam: one streaming pass of the native reader records the BGZF virtual offset of every kept record,
    `get_alignments(read_id)` seeks to them.  Same constructor arguments and attributes: num_records, num_reads,
    read_ids, skip_reasons, `in`, [] (-> offsets), get_alignments, get_first_alignment, iteration over all records."""

    def __init__(self, bam_path, skip_non_primary=True, req_tags=None, read_id_converter=None, parent_read_id_subset=None,
                 child_read_id_subset=None):
        self.bam_path, self.skip_non_primary, self.req_tags = bam_path, skip_non_primary, req_tags
        self.read_id_converter = read_id_converter
        self.parent_read_id_subset, self.child_read_id_subset = parent_read_id_subset, child_read_id_subset
        self.num_reads = self.num_records = None
        self._bam_idx = None
        self._handle = None
        self._handle_pid = None
        self._lock = threading.Lock()
        self.compute_read_index()

    reference_filename = property(lambda s: s.bam_path)
    filename = property(lambda s: s.bam_path)

    def compute_read_index(self):
        from collections import defaultdict

        idx = defaultdict(list)
        self.num_records = 0
        self.skip_reasons = defaultdict(int)
        for rec in iter_bam_records(self.bam_path):
            if self.child_read_id_subset is not None and rec.query_name not in self.child_read_id_subset:
                self.skip_reasons["Child read ID filtered"] += 1
                continue
            rid = get_parent_id(rec)
            if self.parent_read_id_subset is not None and rid not in self.parent_read_id_subset:
                self.skip_reasons["Parent read ID filtered"] += 1
                continue
            if self.read_id_converter is not None:
                rid = self.read_id_converter(rid)
            if self.req_tags is not None and set(self.req_tags).difference(k for k, _ in rec.tags):
                self.skip_reasons["Missing BAM tags"] += 1
                continue
            if self.skip_non_primary and not read_is_primary(rec):
                self.skip_reasons["Non-primary alignment"] += 1
                continue
            self.num_records += 1
            idx[rid].append(rec.voffset)
        self._bam_idx = dict(idx)
        self.num_reads = len(self._bam_idx)

    def get_alignments(self, read_id, want_ref=True):
        """Generator over the alignments of `read_id` (src/remora/io.py:303-325): errors surface on the first `next`,
        as with the reference's generator."""
        if self._bam_idx is None:
            raise RemoraError("Bam index not yet computed")
        try:
            offsets = self._bam_idx[read_id]
        except KeyError:
            raise RemoraError(f"Could not find {read_id} in {self.bam_path}")
        # one native handle per process for the lifetime of the index (the reference keeps its pysam handle open the
        # same way): a seek + one record per offset instead of a header parse and an inflate pool per look-up.  A
        # handle inherited through fork() shares its file offset with the parent: a child re-opens on first use.
        lib = L.lib()
        for vo in offsets:
            with self._lock:  # one record per lock hold: other threads' look-ups interleave between the records
                if self._handle is None or self._handle_pid != os.getpid():
                    h = ctypes.c_void_p()
                    L.check(lib.rmr_bam_open(str(self.bam_path).encode(), ctypes.byref(h)))
                    self._handle, self._handle_pid = h, os.getpid()
                L.check(lib.rmr_bam_seek(self._handle, int(vo)))
                recs = list(_native_batches(lib, self._handle, want_ref, 1, once=True))
            yield from recs

    def get_first_alignment(self, read_id):
        return next(self.get_alignments(read_id))

    def close(self):
        with self._lock:
            if self._handle is not None:
                if self._handle_pid == os.getpid():  # a forked child does not close (or free) the parent's handle
                    L.lib().rmr_bam_close(self._handle)
                self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def __getstate__(self):  # handles do not travel to worker processes; each re-opens on first use
        d = dict(self.__dict__)
        d["_handle"] = None
        d.pop("_lock", None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._lock = threading.Lock()

    def __contains__(self, read_id):
        return read_id in self._bam_idx

    def __getitem__(self, read_id):
        return self._bam_idx[read_id]

    @property
    def read_ids(self):
        return list(self._bam_idx.keys())

    def __iter__(self):
        return iter_bam_records(self.bam_path)


def get_read_ids(bam_idx, pod5_file, num_reads, return_num_bam_reads=False):
    """Read ids present in both the BAM index and the POD5 file, and how many to process (src/remora/io.py:362-391):
    the number of parent reads, or - `return_num_bam_reads` - of BAM records under those parents, capped at num_reads."""
    both = list(set(pod5_file.read_ids).intersection(bam_idx.read_ids))
    count = sum(len(bam_idx[r]) for r in both) if return_num_bam_reads else len(both)
    return both, count if num_reads is None else min(num_reads, count)


def extract_alignments(read_err, bam_idx, rev_sig=False, pa_scaling=None):
    """Every alignment of a signal read as its own io.Read (src/remora/io.py:489-511): [(read, error | None)]."""
    io_read, err = read_err
    if io_read is None:
        return [read_err]
    out = []
    try:
        for rec in bam_idx.get_alignments(io_read.read_id):
            aligned = io_read.copy()
            try:
                aligned.add_alignment(rec, reverse_signal=rev_sig, pa_scaling=pa_scaling)
                out.append((aligned, None))
            except RemoraError as e:
                out.append((aligned, str(e)))
    except RemoraError as e:
        return [(io_read, str(e))]
    return out


def iter_bam_records(bam_path, want_ref=False, batch=512, native=True, shard=None):
    """Yield a BamRecord for every alignment of a BAM file, streaming.  By default the records come from the native
    reader (rmr_bam_read_batch: BGZF inflate, record split, hot tags and - with want_ref - the MD reconstruction in
    C++, `batch` records per call); native=False is the pure-Python reader the native one is tested against.
    `shard=(rank, world)`: only that rank's contiguous share of the records (`shard_of`: by byte range, or by record
    count with REMORA_AMD_BAM_SHARD=scan); or a future of such a result, started earlier."""
    if shard is not None and (hasattr(shard, "result") or int(shard[1]) > 1):
        if not native:
            raise RemoraError("sharded reading needs the native BAM reader")
        # (rank, world), or a future of bam_shard's result started earlier (the scan of a large file takes seconds: a
        # caller overlaps it with its own start-up, e.g. the model load of `infer --gpus N`)
        got = shard.result() if hasattr(shard, "result") else shard_of(bam_path, int(shard[0]), int(shard[1]))
        if len(got) == 3:  # ("bytes", start, end): a share by byte range (bam_byte_shard)
            _, start, end = got
            if start is not None:
                yield from _iter_bam_records_native(bam_path, want_ref, batch, start_voffset=start, end_voffset=end)
            return
        start, count = got
        if count is None:  # a single worker: the whole file
            yield from _iter_bam_records_native(bam_path, want_ref, batch)
            return
        if count:
            yield from _iter_bam_records_native(bam_path, want_ref, batch, start_voffset=start, max_records=count)
        return
    if native:
        yield from _iter_bam_records_native(bam_path, want_ref, batch)
        return
    yield from _iter_bam_records_py(bam_path)


def iter_bam_raw_batches(bam_path, want_ref=False, batch=512, shard=None, light=False, inflate_threads=0):
    """The alignments of a BAM file (or of a rank's share of it, `shard` as in iter_bam_records) as (RawBamBatch, records)
    pairs - `records(rb)` builds the record objects of a batch when somebody needs them - straight from the native reader:
    no Python object per record.  The batch form of iter_bam_records (same shares, same boundary check).
    `inflate_threads`: BGZF inflate workers of this reader (0 = its default)."""
    start = count = end = None
    if shard is not None and (hasattr(shard, "result") or int(shard[1]) > 1):
        got = shard.result() if hasattr(shard, "result") else shard_of(bam_path, int(shard[0]), int(shard[1]))
        if len(got) == 3:  # ("bytes", start, end)
            _, start, end = got
            if start is None:
                return
        else:
            start, count = got
            if count is not None and not count:
                return
    lib = L.lib()
    h = ctypes.c_void_p()
    L.check(lib.rmr_bam_open_threads(str(bam_path).encode(), int(inflate_threads), ctypes.byref(h)))
    refs = {}

    def named(rb):  # reference names are looked up while the file is open: `records` stays usable after the iteration ends
        for rid in np.unique(rb.ref_id).tolist():
            if rid >= 0 and rid not in refs:
                nm = lib.rmr_bam_ref_name(h, rid)
                refs[rid] = nm.decode() if nm is not None else None
        return rb

    records = lambda rb: list(_records_of(rb, None, None, refs))  # noqa: E731
    try:
        if start is not None:
            L.check(lib.rmr_bam_seek(h, int(start)))
        if end is None:
            for rb in _native_raw_batches(lib, h, want_ref, batch, limit=count, light=light):
                yield named(rb), records
            return
        end = int(end)
        for rb in _native_raw_batches(lib, h, want_ref, batch, light=light):
            named(rb)
            past = np.nonzero(rb.voffset >= end)[0]
            if past.size:
                k = int(past[0])
                if int(rb.voffset[k]) != end:
                    raise RemoraError(f"{bam_path}: the records of this share run past virtual offset {end} without one "
                                      f"starting there - the next share's start was guessed wrong (REMORA_AMD_BAM_SHARD=scan "
                                      f"splits by an exact pass over the file instead)")
                if k:
                    yield rb.head(k), records
                return
            yield rb, records
        raise RemoraError(f"{bam_path}: end of file before the record at virtual offset {end} where the next share begins")
    finally:
        lib.rmr_bam_close(h)


def _iter_bam_records_py(bam_path):
    """Pure-Python BGZF/BAM reader (gzip + struct), one record in memory at a time."""
    with gzip.open(bam_path, "rb") as fh:  # BGZF members are valid concatenated gzip members
        if fh.read(4) != b"BAM\x01":
            raise RemoraError(f"{bam_path} is not a BAM file")
        l_text = struct.unpack("<i", _read_exact(fh, 4))[0]
        _read_exact(fh, l_text)
        n_ref = struct.unpack("<i", _read_exact(fh, 4))[0]
        refs = []
        for _ in range(n_ref):
            l_name = struct.unpack("<i", _read_exact(fh, 4))[0]
            refs.append(_read_exact(fh, l_name)[:-1].decode())
            _read_exact(fh, 4)
        while True:
            head = fh.read(4)
            if not head:
                return
            if len(head) != 4:
                raise RemoraError("truncated BAM file")
            rec = _read_exact(fh, struct.unpack("<i", head)[0])
            ref_id, pos, l_read_name, mapq, _bin, n_cig, flag, l_seq, _nref, _npos, _tlen = struct.unpack_from(
                "<iiBBHHHiiii", rec, 0)
            q = 32
            name = rec[q : q + l_read_name - 1].decode(); q += l_read_name
            cig = np.frombuffer(rec, dtype="<u4", count=n_cig, offset=q); q += 4 * n_cig
            sb = np.frombuffer(rec, dtype=np.uint8, count=(l_seq + 1) // 2, offset=q); q += (l_seq + 1) // 2
            codes = np.empty(2 * sb.size, np.uint8)
            codes[0::2], codes[1::2] = sb >> 4, sb & 0xF
            seq = _NT16[codes[:l_seq]].tobytes().decode()
            qual = bytes(rec[q : q + l_seq]); q += l_seq
            spans = []
            tags = _parse_tags(rec[q:], spans)
            stored = cig
            cig = _resolve_long_cigar(cig, l_seq, lambda: tags) if ref_id >= 0 and pos >= 0 else cig
            if cig is not stored:  # resolved from CG: the tag disappears from the exposed list (htslib deletes it)
                tags = [(k, v) for k, v in tags if k != "CG"]
            cigartuples = list(zip((cig & 0xF).tolist(), (cig >> 4).tolist()))
            yield BamRecord(name, flag, ref_id, refs[ref_id] if ref_id >= 0 else None, pos, mapq, cigartuples, seq,
                            qual, tags, bytes(rec), q, spans)


_ZSTD = None


def _libzstd():
    """The system libzstd through ctypes: the one-shot API is re-entrant, so signal rows can be inflated from the
    ingest thread (pyarrow's CompressedInputStream crashes when it is used from short-lived threads)."""
    global _ZSTD
    if _ZSTD is None:
        import ctypes.util

        name = ctypes.util.find_library("zstd") or "libzstd.so.1"
        try:
            lib = ctypes.CDLL(name)
            lib.ZSTD_getFrameContentSize.restype = ctypes.c_ulonglong
            lib.ZSTD_getFrameContentSize.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
            lib.ZSTD_decompress.restype = ctypes.c_size_t
            lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
            lib.ZSTD_isError.restype = ctypes.c_uint
            lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
            _ZSTD = lib
        except OSError:
            _ZSTD = False
    return _ZSTD


def _zstd_decompress(blob):
    """One zstd frame -> bytes."""
    lib = _libzstd()
    if lib:
        blob = bytes(blob)
        size = lib.ZSTD_getFrameContentSize(blob, len(blob))
        if size < (1 << 62):  # not CONTENTSIZE_UNKNOWN / _ERROR
            out = ctypes.create_string_buffer(max(int(size), 1))
            got = lib.ZSTD_decompress(out, int(size), blob, len(blob))
            if lib.ZSTD_isError(got) or got != size:
                raise RemoraError("corrupt zstd frame in POD5 signal row")
            return out.raw[: int(size)]
    import pyarrow as pa

    return pa.CompressedInputStream(pa.BufferReader(blob), "zstd").read()


def vbz_decode_rows(addr, size, n_samples, engine=None):
    """vbz_decode_batch for rows given by address and compressed size (Pod5File.rows_of_reads: pointers into the mapped
    file, no bytes object per row); the samples stay on the GPU.  Returns (int16 CUDA tensor, row offsets int64[n + 1])."""
    import torch

    from .engine import get_engine

    eng = engine if engine is not None else get_engine()
    n = int(np.asarray(size).size)
    src = np.ascontiguousarray(addr, np.uint64)
    src_len = np.ascontiguousarray(size, np.int64)
    out_off = np.zeros(n + 1, np.int64)
    np.cumsum(np.asarray(n_samples, np.int64), out=out_off[1:])
    dev = eng.torch_device
    if n == 0:
        return torch.zeros(0, dtype=torch.int16, device=dev), out_off
    sizes = np.zeros(n, np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    L.check(L.lib().rmr_zstd_frame_sizes(p(src), p(src_len), n, p(sizes)))
    row_off = np.zeros(n + 1, np.int64)
    np.cumsum(sizes, out=row_off[1:])
    total = int(row_off[-1])
    # the zstd layer inflates straight into pinned memory: one copy to the device, no pageable bounce
    host = torch.empty(total + 16, dtype=torch.uint8, pin_memory=True)
    host[total:] = 0
    L.check(L.lib().rmr_zstd_rows(p(src), p(src_len), n, ctypes.c_void_p(host.data_ptr()), p(row_off), min(8, _eff_cpus())))
    rn = np.ascontiguousarray(n_samples, np.int32)
    d_buf = host.to(dev, non_blocking=True)
    d_ro, d_rn = torch.from_numpy(row_off).to(dev), torch.from_numpy(rn).to(dev)
    d_out = torch.empty(max(int(out_off[-1]), 1), dtype=torch.int16, device=dev)
    torch.cuda.current_stream(dev).synchronize()  # the copies ran on torch's stream, the kernel runs on the engine's
    L.check(L.lib().rmr_vbz_decode(eng.handle, d_buf.data_ptr(), d_ro.data_ptr(), d_rn.data_ptr(), n, d_out.data_ptr(), L.MEM_DEVICE))
    eng.synchronize()
    return d_out[: int(out_off[-1])], out_off


def vbz_decode_batch(blobs, n_samples, engine=None, to_host=True):
    """Decode many VBZ signal rows at once: zstd on the host (libzstd through pyarrow), then streamvbyte16 ->
    zigzag -> running sum on the GPU (`rmr_vbz_decode`; one upload of 1..2 bytes per sample).  Returns the rows'
    int16 samples back to back (numpy array, or a CUDA tensor with to_host=False) and the row offsets."""
    import ctypes

    import torch

    from . import _lib as L
    from .engine import get_engine

    eng = engine if engine is not None else get_engine()
    # the zstd layer of all rows, inflated by native threads straight into the buffer the kernel uploads
    n = len(blobs)
    blobs = [b if isinstance(b, bytes) else bytes(b) for b in blobs]
    src = (ctypes.c_char_p * max(n, 1))(*blobs)
    src_len = np.asarray([len(b) for b in blobs], np.int64)
    sizes = np.zeros(max(n, 1), np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    L.check(L.lib().rmr_zstd_frame_sizes(src, p(src_len), n, p(sizes)))
    row_off = np.zeros(n + 1, np.int64)
    np.cumsum(sizes[:n], out=row_off[1:])
    out_off = np.zeros(n + 1, np.int64)
    np.cumsum(np.asarray(n_samples, np.int64), out=out_off[1:])
    buf = np.zeros(int(row_off[-1]) + 16, np.uint8)  # the kernel reads whole dwords: a little slack at the end
    L.check(L.lib().rmr_zstd_rows(src, p(src_len), n, p(buf), p(row_off), min(8, _eff_cpus())))
    rn = np.ascontiguousarray(n_samples, np.int32)
    if to_host:
        out = np.empty(int(out_off[-1]), np.int16)
        L.check(L.lib().rmr_vbz_decode(eng.handle, p(buf), p(row_off), p(rn), n, p(out), L.MEM_HOST))
        return out, out_off
    dev = eng.torch_device
    d_buf, d_ro, d_rn = (torch.from_numpy(a).to(dev) for a in (buf, row_off, rn))
    d_out = torch.empty(max(int(out_off[-1]), 1), dtype=torch.int16, device=dev)
    L.check(L.lib().rmr_vbz_decode(eng.handle, d_buf.data_ptr(), d_ro.data_ptr(), d_rn.data_ptr(), n, d_out.data_ptr(),
                                   L.MEM_DEVICE))
    eng.synchronize()
    return d_out[: int(out_off[-1])], out_off


@dataclasses.dataclass
class Pod5Read:
    read_id: str
    signal: np.ndarray       # int16 DAC samples
    calibration_offset: float
    calibration_scale: float


def _pod5_embedded_files(buf):
    """[(offset, length)] of the Arrow files embedded in a POD5 container, from its footer: the file ends with
    b"FOOTER\0\0" + flatbuffer(Footer{..., contents:[EmbeddedFile{offset, length, format, content_type}]}) + padding +
    int64 footer length + 16-byte section marker + 8-byte signature."""
    n = len(buf)
    if n < 64 or bytes(buf[n - 8 : n]) != b"\x8bPOD\r\n\x1a\n":
        raise RemoraError("no POD5 signature at the end of the file")
    flen = struct.unpack_from("<q", buf, n - 32)[0]
    fb = n - 32 - flen
    if flen <= 0 or fb < 16 or bytes(buf[fb - 8 : fb]) != b"FOOTER\x00\x00":
        raise RemoraError("POD5 footer not found")
    u32 = lambda p: struct.unpack_from("<I", buf, p)[0]  # noqa: E731

    def field(table, k):  # flatbuffers: table -> vtable -> offset of field k (0 = absent)
        vt = table - struct.unpack_from("<i", buf, table)[0]
        if 4 + 2 * k >= struct.unpack_from("<H", buf, vt)[0]:
            return None
        off = struct.unpack_from("<H", buf, vt + 4 + 2 * k)[0]
        return table + off if off else None

    root = fb + u32(fb)
    cp = field(root, 3)
    if cp is None:
        raise RemoraError("POD5 footer without contents")
    vec = cp + u32(cp)
    files = []
    for k in range(u32(vec)):
        e = vec + 4 + 4 * k
        tab = e + u32(e)
        po, pl = field(tab, 0), field(tab, 1)
        off = struct.unpack_from("<q", buf, po)[0] if po else 0
        length = struct.unpack_from("<q", buf, pl)[0] if pl else 0
        if off < 0 or length <= 0 or off + length > n:
            raise RemoraError("POD5 footer entry out of range")
        files.append((off, length))
    return files


class Pod5File:
    """Random access to the reads of a POD5 file without the pod5 package: the file is memory mapped, the
    embedded Arrow IPC tables (signal rows, reads) are opened in place, a read's signal rows are VBZ-decoded
    (zstd on the host, the streamvbyte / zigzag / running-sum layer on the GPU) only when the read is asked for."""

    def __init__(self, pod5_path):
        import mmap
        import uuid

        import pyarrow as pa
        import pyarrow.ipc as ipc

        self._fh = open(pod5_path, "rb")
        self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        mm = self._mm
        if mm[:8] != b"\x8bPOD\r\n\x1a\n":
            raise RemoraError(f"{pod5_path} is not a POD5 file")
        view = memoryview(mm)
        tables = {}

        def classify(t):
            names = set(t.schema.names)
            if {"signal", "samples"} <= names:
                tables["signal"] = t
            elif "calibration_offset" in names:
                tables["reads"] = t

        try:  # the footer lists the embedded Arrow files: O(1) whatever the file size
            for off, length in _pod5_embedded_files(mm):
                classify(ipc.open_file(pa.BufferReader(pa.py_buffer(view[off : off + length]))).read_all())
        except (RemoraError, pa.ArrowInvalid, OSError, struct.error, IndexError, ValueError):
            tables = {}
        if "signal" not in tables or "reads" not in tables:  # damaged / missing footer: look for the Arrow magic
            tables = {}
            marks = []
            i = mm.find(b"ARROW1")
            while i >= 0:
                marks.append(i)
                i = mm.find(b"ARROW1", i + 1)
            k = 0
            while k + 1 < len(marks):  # embedded files are [ARROW1\0\0 ... ARROW1] pairs
                st = marks[k]
                opened = False
                for e in marks[k + 1 :]:
                    try:
                        t = ipc.open_file(pa.BufferReader(pa.py_buffer(view[st : e + 6]))).read_all()
                    except (pa.ArrowInvalid, OSError):
                        continue
                    classify(t)
                    k = marks.index(e) + 1
                    opened = True
                    break
                if not opened:
                    k += 1
        if "signal" not in tables or "reads" not in tables:
            raise RemoraError(f"could not locate the signal / reads tables in {pod5_path}")
        self._sig, self._reads = tables["signal"], tables["reads"]
        self.read_ids = [str(uuid.UUID(bytes=b)) for b in self._reads.column("read_id").to_pylist()]
        self._row = {rid: r for r, rid in enumerate(self.read_ids)}
        # per-read columns once (small), per-row access into the big signal table by (chunk, index): indexing a
        # chunked column by a global row number walks its chunks every time
        self._cal_off = self._reads.column("calibration_offset").to_numpy()
        self._cal_scale = self._reads.column("calibration_scale").to_numpy()
        self._read_rows = self._reads.column("signal")
        self._read_rows_starts = np.cumsum([0] + [len(c) for c in self._read_rows.chunks])
        sig_col, n_col = self._sig.column("signal"), self._sig.column("samples")
        self._sig_chunks, self._n_chunks = sig_col.chunks, n_col.chunks
        self._sig_starts = np.cumsum([0] + [len(c) for c in self._sig_chunks])
        self._n_starts = np.cumsum([0] + [len(c) for c in self._n_chunks])

    def _row_index(self):
        """Flat numpy views of the two tables, built on first use: per signal row its address in the mapped file, its
        compressed size and its sample count; per read the range of its rows.  With them a batch of reads is located
        by array arithmetic instead of one Arrow scalar (`.as_py()`, a bytes copy of the row) per cell."""
        idx = getattr(self, "_ridx", None)
        if idx is None:
            addr, size = [], []
            for c in self._sig_chunks:
                _, offs, data = c.buffers()
                wide = np.int64 if str(c.type).startswith("large") else np.int32
                o = np.frombuffer(offs, dtype=wide, count=c.offset + len(c) + 1)[c.offset :].astype(np.int64)
                addr.append(np.uint64(data.address) + o[:-1].astype(np.uint64))
                size.append(np.diff(o))
            samples = np.concatenate([c.to_numpy(zero_copy_only=False) for c in self._n_chunks]).astype(np.int64) if self._n_chunks else np.zeros(0, np.int64)
            rr = self._read_rows.combine_chunks()
            r_off = rr.offsets.to_numpy().astype(np.int64)
            r_val = rr.values.to_numpy().astype(np.int64)
            idx = self._ridx = (np.concatenate(addr) if addr else np.zeros(0, np.uint64), np.concatenate(size) if size else np.zeros(0, np.int64),
                                samples, r_off - r_off[0], r_val[r_off[0] :] if r_off.size else r_val)
        return idx

    def rows_of_reads(self, read_rows):
        """For reads given by their row numbers in the reads table: (row_first int64[n + 1] - read k's signal rows are
        entries row_first[k] .. row_first[k + 1] of the next three arrays -, address uint64[], compressed size int64[],
        samples int64[])."""
        addr, size, samples, r_off, r_val = self._row_index()
        read_rows = np.asarray(read_rows, np.int64)
        cnt = r_off[read_rows + 1] - r_off[read_rows]
        first = np.zeros(read_rows.size + 1, np.int64)
        np.cumsum(cnt, out=first[1:])
        # entry j of read k is r_val[r_off[read k] + j]
        flat = np.repeat(r_off[read_rows] - first[:-1], cnt) + np.arange(int(first[-1]), dtype=np.int64)
        rows = r_val[flat]
        return first, addr[rows], size[rows], samples[rows]

    @staticmethod
    def _cell(chunks, starts, i):
        k = int(np.searchsorted(starts, i, side="right")) - 1
        return chunks[k][int(i - starts[k])].as_py()

    def _rows_of(self, read_id):
        return self._cell(self._read_rows.chunks, self._read_rows_starts, self._row[read_id])

    def __contains__(self, read_id):
        return read_id in self._row

    def __len__(self):
        return len(self.read_ids)

    def signal_rows(self, read_id):
        """[(zstd-compressed VBZ bytes, number of samples)] of a read's signal rows, in order (table access only)."""
        return [(self._cell(self._sig_chunks, self._sig_starts, i), self._cell(self._n_chunks, self._n_starts, i))
                for i in self._rows_of(read_id)]

    def calibration(self, read_id):
        r = self._row[read_id]
        return float(self._cal_off[r]), float(self._cal_scale[r])

    def get(self, read_id, engine=None):
        """One read (signal decoded on the GPU like a batch of one)."""
        return self.get_many([read_id], engine)[0]

    def get_many(self, read_ids, engine=None):
        """The signals of several reads decoded in one GPU call (see vbz_decode_batch); list of Pod5Read."""
        rows_of, blobs, ns = [], [], []
        for rid in read_ids:
            rows = self.signal_rows(rid)
            rows_of.append(len(rows))
            for blob, n in rows:
                blobs.append(blob)
                ns.append(n)
        flat, off = vbz_decode_batch(blobs, ns, engine)
        out, k = [], 0
        for rid, nrows in zip(read_ids, rows_of):
            sig = flat[off[k] : off[k + nrows]]  # a read's rows are consecutive in the batch
            k += nrows
            out.append(Pod5Read(rid, sig, *self.calibration(rid)))
        return out

    def __iter__(self):
        for rid in self.read_ids:
            yield self.get(rid)


def iter_pod5_reads(pod5_path, read_ids=None):
    """Yield Pod5Read for every (requested) read of a POD5 file."""
    f = Pod5File(pod5_path)
    want = None if read_ids is None else set(read_ids)
    for rid in f.read_ids:
        if want is None or rid in want:
            yield f.get(rid)


_COMP = str.maketrans("ACGTBVDHKMRYacgt", "TGCAVBHDMKYRtgca")


def revcomp(seq):
    return seq.translate(_COMP)[::-1]


PA_TO_NORM_SCALING_FACTOR = 1.4826


@dataclasses.dataclass
class RefRegion:
    """Contig, strand and 0-based half-open span of an alignment (src/remora/io.py:86-101)."""

    ctg: str
    strand: str
    start: int
    end: int = None

    @property
    def len(self):
        return self.end - self.start

    @property
    def coord_range(self):
        return range(self.start, self.end)


def parse_bed_lines(bed_path):
    """One RefRegion per BED line; strand None unless column 6 is + or - (src/remora/io.py:105-115)."""
    with open(bed_path) as fh:
        for line in fh:
            fields = line.split()
            strand = fields[5] if len(fields) >= 6 and fields[5] in "+-" else None
            yield RefRegion(fields[0], strand, int(fields[1]), int(fields[2]))


def parse_bed(bed_path):
    """{(contig, strand): set of 0-based positions}; unstranded lines count for both strands (:118-126)."""
    from collections import defaultdict

    regs = defaultdict(set)
    for reg in parse_bed_lines(bed_path):
        for strand in ("+-" if reg.strand is None else reg.strand):
            regs[(reg.ctg, strand)].update(reg.coord_range)
    return regs


def _ref_to_signal_of_bam_cigar(cig, is_reverse, query_to_signal, expect):
    """compute_ref_to_signal for a CIGAR in BAM's uint32 form: one native walk (rmr_ref_to_signal, np.interp's arithmetic: the
    same integers) instead of two interpolations over arrays of the read's length.  `expect`: the usual size of the result
    (reference sequence + 1), the first buffer tried."""
    from .data_chunks import compute_ref_to_signal

    cig = np.ascontiguousarray(cig, dtype=np.uint32)
    q2s = np.ascontiguousarray(query_to_signal, dtype=np.int64)
    n = ctypes.c_int64(0)
    cap = int(expect)
    for _ in range(2):
        out = np.empty(max(cap, 1), np.int64)
        rc = L.lib().rmr_ref_to_signal(cig.ctypes.data, cig.size, int(bool(is_reverse)), q2s.ctypes.data, q2s.size, out.ctypes.data,
                                       out.size, ctypes.byref(n))
        if rc == 0:
            return out[: n.value]
        if n.value <= out.size:  # not a matter of room
            break
        cap = int(n.value)
    msg = (L.lib().rmr_last_error() or b"").decode(errors="replace")
    if msg.startswith("cigar with an empty match run"):  # never in a valid BAM; the array form does what numpy does with it
        ops, lens = (cig & 0xF).astype(np.int64), (cig >> 4).astype(np.int64)
        return compute_ref_to_signal(query_to_signal=query_to_signal, cigar=(ops[::-1], lens[::-1]) if is_reverse else (ops, lens))
    raise RemoraError(msg or "rmr_ref_to_signal failed")


@dataclasses.dataclass
class Read:
    """Signal + basecalls + their mapping for one read: the subset of remora.io.Read
    (src/remora/io.py:1747-2177) that the basecall-anchored inference path uses."""

    read_id: str
    dacs: np.ndarray = None
    seq: str = None
    stride: int = None
    mv_table: np.ndarray = None
    query_to_signal: np.ndarray = None
    shift_dacs_to_pa: float = None
    scale_dacs_to_pa: float = None
    shift_pa_to_norm: float = None
    scale_pa_to_norm: float = None
    shift_dacs_to_norm: float = None
    scale_dacs_to_norm: float = None
    shift_pa_to_zc_pa: float = None
    scale_pa_to_zc_pa: float = None
    full_align: dict = None
    _child_read_id: str = None
    record: object = None  # the BamRecord this read was aligned with (for output)
    ref_seq: str = None
    ref_reg: object = None
    cigar: list = None
    ref_to_signal: np.ndarray = None

    @property
    def child_read_id(self):
        return self.read_id if self._child_read_id is None else self._child_read_id

    @property
    def sig_len(self):
        return None if self.dacs is None else self.dacs.size

    @property
    def pa_signal(self):
        return (self.dacs - self.shift_dacs_to_pa) / self.scale_dacs_to_pa

    def compute_pa_to_norm_scaling(self, factor=PA_TO_NORM_SCALING_FACTOR):
        """src/remora/io.py:1851-1856 (median / MAD)."""
        pa = self.pa_signal
        self.shift_pa_to_norm = np.median(pa)
        self.scale_pa_to_norm = max(1.0, np.median(np.abs(pa - self.shift_pa_to_norm)) * factor)

    @classmethod
    def from_pod5(cls, pod5_read, reverse_signal=False, infer_convention=True):
        """`infer_convention=True` follows iter_signal (used by `remora infer`:
        shift=offset, scale=scale, src/remora/io.py:466-472); False follows
        from_pod5_and_alignment (shift=-offset, scale=1/scale, :2105-2115)."""
        dacs = pod5_read.signal[::-1] if reverse_signal else pod5_read.signal
        if infer_convention:
            sh, sc = pod5_read.calibration_offset, pod5_read.calibration_scale
        else:
            sh, sc = -pod5_read.calibration_offset, 1 / pod5_read.calibration_scale
        return cls(read_id=pod5_read.read_id, dacs=dacs, shift_dacs_to_pa=sh, scale_dacs_to_pa=sc)

    def add_alignment(self, rec, parse_ref_align=True, reverse_signal=False, pa_scaling=None, parsed_moves=None):
        """Signal trimming by sp/ts/ns, read-id checks, strand-aware sequence, move table ->
        query_to_signal, sm/sd (or median/MAD) norm scaling composed with the pA calibration, and - for
        mapped records when `parse_ref_align` - the reference region, the reference sequence (MD tag),
        the read-oriented CIGAR and the reference-to-signal mapping (src/remora/io.py:1972-2084)."""
        if pa_scaling is not None:
            self.shift_pa_to_zc_pa, self.scale_pa_to_zc_pa = pa_scaling
        if rec.reference_name is None and rec.is_reverse:
            raise RemoraError("Unmapped reads cannot map to reverse strand.")
        if self.dacs is None:
            raise RemoraError("Must add signal to io.Read before alignment.")
        self.full_align = rec.to_dict()
        self.record = rec
        tags = rec.hot_tags() if hasattr(rec, "hot_tags") else dict(rec.tags)
        if reverse_signal:
            self.dacs = self.dacs[::-1]
        self.dacs = self.dacs[tags.get("sp", 0) :]
        self.dacs = self.dacs[tags.get("ts", 0) : tags.get("ns", self.dacs.size)]
        if reverse_signal:
            self.dacs = self.dacs[::-1]
        parent = tags.get("pi", None)
        if parent is None:
            if rec.query_name != self.read_id:
                raise RemoraError("Read IDs mismatch")
        else:
            if parent != self.read_id:
                raise RemoraError("Split read IDs mismatch")
            self._child_read_id = rec.query_name
        self.seq = revcomp(rec.query_sequence) if rec.is_reverse else rec.query_sequence
        if "mv" in tags and parsed_moves is not None:  # expanded together with the rest of the batch (parse_move_tags)
            if isinstance(parsed_moves, Exception):
                raise parsed_moves
            self.query_to_signal, self.mv_table, self.stride = parsed_moves
        elif "mv" in tags:
            self.query_to_signal, self.mv_table, self.stride = parse_move_tag(
                tags["mv"], sig_len=self.sig_len, seq_len=len(self.seq), reverse_signal=reverse_signal)
        else:
            self.query_to_signal = self.mv_table = self.stride = None
        if "sm" in tags and "sd" in tags:
            self.shift_pa_to_norm, self.scale_pa_to_norm = tags["sm"], tags["sd"]
        else:
            self.compute_pa_to_norm_scaling()
        self.shift_dacs_to_norm = self.shift_dacs_to_pa + (self.scale_dacs_to_pa * self.shift_pa_to_norm)
        self.scale_dacs_to_norm = self.scale_dacs_to_pa * self.scale_pa_to_norm
        if not parse_ref_align or rec.is_unmapped:
            return
        from .data_chunks import compute_ref_to_signal

        self.ref_reg = RefRegion(ctg=rec.reference_name, strand="-" if rec.is_reverse else "+", start=rec.reference_start)
        try:
            self.ref_seq = rec.get_reference_sequence().upper()
        except ValueError:
            self.ref_seq = None
        self.cigar = list(rec.cigartuples)
        if rec.is_reverse:
            if self.ref_seq is not None:
                self.ref_seq = revcomp(self.ref_seq)
            self.cigar = self.cigar[::-1]
        if self.ref_reg.ctg is not None and self.ref_seq is not None and self.query_to_signal is not None:
            cig = getattr(rec, "_cigar", None)  # the native reader's uint32 operations: no tuple per operation on the way
            if isinstance(cig, np.ndarray) and cig.size == len(self.cigar):
                self.ref_to_signal = _ref_to_signal_of_bam_cigar(cig, rec.is_reverse, self.query_to_signal, len(self.ref_seq) + 1)
            else:
                self.ref_to_signal = compute_ref_to_signal(query_to_signal=self.query_to_signal, cigar=self.cigar)
            if self.ref_to_signal.size != len(self.ref_seq) + 1:  # knots include the end of the last base
                raise RemoraError("Discordant ref seq lengths")
            self.ref_reg.end = self.ref_reg.start + self.ref_to_signal.size - 1

    def copy(self):
        """Shallow copy with its own attribute set (arrays are shared, add_alignment rebinds them)."""
        return dataclasses.replace(self)

    def get_filtered_focus_positions(self, select_focus_positions):
        """Read-oriented offsets into the reference sequence of this alignment for the positions of
        `select_focus_positions` (as from parse_bed) that the alignment covers (src/remora/io.py:2215-2247)."""
        if self.ref_reg is None or self.ref_seq is None:
            raise RemoraError("Cannot extract focus positions without mapping")
        reg, n = self.ref_reg, len(self.ref_seq)
        wanted = select_focus_positions.get((reg.ctg, reg.strand))
        if wanted is None:
            return np.array([], dtype=int)
        hit = np.array(sorted(set(range(reg.start, reg.start + n)).intersection(wanted)), dtype=int)
        return hit - reg.start if reg.strand == "+" else reg.start + n - hit[::-1] - 1

    def get_basecall_anchored_focus_bases(self, motifs, select_focus_reference_positions):
        """Basecall positions on a motif whose aligned reference position is on a motif too (or in the BED
        selection) (src/remora/io.py:2249-2289)."""
        from . import util
        from .data_chunks import make_sequence_coordinate_mapping

        if self.cigar is None:
            raise RemoraError("missing alignment")
        called = util.find_focus_bases_in_int_sequence(util.seq_to_int(self.seq), motifs)
        ref_to_query = make_sequence_coordinate_mapping(self.cigar).astype(int)
        if select_focus_reference_positions is None:
            ref_pos = util.find_focus_bases_in_int_sequence(util.seq_to_int(self.ref_seq), motifs)
        else:
            ref_pos = self.get_filtered_focus_positions(select_focus_reference_positions)
        supported = set(ref_to_query[ref_pos].tolist())
        return np.array([fb for fb in called if fb in supported])

    def into_remora_read(self, use_reference_anchor=False):
        """RemoraRead anchored on the basecalls (move table) or on the reference (move table composed with the
        alignment), src/remora/io.py:2123-2177."""
        from .data_chunks import RemoraRead, compute_ref_to_signal

        if use_reference_anchor:
            if self.ref_to_signal is None:
                if self.cigar is None or self.ref_seq is None:
                    raise RemoraError("Missing reference alignment")
                self.ref_to_signal = compute_ref_to_signal(self.query_to_signal, self.cigar)
                if self.ref_to_signal.size != len(self.ref_seq) + 1:
                    raise RemoraError("Discordant ref seq lengths")
            trim = self.dacs[self.ref_to_signal[0] : self.ref_to_signal[-1]]
            if self.shift_pa_to_zc_pa is None or self.scale_pa_to_zc_pa is None:
                shift, scale = self.shift_dacs_to_norm, self.scale_dacs_to_norm
            else:
                shift = self.shift_dacs_to_pa + self.scale_dacs_to_pa * self.shift_pa_to_zc_pa
                scale = self.scale_dacs_to_pa * self.scale_pa_to_zc_pa
            rr = RemoraRead(dacs=trim, shift=shift, scale=scale, seq_to_sig_map=self.ref_to_signal - self.ref_to_signal[0],
                            str_seq=self.ref_seq, read_id=self.read_id)
            rr.check()
            return rr
        if self.query_to_signal is None:
            raise RemoraError("Missing query_to_signal (move table)")
        trim = self.dacs[self.query_to_signal[0] : self.query_to_signal[-1]]
        if self.shift_pa_to_zc_pa is None or self.scale_pa_to_zc_pa is None:
            shift, scale = self.shift_dacs_to_norm, self.scale_dacs_to_norm
        else:
            shift = self.shift_dacs_to_pa + self.scale_dacs_to_pa * self.shift_pa_to_zc_pa
            scale = self.scale_dacs_to_pa * self.scale_pa_to_zc_pa
        rr = RemoraRead(dacs=trim, shift=shift, scale=scale,
                        seq_to_sig_map=self.query_to_signal - self.query_to_signal[0], str_seq=self.seq,
                        read_id=self.read_id)
        rr.check()
        return rr


def iter_reads_from_pod5_and_bam(pod5_path, bam_path, reverse_signal=False, pa_scaling=None,
                                 skip_non_primary=True, decode_batch=256, parse_ref_align=True, shard=None, device=None):
    """(io.Read, error-or-None) for every BAM record whose signal is in the POD5 file — the
    read-producing front of infer_from_pod5_and_bam (src/remora/inference.py:477-519).  The signals of
    `decode_batch` consecutive records are decompressed together (zstd on the host, VBZ on the GPU) and their
    move tables are expanded in one launch; decode_batch <= 1 works read by read (one launch per read).  `parse_ref_align=False` skips the
    reference side of the alignment (MD reconstruction, ref_to_signal) when only basecall-anchored reads are needed.
    `device`: the GPU that decodes (the model's): the generator usually runs in a producer THREAD, and a new thread's
    current device is 0 whatever the rank's device is - every rank of a `--gpus N` run would otherwise decode on GPU 0."""
    signals = Pod5File(pod5_path)
    from .engine import get_ingest_engine

    ingest_eng = get_ingest_engine(device) if decode_batch > 1 else None  # own stream: not behind the model's kernels
    yield from _reads_of_records(iter_bam_records(bam_path, want_ref=parse_ref_align, shard=shard), signals, ingest_eng, reverse_signal,
                                 pa_scaling, skip_non_primary, decode_batch, parse_ref_align)


def _reads_of_records(records, signals, ingest_eng, reverse_signal, pa_scaling, skip_non_primary, decode_batch, parse_ref_align):
    """(io.Read, error-or-None) for the records of an iterable whose signal is in `signals` (see iter_reads_from_pod5_and_bam)."""

    def emit(recs):
        if decode_batch > 1 and recs:
            ids = list(dict.fromkeys(rid for _, rid in recs))
            pods = dict(zip(ids, signals.get_many(ids, engine=ingest_eng)))
        else:
            pods = None
        reads = [Read.from_pod5(pods[rid] if pods is not None else signals.get(rid), reverse_signal=reverse_signal)
                 for _rec, rid in recs]
        moves = [None] * len(recs)
        if decode_batch > 1:  # the move tables of the whole batch in one launch
            have, mvs, sls, qls = [], [], [], []
            for k, ((rec, _rid), read) in enumerate(zip(recs, reads)):
                tags = rec.hot_tags() if hasattr(rec, "hot_tags") else dict(rec.tags)
                if "mv" in tags and read.dacs is not None:
                    # signal length after the sp / ts / ns trimming add_alignment applies (same slicing rules)
                    sls.append(len(range(read.dacs.size)[tags.get("sp", 0) :][tags.get("ts", 0) : tags.get("ns", None)]))
                    mvs.append(tags["mv"])
                    qls.append(len(rec.query_sequence))
                    have.append(k)
            for k, res in zip(have, parse_move_tags(mvs, sls, qls, reverse_signal=reverse_signal, engine=ingest_eng)):
                moves[k] = res
        for (rec, rid), read, mv in zip(recs, reads, moves):
            try:
                read.add_alignment(rec, parse_ref_align=parse_ref_align, reverse_signal=reverse_signal,
                                   pa_scaling=pa_scaling, parsed_moves=mv)
            except RemoraError as e:
                read.record = rec  # (add_alignment turns some records away before it stores them: the output still copies them)
                yield read, str(e)
                continue
            yield read, None

    pending = []
    for rec in records:
        if skip_non_primary and (rec.is_secondary or rec.is_supplementary):
            continue
        rid = (rec.hot_tags() if hasattr(rec, "hot_tags") else dict(rec.tags)).get("pi", rec.query_name)
        if rid not in signals:
            continue
        pending.append((rec, rid))
        if len(pending) >= max(decode_batch, 1):
            yield from emit(pending)
            pending = []
    yield from emit(pending)


class ReadStub:
    """What the batched call path reads from and writes back to a read object (inference.call_reads_mods, the refiner's
    device passes) when the read itself only exists as rows of a DeviceReads batch."""

    __slots__ = ("shift", "scale", "seq_to_sig_map", "focus_bases", "_sig", "read_id")

    def __init__(self, shift, scale, read_id=None):
        self.shift, self.scale, self.read_id = shift, scale, read_id
        self.seq_to_sig_map, self.focus_bases, self._sig = _NO_MAP, None, None


_NO_MAP = np.zeros(0, np.int64)
_COMP_BYTES = bytes.maketrans(b"ACGTBVDHKMRYacgt", b"TGCAVBHDMKYRtgca")  # the byte form of _COMP (revcomp): the same letters


class IngestBatch:
    """The kept alignments of one native BAM batch, ready for the GPU without a Python object per read:
    `rb` / `keep` - the raw batch and the indices of its records that are handed on (primary, signal present), in input order;
    `err[k]` - None or the reason record keep[k] cannot be called (the strings Read.add_alignment / into_remora_read raise);
    `good` - positions in `keep` of the callable reads; for them `dr` (DeviceReads assembled on the GPU), `reads`
    (ReadStub per good read), `seq` (their strand-oriented bases, ASCII, back to back) and `seq_off`.  Reference-anchored
    batches: `seq` holds the reference bases of the alignments in read orientation (what the reads are anchored on), and
    `ref_fwd` / `ref_fwd_off` (int64[len(keep) + 1]) the same bases in forward-strand orientation per KEPT record (empty for
    records that cannot be called) - what the output records are rewritten with; None otherwise.  `per_read()` - the same
    records as (io.Read, error) pairs of the per-read path, built on demand."""

    __slots__ = ("rb", "keep", "err", "good", "dr", "reads", "seq", "iseq", "seq_off", "records", "ref_fwd", "ref_fwd_off", "per_read")

    def __len__(self):
        return int(self.keep.size)

    def head(self, k):
        """The first k kept records (a `num_reads` limit that ends inside a batch): the callable ones among them keep their
        rows of `dr`, which is cut as well."""
        if k >= len(self):
            return self
        out = IngestBatch()
        out.rb, out.records, out.keep, out.err = self.rb, self.records, self.keep[:k], self.err[:k]
        out.ref_fwd, out.ref_fwd_off = self.ref_fwd, (None if self.ref_fwd_off is None else self.ref_fwd_off[: k + 1])
        out.per_read = (lambda pr=self.per_read, kk=k: pr()[:kk]) if self.per_read is not None else None
        g = int(np.searchsorted(self.good, k))
        out.good, out.reads, out.seq_off = self.good[:g], self.reads[:g], self.seq_off[: g + 1]
        out.seq = self.seq[: int(self.seq_off[g])]
        out.iseq = self.iseq[: int(self.seq_off[g])]
        dr = self.dr
        if dr is not None and g:
            from .data_chunks import DeviceReads

            so, qo = dr.sig_off[: g + 1], dr.seq_off[: g + 1]
            out.dr = DeviceReads.from_device(dr.engine, so, qo, dr.dacs, dr.s2s, dr.iseq, dr.d_sig_off[: g + 1], dr.d_seq_off[: g + 1],
                                             dr.shift[:g].cpu().numpy(), dr.scale[:g].cpu().numpy())
        else:
            out.dr = None
        return out


def _trim_span(size, sp, ts, ns, has_ns):
    """(offset, length) of dacs[sp:][ts:ns] inside dacs for arrays of non-negative bounds (src/remora/io.py:2003-2012: the
    signal of an alignment), python slice semantics: bounds beyond the end clip, an end in front of the start leaves
    nothing; `ns` counts where `has_ns`."""
    size, sp, ts, ns = (np.asarray(x, np.int64) for x in (size, sp, ts, ns))
    start = np.minimum(sp, size)
    rem = size - start
    a = np.minimum(ts, rem)
    b = np.where(has_ns, np.minimum(ns, rem), rem)
    return start + a, np.maximum(b - a, 0)


_REF_ANCHOR_ERRORS = {1: "Discordant ref seq lengths", 2: "Invalid cigar op(s)", 3: "No match operations found in alignment cigar"}


def _weighted_median(values, counts):
    """np.median of an array that holds values[i] counts[i] times (float64), without building it: the middle order statistic,
    or numpy's mean of the two middle ones."""
    order = np.argsort(values, kind="stable")
    v, c = values[order], np.cumsum(counts[order])
    n = int(c[-1])
    kth = lambda k: v[np.searchsorted(c, k, side="right")]  # noqa: E731 - k-th smallest, zero-based
    return np.mean(np.asarray([kth(n // 2)] if n % 2 else [kth(n // 2 - 1), kth(n // 2)], np.float64))


def _median_mad_scaling(flat, start, length, cal_off, cal_scale, eng):
    """(shift_pa_to_norm, scale_pa_to_norm) float64[n] of io.Read.compute_pa_to_norm_scaling (src/remora/io.py:1851-1856:
    np.median of the pA signal, max(1, np.median(|pA - median|) * factor)) for n spans flat[start : start + length] of the
    device-resident decoded signal.  The GPU counts (rmr_signal_histograms: range, then histogram of every span); the float64
    arithmetic runs here on the occupied bins - the operations numpy applies to the samples, on every distinct sample once."""
    n = int(np.asarray(start).size)
    start, length = np.ascontiguousarray(start, np.int64), np.ascontiguousarray(length, np.int64)
    lo, hi = np.empty(n, np.int32), np.empty(n, np.int32)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    lib = L.lib()
    L.check(lib.rmr_signal_histograms(eng.handle, flat.data_ptr(), p(start), p(length), n, p(lo), p(hi), None, None))
    hist_off = np.zeros(n + 1, np.int64)
    np.cumsum(np.maximum(hi.astype(np.int64) - lo + 1, 0), out=hist_off[1:])
    hist = np.zeros(max(int(hist_off[-1]), 1), np.uint32)
    L.check(lib.rmr_signal_histograms(eng.handle, flat.data_ptr(), p(start), p(length), n, p(lo), p(hi), p(hist_off), p(hist)))
    sm, sd = np.empty(n, np.float64), np.empty(n, np.float64)
    for i in range(n):
        h = hist[hist_off[i] : hist_off[i + 1]]
        occ = np.nonzero(h)[0]
        counts = h[occ].astype(np.int64)
        if int(counts.sum()) != int(length[i]):
            raise RemoraError("signal histogram does not add up to the span's length")
        dacs = (occ + int(lo[i])).astype(np.int16)
        pa = (dacs - float(cal_off[i])) / float(cal_scale[i])  # Read.pa_signal: int16 array with Python floats -> float64
        med = _weighted_median(pa, counts)
        sm[i] = med
        sd[i] = max(1.0, _weighted_median(np.abs(pa - med), counts) * PA_TO_NORM_SCALING_FACTOR)
    return sm, sd


_SCRATCH = threading.local()


def _scratch(name, count, dtype):
    """A per-thread array of at least `count` items that lives across calls: the ingest's per-batch work buffers (10 MB of
    reference-to-signal knots, the bases of a batch) cost more in first-touch page faults than in the work done on them when
    they were allocated fresh for every batch."""
    buf = getattr(_SCRATCH, name, None)
    if buf is None or buf.size < count or buf.dtype != np.dtype(dtype):
        buf = np.empty(int(count) + int(count) // 4 + 1024, dtype)
        setattr(_SCRATCH, name, buf)
    return buf[:count]


INGEST_CLOCK = {}


def _section_clock():
    """clock(name): the time since the previous call is added to INGEST_CLOCK[name]."""
    import time

    last = [time.perf_counter()]

    def clock(name):
        now = time.perf_counter()
        INGEST_CLOCK[name] = INGEST_CLOCK.get(name, 0.0) + now - last[0]
        last[0] = now

    return clock


def _ingest_batch(rb, records, signals, eng, pa_scaling, skip_non_primary, ref_anchored=False):
    """IngestBatch of one raw BAM batch, or None when nothing of it is kept.  Everything Read.from_pod5 + add_alignment +
    into_remora_read (forward signal) do per read, for the batch: trimming by sp / ts / ns, strand-aware
    sequence, move tables -> query_to_signal (one launch), sm / sd composed with the calibration, the trim to the mapped span.
    `ref_anchored` (`infer --reference-anchored`; `rb` read with want_ref): the reads are anchored on the reference bases of
    their alignments instead of their basecalls - move table and CIGAR composed per record by native threads
    (rmr_ref_anchor_batch: src/remora/io.py:2066-2084), the reference sequence rebuilt from MD by the native reader, and
    the same assembly kernel cuts signal and mapping (ref_to_signal in place of query_to_signal).
    Reads without sm / sd tags get the median / MAD scaling from GPU histograms of their trimmed signal (_median_mad_scaling).
    A batch that holds something the array form does not reproduce (negative trim tags, bases outside A-Z, an aligned record
    without a move table) is returned as the string "slow": the caller sends its records through the per-read path.
    RMR_INFER_TIMING=1: seconds per section accumulate in io.INGEST_CLOCK (tools/prof_ingest_batches.py prints them)."""
    import torch

    from .data_chunks import DeviceReads

    clock = _section_clock() if os.environ.get("RMR_INFER_TIMING") else (lambda name: None)
    n_all = rb.n
    flag = rb.flag
    keep_mask = np.ones(n_all, bool) if not skip_non_primary else (flag & 0x900) == 0
    # the read a record's signal belongs to: the parent (pi) of a split read, else the record's name
    names, pi = rb.names, rb.pi
    row_of, kept = signals._row, []
    rows = []
    has_pi = (rb.has & 64) != 0
    no, po = rb.name_off.tolist(), rb.pi_off.tolist()
    for i in np.nonzero(keep_mask)[0].tolist():
        rid = (pi[po[i] : po[i + 1]] if has_pi[i] else names[no[i] : no[i + 1]]).decode("latin-1")
        r = row_of.get(rid)
        if r is not None:
            kept.append(i)
            rows.append(r)
    clock("ids")
    if not kept:
        return None
    keep = np.asarray(kept, np.int64)
    nk = keep.size
    has = rb.has[keep]
    sp = np.where(has & 8, rb.sp[keep], 0).astype(np.int64)
    ts = np.where(has & 2, rb.ts[keep], 0).astype(np.int64)
    ns = rb.ns[keep].astype(np.int64)
    if (sp < 0).any() or (ts < 0).any() or (((has & 4) != 0) & (ns < 0)).any():
        return "slow"
    seq_len_all = np.diff(rb.seq_off)
    sb = np.frombuffer(rb.seq, np.uint8)
    if sb.size and (sb.min() < 65 or sb.max() > 90):
        return "slow"
    clock("tags")
    # ---- signals: every distinct read of the batch decoded once, on the GPU ----
    uniq, inv = np.unique(np.asarray(rows, np.int64), return_inverse=True)
    first, addr, size, samples = signals.rows_of_reads(uniq)
    flat, row_out = vbz_decode_rows(addr, size, samples, eng)
    clock("vbz_decode")
    read_start = row_out[first[:-1]]                      # where a distinct read's samples begin in `flat`
    read_size = row_out[first[1:]] - read_start
    size_k, base_k = read_size[inv], read_start[inv]
    off, sig_len = _trim_span(size_k, sp, ts, ns, (has & 4) != 0)
    src_start = base_k + off
    # ---- norm scaling: the sm / sd tags, or - for records without both - median and MAD of the trimmed signal (:2036-2041);
    #      unused when the caller overrides the scaling (into_remora_read, :2147-2153) ----
    sm_k, sd_k = rb.sm[keep].astype(np.float64), rb.sd[keep].astype(np.float64)
    untagged = np.nonzero((has & 48) != 48)[0]
    if untagged.size and pa_scaling is None:
        if (sig_len[untagged] <= 0).any():
            return "slow"  # the median of nothing: numpy's warning and NaN belong to the per-read path
        row = uniq[inv][untagged]
        sm_k[untagged], sd_k[untagged] = _median_mad_scaling(flat, src_start[untagged], sig_len[untagged], signals._cal_off[row],
                                                             signals._cal_scale[row], eng)
    sl_all = np.zeros(n_all, np.int64)
    sl_all[keep] = sig_len
    dev = eng.torch_device
    total_mv = int(rb.mv_off[n_all])
    is_rev_all = (flag & 16) != 0
    if ref_anchored:
        # ---- move table and alignment composed per record on native threads: ref_to_signal of every record with a reference ----
        if not getattr(rb, "want_ref", False):
            raise RemoraError("reference-anchored ingest needs BAM batches read with want_ref")
        ref_len_all = np.full(n_all, -1, np.int64)
        mapped_ref = (rb.ref_id[keep] >= 0) & (rb.ref_ok[keep] != 0) & ((flag[keep] & 4) == 0)
        if (mapped_ref & ((has & 1) == 0)).any():
            return "slow"  # an aligned record without a move table: the per-read path's business
        ref_len_all[keep[mapped_ref]] = np.diff(rb.refseq_off)[keep[mapped_ref]]
        r2s_off = np.zeros(n_all + 1, np.int64)
        np.cumsum(np.maximum(ref_len_all, -1) + 1, out=r2s_off[1:])
        r2s = _scratch("r2s", max(int(r2s_off[-1]), 1), np.int64)  # (uploaded before this function returns)
        status_all = np.zeros(n_all, np.int32)
        pp = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        mv_arr = rb.mv if total_mv else np.zeros(1, np.int8)
        cig_arr = np.ascontiguousarray(rb.cigar, np.uint32) if rb.cigar.size else np.zeros(1, np.uint32)
        rev8 = np.ascontiguousarray(is_rev_all, np.uint8)
        L.check(L.lib().rmr_ref_anchor_batch(n_all, pp(mv_arr), pp(np.ascontiguousarray(rb.mv_off, np.int64)), pp(sl_all),
                                             pp(np.ascontiguousarray(seq_len_all, np.int64)), pp(cig_arr),
                                             pp(np.ascontiguousarray(rb.cigar_off, np.int64)), pp(rev8), pp(ref_len_all), pp(r2s),
                                             pp(r2s_off), pp(status_all), int(os.environ.get("RMR_PACK_THREADS", "8"))))
        status = status_all[keep]
        if (status == 4).any():
            return "slow"  # a CIGAR with an empty match run: the array form of the per-read path does what numpy does with it
    else:
        # ---- move tables of the whole raw batch in one launch (tables of records that are not kept: length 0 -> ignored) ----
        d_mv = torch.from_numpy(rb.mv if total_mv else np.zeros(1, np.int8)).to(dev)
        d_off, d_sl, d_ql = (torch.from_numpy(np.ascontiguousarray(x, np.int64)).to(dev) for x in (rb.mv_off, sl_all, seq_len_all))
        d_q2s = torch.empty(max(total_mv, 1), dtype=torch.int64, device=dev)
        # torch.empty, not zeros: a fill would be queued on this thread's torch stream while the kernel runs on the ingest
        # engine's own (unordered) stream and could land AFTER it; moves_batch_kernel writes counts[b] and status[b] of every
        # record, tables of length 0 included
        d_cnt = torch.empty(n_all, dtype=torch.int64, device=dev)
        d_st = torch.empty(n_all, dtype=torch.int32, device=dev)
        L.check(L.lib().rmr_parse_moves_batch(eng.handle, d_mv.data_ptr(), d_off.data_ptr(), d_sl.data_ptr(), d_ql.data_ptr(), n_all, 1, 0,
                                              d_q2s.data_ptr(), d_cnt.data_ptr(), d_st.data_ptr(), L.MEM_DEVICE))
        eng.synchronize()
        status = d_st.cpu().numpy()[keep]
    clock("moves_or_ref_anchor")
    # ---- who can be called, and why not (the texts of add_alignment / into_remora_read, in their order) ----
    is_rev = (flag[keep] & 16) != 0
    err = [None] * nk
    mv_len = np.diff(rb.mv_off)[keep]
    unmapped_rev = (rb.ref_id[keep] < 0) & is_rev
    no_mv = np.zeros(nk, bool) if ref_anchored else (has & 1) == 0
    for k in np.nonzero(unmapped_rev | no_mv | (status != 0))[0].tolist():  # (the reads with nothing to report are not visited)
        if unmapped_rev[k]:
            err[k] = "Unmapped reads cannot map to reverse strand."
        elif ref_anchored:
            st = int(status[k])
            if st in (8, 9):  # no move table / no reference sequence: nothing to anchor the read on (io.py:2131-2137)
                err[k] = "Read prep error: Missing reference alignment"
            elif st > 0:
                err[k] = _REF_ANCHOR_ERRORS[st]
            elif st < 0:
                err[k] = _MOVE_ERRORS.get(st, "empty move tag" if mv_len[k] < 1 else f"move table stride {int(rb.mv[rb.mv_off[keep[k]]])}")
        elif no_mv[k]:
            err[k] = "Read prep error: Missing query_to_signal (move table)"
        elif status[k] != 0:
            err[k] = _MOVE_ERRORS.get(int(status[k]), "empty move tag" if mv_len[k] < 1 else f"move table stride {int(rb.mv[rb.mv_off[keep[k]]])}")
    good = np.asarray([k for k in range(nk) if err[k] is None], np.int64)
    out = IngestBatch()
    out.rb, out.records, out.keep, out.err, out.good = rb, records, keep, err, good
    out.dr, out.reads, out.seq, out.seq_off, out.iseq = None, [], b"", np.zeros(1, np.int64), np.zeros(0, np.int8)
    out.ref_fwd, out.ref_fwd_off = (b"", np.zeros(nk + 1, np.int64)) if ref_anchored else (None, None)
    out.per_read = None
    if not good.size:
        return out
    gk = keep[good]
    clock("errors")
    # ---- strand-aware bases and their integer codes, one native pass (rmr_orient_bases): seq = revcomp(query_sequence) for
    #      reverse-strand records (:2023); reference anchor: the reference bases of the alignments instead - forward strand,
    #      upper case, for the output records (ref_fwd), read orientation for the reads (ref_seq, :2058-2060) ----
    from .util import _SEQ_TRANS

    src_blob, src_off = (rb.refseq, rb.refseq_off) if ref_anchored else (rb.seq, rb.seq_off)
    start = np.ascontiguousarray(src_off[gk], np.int64)
    seq_len = np.ascontiguousarray(src_off[gk + 1] - src_off[gk], np.int64)
    out.seq_off = np.zeros(good.size + 1, np.int64)
    np.cumsum(seq_len, out=out.seq_off[1:])
    n_bases = max(int(out.seq_off[-1]), 1)
    oriented, iseq = _scratch("oriented", n_bases, np.uint8), np.empty(n_bases, np.int8)  # (oriented, fwd: copied into bytes below)
    fwd = _scratch("fwd", n_bases, np.uint8) if ref_anchored else None
    rev_good = np.ascontiguousarray(is_rev[good], np.uint8)
    pq = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    L.check(L.lib().rmr_orient_bases(src_blob if src_blob else b"\0", pq(start), pq(seq_len), pq(rev_good), int(good.size), int(bool(ref_anchored)),
                                     _COMP_BYTES, _SEQ_TRANS, pq(fwd) if fwd is not None else None, pq(oriented), pq(iseq),
                                     int(os.environ.get("RMR_PACK_THREADS", "8"))))
    oriented, iseq = oriented[: int(out.seq_off[-1])], iseq[: int(out.seq_off[-1])]
    out.seq, out.iseq = oriented.tobytes(), iseq
    if ref_anchored:
        out.ref_fwd = fwd[: int(out.seq_off[-1])].tobytes()
        per_kept = np.zeros(nk, np.int64)
        per_kept[good] = seq_len
        np.cumsum(per_kept, out=out.ref_fwd_off[1:])
    clock("orient_bases")
    # ---- scaling: sm / sd composed with the calibration (:2036-2041, :2147-2153), float64 as on the per-read path ----
    cal_off, cal_scale = signals._cal_off[uniq][inv][good].astype(np.float64), signals._cal_scale[uniq][inv][good].astype(np.float64)
    if pa_scaling is None:
        sm, sd = sm_k[good], sd_k[good]
    else:
        sm, sd = np.full(good.size, float(pa_scaling[0])), np.full(good.size, float(pa_scaling[1]))
    shift, scale = cal_off + cal_scale * sm, cal_scale * sd
    # ---- dacs = trimmed[map[0]:map[-1]], mapping re-based: assembled where the pieces already are ----
    n_good = int(good.size)
    n_seq = int(out.seq_off[-1])
    if ref_anchored:  # ref_to_signal of the good records takes query_to_signal's place
        d_q2s = torch.from_numpy(r2s).to(dev)
        map_off = np.ascontiguousarray(r2s_off[gk], np.int64)
        sig_total = int((r2s[r2s_off[gk] + seq_len] - r2s[r2s_off[gk]]).sum())
    else:
        map_off = np.ascontiguousarray(rb.mv_off[gk], np.int64)
        sig_total = int(sig_len[good].sum())
    clock("upload_mapping")
    dacs = torch.empty(max(sig_total, 1), dtype=torch.int16, device=dev)
    s2s = torch.empty(n_seq + n_good, dtype=torch.int64, device=dev)
    d_sig_off = torch.empty(n_good + 1, dtype=torch.int64, device=dev)
    d_seq_off = torch.empty(n_good + 1, dtype=torch.int64, device=dev)
    sig_off = np.zeros(n_good + 1, np.int64)
    p = lambda x: x.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    ss = np.ascontiguousarray(src_start[good], np.int64)
    L.check(L.lib().rmr_assemble_reads(eng.handle, n_good, flat.data_ptr(), p(ss), d_q2s.data_ptr(), p(map_off), p(seq_len), dacs.data_ptr(),
                                       dacs.numel(), s2s.data_ptr(), d_sig_off.data_ptr(), d_seq_off.data_ptr(), p(sig_off)))
    clock("assemble_launch")
    d_iseq = torch.from_numpy(iseq).to(dev)
    eng.synchronize()
    clock("assemble_wait")
    from .engine import get_prep_engine

    # the batch is used on the extraction engine (own stream, own mutex): not behind the ingest of the next batches
    out.dr = DeviceReads.from_device(get_prep_engine(eng.device), sig_off, out.seq_off, dacs, s2s, d_iseq, d_sig_off, d_seq_off, shift, scale)
    out.reads = [ReadStub(sh, sc) for sh, sc in zip(shift.tolist(), scale.tolist())]
    return out


def _readahead(gen, depth=2):
    """`gen` iterated by a thread of its own, up to `depth` items ahead of the consumer (the native BAM reader releases the
    GIL while it parses a batch: reading batch k + 1 overlaps the assembly of batch k).  Exceptions of the generator are
    raised in the consumer; closing this generator stops the thread and closes `gen` there."""
    import queue
    import threading

    q = queue.Queue(maxsize=max(int(depth), 1))
    stop = threading.Event()

    def put(item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def run():
        try:
            for item in gen:
                if not put(("item", item)):
                    return
            put(("end", None))
        except BaseException as e:  # noqa: BLE001 - handed to the consumer
            put(("error", e))
        finally:
            if hasattr(gen, "close"):
                gen.close()

    th = threading.Thread(target=run, name="rmr-bam-readahead", daemon=True)
    th.start()
    try:
        while True:
            kind, value = q.get()
            if kind == "end":
                return
            if kind == "error":
                raise value
            yield value
    finally:
        stop.set()
        th.join()


def iter_ingest_batches(pod5_path, bam_path, pa_scaling=None, skip_non_primary=True, batch=256, shard=None, device=None,
                        ref_anchored=False):
    """The batch form of iter_reads_from_pod5_and_bam for calling of forward signal, anchored on the basecalls or (`ref_anchored`)
    on the reference bases of the alignments: IngestBatch objects
    (arrays on the GPU) instead of (io.Read, error) pairs; a batch the array form does not cover comes as the list of
    (io.Read, error) pairs the per-read path yields for its records."""
    from .engine import get_ingest_engine

    signals = Pod5File(pod5_path)
    eng = get_ingest_engine(device)
    raw_batches = iter_bam_raw_batches(bam_path, want_ref=bool(ref_anchored), batch=batch, shard=shard)
    raw_batches = _readahead(raw_batches, 2)  # the native parser releases the GIL: BAM batches are read one ahead
    for rb, records in raw_batches:
        got = _ingest_batch(rb, records, signals, eng, pa_scaling, skip_non_primary, ref_anchored=ref_anchored)
        if got is None:
            continue
        per_read = lambda rb=rb: list(_reads_of_records(records(rb), signals, eng, False, pa_scaling, skip_non_primary,  # noqa: E731
                                                        max(batch, 2), bool(ref_anchored)))
        if isinstance(got, str):  # "slow": the per-read path for the records of this batch
            yield per_read()
            continue
        got.per_read = per_read  # for a consumer that finds it cannot use the batch after all (dataset prepare: a refiner's band error)
        yield got


# ---- BAM output: the reference writes `pysam.AlignedSegment.from_dict(
# io_read.full_align)` = the input record + MM/ML tags (src/remora/inference.py:450, :619-623);
# here the stored record bytes are copied, old MM/ML tags dropped, new ones appended, and the
# stream is BGZF-compressed (gzip members <= 64 KiB with the BC extra field + EOF marker). ----
import zlib

_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def read_bam_header_bytes(bam_path):
    """Everything before the first alignment record (magic, text header, reference list)."""
    with gzip.open(bam_path, "rb") as fh:
        out = bytearray(_read_exact(fh, 8))
        if out[:4] != b"BAM\x01":
            raise RemoraError(f"{bam_path} is not a BAM file")
        out += _read_exact(fh, struct.unpack_from("<i", out, 4)[0])
        nref = _read_exact(fh, 4)
        out += nref
        for _ in range(struct.unpack("<i", nref)[0]):
            ln = _read_exact(fh, 4)
            out += ln + _read_exact(fh, struct.unpack("<i", ln)[0] + 4)
    return bytes(out)


def _pack_seq(seq):
    codes = np.frombuffer(seq.encode(), dtype=np.uint8)
    lut = np.full(256, 15, np.uint8)
    for i, c in enumerate(_SEQ_NT16):
        lut[ord(c)] = i
    nib = lut[codes]
    if nib.size % 2:
        nib = np.concatenate([nib, np.zeros(1, np.uint8)])
    return ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8).tobytes()


def record_with_mod_tags(rec, mm_tag, ml_tag, ref_anchored_seq=None):
    """Record bytes (incl. block_size) of `rec` with any MM/ML/Mm/Ml tags replaced by the given
    MM string / ML uint8 values (None = leave the record without modified-base tags).
    `ref_anchored_seq` (the reference bases of the alignment in forward-strand orientation) turns the
    record into the reference-anchored form the reference writes: CIGAR `<len>M`, that sequence, no
    qualities (src/remora/inference.py:452-458)."""
    tag_region = rec.raw[rec.tags_offset :]
    # records that carry no modified-base tag keep their tag bytes as they are; only a record in which one of the four
    # tag headers occurs (as a tag, or by chance inside another tag's data) is walked tag by tag (33 us in Python)
    tr = bytes(tag_region)
    if any(tr.find(h) >= 0 for h in (b"MMZ", b"MLB", b"MmZ", b"MlB")):
        kept = b"".join(tag_region[s:e] for name, s, e in rec.tag_spans if name not in ("MM", "ML", "Mm", "Ml"))
    else:
        kept = tr
    new = b""
    if mm_tag is not None:
        ml = np.asarray(ml_tag, dtype=np.uint8)
        new = b"MMZ" + mm_tag.encode() + b"\x00" + b"MLBC" + struct.pack("<i", ml.size) + ml.tobytes()
    core = rec.raw[: rec.tags_offset]
    if ref_anchored_seq is not None:
        l_name = core[8]
        n = len(ref_anchored_seq)
        fixed = bytearray(core[:32])
        struct.pack_into("<H", fixed, 12, 1)  # n_cigar_op
        struct.pack_into("<i", fixed, 16, n)  # l_seq
        core = (bytes(fixed) + core[32 : 32 + l_name] + struct.pack("<I", (n << 4) | 0) + _pack_seq(ref_anchored_seq)
                + b"\xff" * n)
    body = core + kept + new
    return struct.pack("<i", len(body)) + body


def records_with_mod_tags_batch(records, mm, mm_off, ml, ml_off, has_tags):
    """record_with_mod_tags for a batch in one native call (rmr_records_with_mod_tags): the records' bytes (block_size
    included) one after the other, old MM/ML/Mm/Ml tags removed, read r's MM / ML slices appended where has_tags[r].
    `records`: objects with `.raw` (bytes) and `.tags_offset`; the tag arrays as format_mm_ml_tags_batch returns them."""
    n = len(records)
    raws = [bytes(r.raw) if not isinstance(r.raw, bytes) else r.raw for r in records]
    ptrs = (ctypes.c_char_p * n)(*raws)
    raw_len = np.fromiter((len(b) for b in raws), np.int64, n)
    tags_off = np.fromiter((r.tags_offset for r in records), np.int64, n)
    has = np.ascontiguousarray(has_tags, np.uint8)
    mm_off, ml_off = np.ascontiguousarray(mm_off, np.int64), np.ascontiguousarray(ml_off, np.int64)
    out = np.empty(int(raw_len.sum()) + 16 * n + int(mm_off[-1]) + int(ml_off[-1]) + 16, np.uint8)
    out_len = ctypes.c_int64()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    L.check(L.lib().rmr_records_with_mod_tags(n, ctypes.cast(ptrs, ctypes.c_void_p), p(raw_len), p(tags_off), p(mm), p(mm_off), p(ml),
                                              p(ml_off), p(has), p(out), out.size, ctypes.byref(out_len)))
    return out[: out_len.value].tobytes()


def records_with_mod_tags_flat(raw, raw_start, raw_len, tags_off, mm, mm_off, ml, ml_off, has_tags, ref_seq=None, ref_off=None):
    """records_with_mod_tags_batch for records that lie in ONE bytes object (a RawBamBatch's `raw`): record r is
    raw[raw_start[r] : raw_start[r] + raw_len[r]], its tags begin tags_off[r] bytes into it.  `ref_seq` / `ref_off`
    (reference-anchored calling): a record that gets tags and owns a non-empty slice of `ref_seq` - the forward-strand reference
    bases of its alignment - is written as record_with_mod_tags(..., ref_anchored_seq=...) writes it
    (rmr_records_with_mod_tags_ref)."""
    n = int(np.asarray(raw_len).size)
    base = ctypes.cast(ctypes.c_char_p(raw), ctypes.c_void_p).value or 0
    ptrs = np.uint64(base) + np.ascontiguousarray(raw_start, np.int64).astype(np.uint64)
    raw_len, tags_off = np.ascontiguousarray(raw_len, np.int64), np.ascontiguousarray(tags_off, np.int64)
    has = np.ascontiguousarray(has_tags, np.uint8)
    mm_off, ml_off = np.ascontiguousarray(mm_off, np.int64), np.ascontiguousarray(ml_off, np.int64)
    out_len = ctypes.c_int64()
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    if ref_seq is not None:
        ref_off = np.ascontiguousarray(ref_off, np.int64)
        ref_arr = np.frombuffer(ref_seq, np.uint8) if len(ref_seq) else np.zeros(1, np.uint8)
        out = np.empty(int(raw_len.sum()) + 24 * n + int(mm_off[-1]) + int(ml_off[-1]) + 2 * int(ref_off[-1]) + 16, np.uint8)
        L.check(L.lib().rmr_records_with_mod_tags_ref(n, p(ptrs), p(raw_len), p(tags_off), p(mm), p(mm_off), p(ml), p(ml_off), p(has),
                                                      p(ref_arr), p(ref_off), p(out), out.size, ctypes.byref(out_len)))
        return out[: out_len.value].tobytes()
    out = np.empty(int(raw_len.sum()) + 16 * n + int(mm_off[-1]) + int(ml_off[-1]) + 16, np.uint8)
    L.check(L.lib().rmr_records_with_mod_tags(n, p(ptrs), p(raw_len), p(tags_off), p(mm), p(mm_off), p(ml), p(ml_off), p(has), p(out),
                                              out.size, ctypes.byref(out_len)))
    return out[: out_len.value].tobytes()


_BGZF_LEVEL = int(os.environ.get("RMR_BAM_LEVEL", "6"))  # htslib's default level
# Level 1 (`--bam-level 1`: "fast") takes zlib's Z_HUFFMAN_ONLY - 1.9x the speed of level 1's matcher on BAM records with move
# tables for a 19 % larger file (still plain deflate: any reader takes it): 28.1 k -> 33.2 k reads/s file to file with six
# processes on 16 cores, 3.3 -> 3.9 GB of output (profiles/r03_infer_cli_336k_byte_shares_huffman.log); other levels: zlib's matcher


def _eff_cpus():
    from .util import effective_cpu_count

    return effective_cpu_count()


def _bgzf_block(chunk, level=None):
    """One BGZF member (gzip with the BC extra field) for up to 64 KiB of payload."""
    level = _BGZF_LEVEL if level is None else int(level)
    strategy = zlib.Z_HUFFMAN_ONLY if level == 1 else zlib.Z_DEFAULT_STRATEGY
    comp = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    cdata = comp.compress(chunk) + comp.flush()
    return b"".join((b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00", struct.pack("<H", len(cdata) + 25),
                     cdata, struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))))


def _bgzf_huffman_native(chunk):
    """BGZF members (Huffman-only deflate, rmr_bgzf_huffman) for a run of payload bytes: one member per 0xFF00 bytes."""
    n = len(chunk)
    out = np.empty(max((n + 0xFEFF) // 0xFF00, 1) * 65311, np.uint8)
    out_len = ctypes.c_int64()
    src = ctypes.cast(ctypes.c_char_p(chunk), ctypes.c_void_p) if n else None  # (`chunk` is a bytes object: read in place)
    L.check(L.lib().rmr_bgzf_huffman(src, n, 1, out.ctypes.data_as(ctypes.c_void_p), out.size, ctypes.byref(out_len)))
    return memoryview(out)[: out_len.value]  # written to the file as it is (no copy into a bytes object)


class BamWriter:
    """Minimal BGZF/BAM writer: header bytes copied from the template BAM, records appended.  Blocks are deflated
    by a small thread pool (zlib releases the GIL) and written in order, so compression - ~1 ms per 5 kb read with
    its move table - runs beside the caller instead of in it; the file is the same as with inline compression."""

    def __init__(self, path, header_bytes, threads=None, max_pending=None, eof=True, level=None):
        """`header_bytes` = everything before the first record (b"" for a part file that holds records only);
        `eof=False` leaves the end-of-file marker out — part files of a multi-GPU run are whole BGZF members and are
        joined byte for byte by `concat_bam_parts`."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        # deflate at level 6 runs at ~35 MB/s per thread and a 5 kb read with its move table is ~25 KB of BAM: four
        # threads cap the writer at ~5 k reads/s, which the batched GPU path exceeds tenfold
        if threads is None:
            from .util import effective_cpu_count

            # the cores this process may really use (cgroup quota, not os.cpu_count()), shared with the other ranks of
            # the node when several processes run side by side
            local_ranks = max(int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1), 1)
            threads = min(16, max(4, effective_cpu_count() // local_ranks))
        if max_pending is None:
            max_pending = 8 * int(threads)

        self._fh = open(path, "wb")
        self._eof = bool(eof)
        self._level = level  # zlib level of the BGZF members (None = RMR_BAM_LEVEL, default 6 = htslib's)
        self._buf = bytearray()
        self._pool = ThreadPoolExecutor(max_workers=max(int(threads), 1))
        self._pending, self._max_pending = deque(), int(max_pending)
        # level 1 with the default strategy (Huffman coding only): the library's own encoder, 2.5x zlib's on BAM records;
        # whole runs of 16 payloads per job instead of one payload per job (RMR_BGZF_NATIVE=0: zlib for these too)
        lvl = _BGZF_LEVEL if level is None else int(level)
        self._native = lvl == 1 and os.environ.get("RMR_BGZF_NATIVE", "1") != "0"
        # the header goes through write(): one with many reference sequences (hg38 with alt / decoy contigs: > 64 KiB) is
        # split into members of at most 0xFF00 bytes like everything else (a BGZF member holds at most 64 KiB)
        self.write(bytes(header_bytes))

    def _drain(self, keep):
        while len(self._pending) > keep:
            self._fh.write(self._pending.popleft().result())

    def _flush_block(self, chunk):
        self._pending.append(self._pool.submit(_bgzf_block, bytes(chunk), self._level))
        self._drain(self._max_pending)

    def write(self, record_bytes):
        """Append record bytes (one record or a whole batch of them); full 0xFF00-byte blocks go to the deflate pool."""
        if self._native:
            buf = self._buf
            buf += record_bytes
            run = 16 * 0xFF00
            if len(buf) >= run:  # whole payloads only: the members are cut exactly where the per-block path cuts them
                k = len(buf) // 0xFF00 * 0xFF00
                view = memoryview(buf)
                for a in range(0, k, run):
                    self._pending.append(self._pool.submit(_bgzf_huffman_native, bytes(view[a : min(a + run, k)])))
                del view
                self._buf = bytearray(buf[k:])
                self._drain(self._max_pending)
            return
        buf = self._buf
        if len(buf) + len(record_bytes) < 0xFF00:
            buf += record_bytes
            return
        view = memoryview(record_bytes)
        start = 0
        if buf:  # top the pending partial block up first
            start = 0xFF00 - len(buf)
            buf += view[:start]
            self._flush_block(buf)
            self._buf = buf = bytearray()
        end = len(view)
        while end - start >= 0xFF00:  # whole blocks straight from the caller's buffer (no front deletions of a large bytearray)
            self._flush_block(view[start : start + 0xFF00])
            start += 0xFF00
        buf += view[start:]

    def close(self):
        if self._fh is None:
            return
        if self._buf:
            if self._native:
                self._pending.append(self._pool.submit(_bgzf_huffman_native, bytes(self._buf)))
            else:
                self._flush_block(self._buf)
            self._buf = bytearray()
        self._drain(0)
        self._pool.shutdown()
        if self._eof:
            self._fh.write(_BGZF_EOF)
        self._fh.close()
        self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def concat_bam_parts(out_path, part_paths, remove=True):
    """Join the part files of a multi-GPU run (rank 0: header + its records, the others: records only, none with an
    end-of-file marker) into one BAM: BGZF members are self-contained, so the parts are appended byte for byte and the
    28-byte EOF marker closes the file (SAM spec 4.1).  Records keep the order of the input file because the ranks took
    contiguous shares in rank order.  With `remove` the first part BECOMES the output (a rename, nothing copied) and the
    others are appended inside the kernel (os.copy_file_range / os.sendfile; a plain read-write loop where those are
    not to be had) - the join of a run is a copy of everything the other ranks wrote, on rank 0, inside the clock."""
    import shutil

    part_paths = list(part_paths)

    how = {"copy_file_range": 0, "sendfile": 0, "read_write": 0}  # parts appended by each way (returned: tests look at it)

    def append(src_path, out):
        with open(src_path, "rb") as fh:
            left = os.fstat(fh.fileno()).st_size
            out.flush()
            for name in ("copy_file_range", "sendfile"):
                if not hasattr(os, name):
                    continue
                try:
                    while left > 0:
                        if name == "copy_file_range":  # (both advance the file offsets of the descriptors they are given)
                            n = os.copy_file_range(fh.fileno(), out.fileno(), min(left, 1 << 30))
                        else:
                            n = os.sendfile(out.fileno(), fh.fileno(), None, min(left, 1 << 30))
                        if n == 0:
                            break
                        left -= n
                    if left == 0:
                        how[name] += 1
                        out.seek(0, os.SEEK_END)  # the Python file object did not see the descriptor move
                        return
                except OSError:
                    pass  # not supported between these files: the next way down
            out.seek(0, os.SEEK_END)
            shutil.copyfileobj(fh, out, 1 << 22)  # (fh stands where the kernel copies stopped)
            how["read_write"] += 1

    rest = part_paths
    if remove and part_paths:
        os.replace(part_paths[0], out_path)
        rest = part_paths[1:]
    # "r+b" and a seek to the end, NOT "ab": copy_file_range / sendfile refuse a descriptor opened with O_APPEND (EBADF /
    # EINVAL), which silently sent every join down the read-write loop
    with open(out_path, "r+b" if remove and part_paths else "wb") as out:
        out.seek(0, os.SEEK_END)
        for p in rest:
            append(p, out)
        out.write(_BGZF_EOF)
    part_paths = rest
    if remove:
        for p in part_paths:
            os.remove(p)
    return how
