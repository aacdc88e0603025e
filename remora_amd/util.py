"""Sequence / motif / tag helpers of the hot path that stay on the host, mirroring the
reference's `remora.util` for these names (src/remora/util.py): seq_to_int :131-142,
int_to_seq :145-159, softmax_axis1 :182-186, Motif :190-378, find_focus_bases_in_int_sequence
:413-426, format_mm_ml_tags :485-537.  Pure numpy / python (string and set handling; nothing
here is on the GPU roofline)."""
import array
import os
import re
import shutil
from dataclasses import dataclass
from itertools import product

import numpy as np

from . import RemoraError

CAN_ALPHABET = "ACGT"
CONV_ALPHABET = "ACGTN"
SINGLE_LETTER_CODE = {
    "A": "A", "C": "C", "G": "G", "T": "T", "B": "CGT", "D": "AGT", "H": "ACT", "K": "GT",
    "M": "AC", "N": "ACGT", "R": "AG", "S": "CG", "V": "ACG", "W": "AT", "Y": "CT",
}
BASES_TO_CODES = {v: k for k, v in SINGLE_LETTER_CODE.items()}
_SEQ_LUT = np.full(256, -1, dtype=int)
for _i, _b in enumerate(CAN_ALPHABET):
    _SEQ_LUT[ord(_b)] = _i
# the same table for bytes.translate (C speed, one pass, one byte out per base: 255 = -1 as int8)
_SEQ_TRANS = bytes((int(v) & 0xFF) for v in _SEQ_LUT)


def seq_to_int(seq):
    """A,C,G,T -> 0..3, any other upper-case letter -> -1 (src/remora/util.py:131-142)."""
    codes = np.frombuffer(seq.encode("ascii"), dtype=np.uint8)
    if codes.size and (codes.min() < ord("A") or codes.max() > ord("Z")):
        raise IndexError("sequence contains characters outside A-Z")
    return np.take(_SEQ_LUT, codes)  # half the time of the fancy index on a 7 kb read


def int_to_seq(np_seq, alphabet=CONV_ALPHABET):
    """src/remora/util.py:145-159 (-1 indexes the last letter, N, as in the reference)."""
    if np_seq.shape[0] == 0:
        return ""
    if np_seq.max() >= len(alphabet):
        raise RemoraError(f"Invalid value in int sequence ({np_seq.max()})")
    return "".join(alphabet[b] for b in np_seq)


def softmax_axis1(x):
    """src/remora/util.py:182-186."""
    shifted = x - np.max(x, axis=1, keepdims=True)
    e_x = np.exp(shifted)
    with np.errstate(divide="ignore"):
        return e_x / e_x.sum(axis=1, keepdims=True)


_PATTERNS = {}


def _int_pattern(raw_motif):
    """Per motif position the array of allowed base codes (cached per motif string)."""
    if raw_motif not in _PATTERNS:
        pats = [np.array([CAN_ALPHABET.index(b) for b in SINGLE_LETTER_CODE[c]]) for c in raw_motif]
        luts = []
        for p in pats:
            lut = np.zeros(5, dtype=bool)
            lut[p] = True
            luts.append(lut)
        _PATTERNS[raw_motif] = (pats, luts, [frozenset(p.tolist()) for p in pats])
    return _PATTERNS[raw_motif][0]


def _allowed_lut(raw_motif):
    _int_pattern(raw_motif)
    return _PATTERNS[raw_motif][1]


def _allowed_sets(raw_motif):
    _int_pattern(raw_motif)
    return _PATTERNS[raw_motif][2]


@dataclass
class Motif:
    """IUPAC motif + focus position (src/remora/util.py:190-378, the subset the hot path
    uses: normalisation, findall, match)."""

    raw_motif: str
    focus_pos: int = 0

    def __post_init__(self):
        try:
            self.focus_pos = int(self.focus_pos)
        except ValueError:
            raise RemoraError(f'Motif focus position not an integer: "{self.focus_pos}"')
        if not isinstance(self.raw_motif, str):
            raise RemoraError("Motif sequence must be a string")
        bad = set(self.raw_motif) - set(SINGLE_LETTER_CODE)
        if bad:
            raise RemoraError(f"Motif contains invalid characters: {bad}")
        if self.focus_pos >= len(self.raw_motif):
            raise RemoraError("Motif focus position is past the end of the motif")
        while len(self.raw_motif) > 1 and self.raw_motif.startswith("N"):
            self.raw_motif = self.raw_motif[1:]
            self.focus_pos -= 1
        while len(self.raw_motif) > 1 and self.raw_motif.endswith("N"):
            self.raw_motif = self.raw_motif[:-1]

    def to_tuple(self):
        return self.raw_motif, self.focus_pos

    def __hash__(self):
        return hash(self.to_tuple())

    @property
    def focus_base(self):
        return self.raw_motif[self.focus_pos]

    @property
    def any_context(self):
        return self.raw_motif == "N"

    @property
    def num_bases_after_focus(self):
        return len(self.raw_motif) - self.focus_pos - 1

    @property
    def pattern(self):
        return re.compile("(?=({}))".format("".join(f"[{SINGLE_LETTER_CODE[c]}]" for c in self.raw_motif)))

    @property
    def int_pattern(self):
        return _int_pattern(self.raw_motif)

    def findall(self, int_seq):
        """Start index of every (overlapping) hit in an integer sequence."""
        m = len(self.raw_motif)
        nwin = int_seq.size - m + 1
        if nwin <= 0:
            return np.zeros(0, dtype=np.int64)
        hit = None
        for po, (codes, allowed) in enumerate(zip(_int_pattern(self.raw_motif), _allowed_lut(self.raw_motif))):
            window = int_seq[po : po + nwin]
            # one allowed base: a comparison; several: the 5-entry table, whose index -1 (= N in the sequence) is never allowed
            here = window == codes[0] if codes.size == 1 else allowed[window]
            hit = here if hit is None else hit & here
        return np.flatnonzero(hit)

    def match(self, int_seq, pos):
        """Does the motif sit on `pos`?  A motif hanging over either end of the sequence is compared on the
        overlapping part only, as the reference does (src/remora/util.py:297-311)."""
        pat = _allowed_sets(self.raw_motif)
        st, en = pos - self.focus_pos, pos + self.num_bases_after_focus + 1
        if st < 0:
            pat, st = pat[-st:], 0
        if en > int_seq.size:
            pat, en = pat[: len(pat) - en + int_seq.size], int_seq.size
        return all(base in allowed for allowed, base in zip(pat, int_seq[st:en].tolist()))

    def match_many(self, int_seq, positions):
        """`match` for an array of positions at once -> bool array (same rule at the ends of the sequence)."""
        positions = np.asarray(positions, dtype=np.int64)
        ok = np.ones(positions.size, dtype=bool)
        if not positions.size:
            return ok
        last = int_seq.size - 1
        for po, allowed in enumerate(_allowed_lut(self.raw_motif)):
            idx = positions + (po - self.focus_pos)
            inside = (idx >= 0) & (idx <= last)
            ok &= ~inside | allowed[int_seq[np.clip(idx, 0, max(last, 0))]]
        return ok

    @property
    def possible_kmers(self):
        return ["".join(bs) for bs in product(*(SINGLE_LETTER_CODE[c] for c in self.raw_motif))]

    def is_super_set(self, other):
        """Every sequence `other` stands for is also one of this motif's (:313-337)."""
        if self.focus_pos > other.focus_pos or self.num_bases_after_focus > other.num_bases_after_focus:
            return False
        window = other.raw_motif[other.focus_pos - self.focus_pos : other.focus_pos + self.num_bases_after_focus + 1]
        return all(set(SINGLE_LETTER_CODE[ob]) <= set(SINGLE_LETTER_CODE[sb]) for sb, ob in zip(self.raw_motif, window))

    def merge(self, other):
        """One motif standing for exactly the union of both, or RemoraError (:339-378)."""
        if self == other or self.is_super_set(other):
            return self
        if other.is_super_set(self):
            return other
        if len(self.raw_motif) != len(other.raw_motif):
            raise RemoraError("Cannot merge motifs of different sizes")
        if self.focus_pos != other.focus_pos:
            raise RemoraError("Cannot merge motifs with different focus pos")
        union = set(self.possible_kmers) | set(other.possible_kmers)
        width = len(self.raw_motif)
        cand = Motif("".join(BASES_TO_CODES["".join(sorted({k[i] for k in union}))] for i in range(width)), self.focus_pos)
        if len(cand.raw_motif) < width:  # leading / trailing N columns were clipped by the constructor
            lead = self.focus_pos - cand.focus_pos
            cols = ["ACGT"] * lead + [SINGLE_LETTER_CODE[c] for c in cand.raw_motif] + \
                   ["ACGT"] * (width - len(cand.raw_motif) - lead)
            spanned = {"".join(bs) for bs in product(*cols)}
        else:
            spanned = set(cand.possible_kmers)
        if spanned != union:
            raise RemoraError(f"Cannot merge motifs {self} {other}")
        return cand


def merge_motifs(motifs):
    """Repeated pairwise merging until the set stops changing (src/remora/util.py:381-410)."""
    motifs = list({m if isinstance(m, Motif) else Motif(*m) for m in motifs})
    seen = None
    while len(motifs) > 1 and (seen is None or set(seen) != set(motifs)):
        seen = motifs
        absorbed, pool = set(), set()
        for a in seen:
            for b in seen[1:]:
                try:
                    ab = a.merge(b)
                except RemoraError:
                    pool.update((a, b))
                    continue
                if ab != a:
                    absorbed.add(a)
                if ab != b:
                    absorbed.add(b)
                pool.add(ab)
        motifs = list(pool - absorbed)
    return motifs


def find_focus_bases_in_int_sequence(int_seq, motifs):
    """Union of motif hits (+focus offset).  The reference builds a python `set` and iterates
    it (src/remora/util.py:413-426); chunk order downstream follows that iteration order, so
    it is reproduced here the same way."""
    hits = set()
    for mot in motifs:  # list -> set.update inserts one by one in list order, like the reference's generator
        hits.update((mot.findall(int_seq) + mot.focus_pos).tolist())
    return np.fromiter(hits, int, len(hits))


_SMALL_INT_STR = [str(i) for i in range(1024)]  # MM gaps are small: a table lookup instead of str() per gap


def format_mm_ml_tags(seq, poss, probs, mod_bases, can_base, strand="+"):
    """MM / ML SAM tags from per-site probabilities (src/remora/util.py:485-537)."""
    order = np.argsort(np.asarray(poss), kind="stable")
    sorted_pos = np.asarray(poss)[order]
    mm_tag, ml_tag = "", array.array("B")
    if sorted_pos.size == 0:
        return mm_tag, ml_tag
    # index of every called base among the canonical bases of the read (= running count - 1): a search in the list of
    # canonical-base positions instead of a cumulative sum over the whole sequence (48 -> 12 us on a 7 kb read)
    can_at = np.flatnonzero(np.frombuffer(seq.encode(), np.uint8) == ord(can_base))
    can_idx = np.searchsorted(can_at, sorted_pos, side="right") - 1
    gaps = np.diff(np.concatenate([[-1], can_idx])) - 1
    gl = gaps.tolist()
    gap_str = ",".join(map(_SMALL_INT_STR.__getitem__, gl)) if 0 <= min(gl) and max(gl) < len(_SMALL_INT_STR) else ",".join(map(str, gl))
    valid = [p is not None for p in probs] if isinstance(probs, list) else None
    if valid is not None and not all(valid):
        raise RemoraError("per-site None probabilities are not supported")
    probs = np.asarray(probs, dtype=np.float64)[order]
    for mi, mod_base in enumerate(mod_bases):
        mm_tag += f"{can_base}{strand}{mod_base}?,{gap_str};"
        scaled = np.floor(probs[:, mi] * 256)
        scaled[scaled == 256] = 255
        ml_tag.frombytes(scaled.astype(np.uint8).tobytes())  # (extend() walks the numpy array element by element: 40 us)
    return mm_tag, ml_tag


def format_mm_ml_tags_batch(seq_bytes, seq_off, pos, probs, call_off, mod_bases, can_base, strand="+"):
    """format_mm_ml_tags for a batch of reads in one native call (rmr_format_mm_ml, host code): `seq_bytes` = the reads'
    sequences one after the other (bytes, read r at seq_off[r] .. seq_off[r+1]), `pos` int64[N] / `probs` float64[N, n_mods]
    = all calls, read r's at call_off[r] .. call_off[r+1] (any order inside a read).  Returns (mm uint8[...], mm_off, ml
    uint8[...], ml_off): read r's MM string is mm[mm_off[r]:mm_off[r+1]].tobytes().decode(), its ML values
    ml[ml_off[r]:ml_off[r+1]] - byte for byte what format_mm_ml_tags returns per read (empty for a read without calls)."""
    import ctypes

    from . import _lib as L

    seq_off = np.ascontiguousarray(seq_off, np.int64)
    call_off = np.ascontiguousarray(call_off, np.int64)
    n = seq_off.size - 1
    pos = np.ascontiguousarray(pos, np.int64)
    probs = np.ascontiguousarray(probs, np.float64).reshape(pos.size, -1) if pos.size else np.zeros((0, len(mod_bases)), np.float64)
    n_mods = len(mod_bases)
    if probs.shape[1] != n_mods or call_off.size != n + 1:
        raise RemoraError("format_mm_ml_tags_batch: inconsistent array shapes")
    codes = b"".join(str(mb).encode() + b"\x00" for mb in mod_bases)
    head = sum(len(str(mb)) for mb in mod_bases) + 5 * n_mods
    mm = np.empty(int(pos.size) * n_mods * 21 + n * head + 16, np.uint8)  # a gap prints in <= 20 characters + its comma
    ml = np.empty(int(pos.size) * n_mods + 16, np.uint8)
    mm_off, ml_off = np.empty(n + 1, np.int64), np.empty(n + 1, np.int64)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    buf = seq_bytes if isinstance(seq_bytes, (bytes, bytearray)) else bytes(seq_bytes)
    L.check(L.lib().rmr_format_mm_ml(n, ctypes.cast(ctypes.c_char_p(bytes(buf)), ctypes.c_void_p), p(seq_off), p(pos), p(probs), p(call_off),
                                     n_mods, codes, can_base.encode(), strand.encode(), p(mm), mm.size, p(mm_off), p(ml), ml.size,
                                     p(ml_off)))
    return mm, mm_off, ml, ml_off


def resolve_path(fn_path):
    """Absolute, user-expanded, symlink-free path; None stays None (src/remora/util.py:161-167)."""
    return None if fn_path is None else os.path.realpath(os.path.expanduser(str(fn_path)))


def prepare_out_dir(out_dir, overwrite):
    """Fresh output directory; an existing path is an error unless `overwrite` (src/remora/util.py:59-69; the
    reference also opens its log file there)."""
    if os.path.exists(out_dir):
        if not overwrite:
            raise RemoraError("Refusing to overwrite existing directory.")
        shutil.rmtree(out_dir) if os.path.isdir(out_dir) else os.remove(out_dir)
    os.makedirs(out_dir, exist_ok=True)


def effective_cpu_count():
    """Cores this process may actually use: the smallest of os.cpu_count(), the scheduler affinity mask and the cgroup CPU
    quota (containers commonly show every host core in os.cpu_count() while cpu.max grants a fraction: thread pools sized
    by the former thrash).  At least 1."""
    import math
    import os

    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:  # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:  # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return max(int(n), 1)
