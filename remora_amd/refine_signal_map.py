"""Signal-mapping refinement: host mirror of `SigMapRefiner`
(src/remora/refine_signal_map.py:150-632) whose banded dynamic programming runs on the GPU
through `rmr_refine_signal_maps` (include/remora_hip.h; kernels in csrc/k_refine.hip).

What runs where:
  * levels, bands, band adjustment/validation, signal normalisation, forward DP (Viterbi or
    dwell penalty) and traceback: HIP kernels (`refine_maps` / `refine_reads` on host arrays,
    `refine_device_reads` on a resident `DeviceReads` batch; `refine_sig_map` keeps the
    reference's per-read signature).  No CPU fallback: without the library or a GPU the call raises.
  * the scalar re-scaling estimators: `rough_rescale` / `rescale` are float64 numpy on the host as in
    the reference (the quantiles through a one-sort replica of numpy's arithmetic);
    `rough_rescale_device` does the per-base level lookup, the centre-sample gather and the two sorts
    of a whole batch on the GPU and leaves only the 19-point line fit of every read on the host -
    bit-identical results."""
import ctypes
import dataclasses
import os
from itertools import product

import numpy as np

from . import RemoraError
from . import _lib as L

DEFAULT_REFINE_HBW = 5  # src/remora/constants.py:42
DEFAULT_REFINE_SHORT_DWELL_PARAMS = (4, 3, 0.5)  # :233
REFINE_ALGO_VIT_NAME = "Viterbi"
REFINE_ALGO_DWELL_PEN_NAME = "dwell_penalty"
REFINE_ALGOS = (REFINE_ALGO_DWELL_PEN_NAME, REFINE_ALGO_VIT_NAME)
DEFAULT_REFINE_ALGO = REFINE_ALGO_DWELL_PEN_NAME
ROUGH_RESCALE_LEAST_SQUARES = "least_squares"
ROUGH_RESCALE_THEIL_SEN = "theil_sen"
ROUGH_RESCALE_METHODS = (ROUGH_RESCALE_LEAST_SQUARES, ROUGH_RESCALE_THEIL_SEN)
DEFAULT_ROUGH_RESCALE_METHOD = ROUGH_RESCALE_LEAST_SQUARES
MAX_POINTS_FOR_THEIL_SEN = 1000


def compute_dwell_pen_array(target, limit, weight):
    """Penalty of a dwell of 0..limit-1 samples, weight * (dwell - target)^2 (:33-40)."""
    limit = min(limit, target)
    return weight * np.square(np.arange(limit, dtype=np.float32) - target)


DEFAULT_REFINE_SHORT_DWELL_PEN = compute_dwell_pen_array(*DEFAULT_REFINE_SHORT_DWELL_PARAMS)


def index_from_kmer(kmer, alphabet="ACGT"):
    """Big-endian base-len(alphabet) index of a k-mer string (:127-146)."""
    idx = 0
    for base in kmer:
        idx = idx * len(alphabet) + alphabet.find(base)
    return idx


# ---- re-scaling estimators (float64 numpy, :55-118) -----------------------------------------


def _sorted_quantile(a, q):
    a = np.sort(np.asarray(a))
    n = a.size
    vi = (n - 1) * q
    prev = np.floor(vi)
    gamma = vi - prev
    pi = prev.astype(np.intp)
    ni = pi + 1
    top = vi >= n - 1
    pi[top] = -1
    ni[top] = -1
    lo, hi = a[pi], a[ni]
    diff = hi - lo
    out = lo + diff * gamma
    m = gamma >= 0.5
    out[m] = (hi - diff * (1 - gamma))[m]
    return out


_QUANTILE_OK = None


def _quantile(a, q):
    """np.quantile(a, q) (method 'linear') for a 1-D float array, computed from one sort with numpy's
    own interpolation arithmetic: bit-identical results at a fifth of the cost (np.quantile spends its
    time in a multi-pivot partition and Python dispatch).  Checked once per process against np.quantile;
    any difference (another numpy) switches back to np.quantile."""
    global _QUANTILE_OK
    if _QUANTILE_OK is None:
        rng = np.random.default_rng(1234)
        ok = True
        for n, dt in ((1, np.float64), (2, np.float32), (37, np.float64), (1000, np.float32), (4097, np.float64)):
            v = np.round(rng.normal(0, 1, n), 2).astype(dt)
            qq = np.arange(0.05, 1, 0.05)
            r = np.quantile(v, qq)
            f = _sorted_quantile(v, qq)
            ok = ok and r.dtype == f.dtype and np.array_equal(r, f)
        _QUANTILE_OK = ok
    if _QUANTILE_OK and a.ndim == 1 and a.size > 0 and a.dtype.kind == "f" and not np.isnan(a).any():
        return _sorted_quantile(a, q)
    return np.quantile(a, q)


def _fit_line(x, y):
    return np.linalg.lstsq(np.column_stack([np.ones_like(x), x]), y, rcond=None)[0]


_BATCHED_LSTSQ_OK = None


def _fit_lines(X, Y):
    """`_fit_line` for every row of X, Y ([n, m] float64): (intercept, slope) [n, 2], bit for bit what the per-row
    np.linalg.lstsq calls return.  numpy's lstsq is a thin wrapper around a LAPACK gufunc that broadcasts over leading
    dimensions; calling it once on the stacked systems removes 16 us of Python per read (32 of the 106 ms a batch of
    2048 reads with a refiner took).  The gufunc is private to numpy, so it is checked against the public function
    on the first rows of the first call and abandoned for good if it ever disagrees."""
    global _BATCHED_LSTSQ_OK
    X, Y = np.asarray(X, np.float64), np.asarray(Y, np.float64)
    n, m = X.shape
    per_row = lambda lo: np.asarray([_fit_line(X[i], Y[i]) for i in range(lo, n)]).reshape(-1, 2)  # noqa: E731
    if _BATCHED_LSTSQ_OK is False or n == 0:
        return per_row(0)
    try:
        from numpy.linalg import _umath_linalg

        A = np.stack([np.ones_like(X), X], axis=-1)
        sol = _umath_linalg.lstsq(A, Y[..., None], np.finfo(np.float64).eps * max(m, 2), signature="ddd->ddid")[0][:, :2, 0]
        if _BATCHED_LSTSQ_OK is None:
            k = min(n, 4)
            _BATCHED_LSTSQ_OK = bool(np.array_equal(sol[:k], np.asarray([_fit_line(X[i], Y[i]) for i in range(k)])))
        if _BATCHED_LSTSQ_OK:
            return sol
    except Exception:  # noqa: BLE001 - a numpy without this private entry point: the public path is always there
        _BATCHED_LSTSQ_OK = False
    return per_row(0)


def rescale_lstsq(dacs, levels, shift, scale):
    inter, slope = _fit_line((dacs - shift) / scale, levels)
    if slope == 0:
        return shift, scale
    return shift - (scale * inter / slope), scale / slope


def rough_rescale_lstsq(dacs, levels, shift, scale, quants):
    inter, slope = _fit_line(_quantile((dacs - shift) / scale, quants), _quantile(levels, quants))
    if slope == 0:
        return shift, scale
    return shift - (scale * inter / slope), scale / slope


def theil_sen(norm_sig, levels, shift, scale):
    """Median of pairwise slopes over pairs with increasing x, median intercept (:86-107)."""
    dx = norm_sig[:, np.newaxis] - norm_sig
    dy = levels[:, np.newaxis] - levels
    keep = dx > 0
    slope = np.median(dy[keep] / dx[keep])
    inter = np.median(levels - slope * norm_sig)
    if slope == 0:
        raise RemoraError("Theil-Sen slope is zero: cannot re-scale")
    return shift + (-inter / slope) * scale, scale * (1 / slope)


def rescale_theil_sen(dacs, levels, shift, scale, max_points=MAX_POINTS_FOR_THEIL_SEN):
    norm_sig = (dacs - shift) / scale
    if levels.shape[0] > max_points:
        samp = np.random.choice(levels.shape[0], max_points, replace=False)
        levels, norm_sig = levels[samp], norm_sig[samp]
    return theil_sen(norm_sig, levels, shift, scale)


def rough_rescale_theil_sen(dacs, levels, shift, scale, quants):
    return theil_sen(_quantile((dacs - shift) / scale, quants), _quantile(levels, quants), shift, scale)


class _RefineDesc(ctypes.Structure):
    _fields_ = [("kmer_levels", ctypes.c_void_p), ("kmer_len", ctypes.c_int32), ("center_idx", ctypes.c_int32),
                ("sd_arr", ctypes.c_void_p), ("sd_len", ctypes.c_int32), ("algo", ctypes.c_int32),
                ("half_bandwidth", ctypes.c_int32), ("min_step", ctypes.c_int32)]


class _DeviceRefiner:
    """rmr_refiner handle (k-mer table + penalties resident on one GPU)."""

    def __init__(self, engine, levels, center_idx, sd_arr, algo, hbw, min_step=2):
        self._lib = L.lib()
        self.engine = engine
        levels = np.ascontiguousarray(levels, np.float32)
        sd = None if sd_arr is None else np.ascontiguousarray(sd_arr, np.float32)
        if algo not in REFINE_ALGOS:
            raise RemoraError(f"Invalid refine algorithm: {algo}")
        desc = _RefineDesc(levels.ctypes.data, int(round(np.log(levels.size) / np.log(4))), int(center_idx),
                           None if sd is None else sd.ctypes.data, 0 if sd is None else int(sd.size),
                           0 if algo == REFINE_ALGO_VIT_NAME else 1, int(hbw), int(min_step))
        h = ctypes.c_void_p()
        L.check(self._lib.rmr_refiner_create(engine.handle, ctypes.byref(desc), ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.rmr_refiner_destroy(h)
            except Exception:
                pass

    def refine(self, dacs, sig_off, maps, int_seq, seq_off, shift, scale):
        """Concatenated host arrays -> (out maps i64 like `maps`, status i32[n_reads])."""
        n = sig_off.size - 1
        out = maps.copy()
        status = np.zeros(n, np.int32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        L.check(self._lib.rmr_refine_signal_maps(self._h, n, p(dacs), p(sig_off), p(maps), p(int_seq), p(seq_off),
                                                  p(shift), p(scale), p(out), p(status), L.MEM_HOST))
        return out, status

    def status_message(self, status):
        return self._lib.rmr_refine_status_message(int(status)).decode()


@dataclasses.dataclass
class SigMapRefiner:
    """Same fields, defaults and methods as the reference dataclass (:150-175)."""

    kmer_model_filename: str = None
    do_rough_rescale: bool = False
    scale_iters: int = -1
    algo: str = DEFAULT_REFINE_ALGO
    half_bandwidth: int = DEFAULT_REFINE_HBW
    sd_params: tuple = None
    do_fix_guage: bool = False
    rough_rescale_method: str = DEFAULT_ROUGH_RESCALE_METHOD
    sd_arr: np.ndarray = dataclasses.field(default_factory=lambda: DEFAULT_REFINE_SHORT_DWELL_PEN)
    _levels_array: np.ndarray = None
    str_kmer_levels: dict = None
    kmer_len: int = None
    kmer_idx_stats = None
    center_idx: int = -1
    is_loaded: bool = False

    def __post_init__(self):
        self._dev = {}
        if self._levels_array is not None and not np.array_equal(self._levels_array, np.array(None)):
            self.is_loaded = True
            self.kmer_len = int(np.log(self._levels_array.size) / np.log(4))
            if 4**self.kmer_len != self._levels_array.size:
                raise RemoraError("k-mer level array size is not a power of 4")
        elif self.kmer_model_filename is not None:
            self.load_kmer_table()
            self.is_loaded = True
            self.determine_dominant_pos()
            if self.do_fix_guage:
                self.fix_gauge()
        elif self.str_kmer_levels is not None:
            self.is_loaded = True
            self.determine_dominant_pos()
            if self.do_fix_guage:
                self.fix_gauge()
        if not self.is_loaded and (self.do_rough_rescale or self.scale_iters >= 0):
            raise RemoraError(
                "Signal re-scaling is requested without levels table. "
                f"is_loaded: {self.is_loaded} do_rough_rescale: {self.do_rough_rescale} "
                f"scale_iters: {self.scale_iters}")
        if self.sd_params is not None:
            self.sd_arr = compute_dwell_pen_array(*self.sd_params)
        if self.rough_rescale_method not in ROUGH_RESCALE_METHODS:
            raise RemoraError(f"Invalid rough re-scale method: {self.rough_rescale_method}")

    def __repr__(self):
        if not self.is_loaded:
            return "No Remora signal refine/map settings loaded"
        txt = f"Loaded {self.kmer_len}-mer table with {self.center_idx + 1} central position."
        if self.do_rough_rescale:
            txt += " Rough re-scaling will be executed."
        if self.scale_iters > 0:
            txt += (f" {self.scale_iters} rounds of signal mapping refinement followed by precise "
                    "re-scaling will be executed.")
        if self.scale_iters >= 0:
            txt += (f" Signal mapping refinement will be executed using the {self.algo} refinement "
                    f"method (band half width: {self.half_bandwidth}).")
            if self.algo == REFINE_ALGO_DWELL_PEN_NAME:
                txt += f" Short dwell penalty array set to {self.sd_arr}."
        return txt

    # ---- table handling (:197-331) ----------------------------------------------------------
    @property
    def bases_before(self):
        return self.center_idx

    @property
    def bases_after(self):
        return self.kmer_len - self.center_idx - 1

    @property
    def is_valid(self):
        if self.is_loaded:
            return self.do_rough_rescale or self.scale_iters >= 0
        return not self.do_rough_rescale and self.scale_iters < 0

    @property
    def kmers(self):
        for kmer in product("ACGT", repeat=self.kmer_len):
            yield "".join(kmer)

    def write_kmer_table(self, fh):
        for kmer in self.kmers:
            fh.write(f"{kmer}\t{self._levels_array[index_from_kmer(kmer)]}\n")

    def load_kmer_table(self):
        """`KMER<ws>level` per line; NaN levels read as 0; all 4^k k-mers required (:226-257)."""
        table = {}
        with open(self.kmer_model_filename) as fp:
            for line in fp:
                if not line.strip():
                    continue
                kmer, level = line.split()
                kmer = kmer.upper()
                if self.kmer_len is None or not table:
                    self.kmer_len = len(kmer)
                if kmer in table:
                    raise RemoraError(f"K-mer found twice in levels file '{kmer}'.")
                if len(kmer) != self.kmer_len:
                    raise RemoraError(f"K-mer lengths not all equal '{len(kmer)} != {self.kmer_len}' for {kmer}.")
                try:
                    val = float(level)
                except ValueError:
                    raise RemoraError(f"Could not convert level to float '{level}'")
                table[kmer] = 0 if np.isnan(val) else val
        if len(table) != 4**self.kmer_len:
            raise RemoraError(
                f"K-mer table contains fewer entries ({len(table)}) than expected ({4 ** self.kmer_len})")
        self.str_kmer_levels = table

    def determine_dominant_pos(self):
        """Position of the k-mer whose base best orders the levels: Kruskal-Wallis H per position over
        the rank of the level-sorted k-mers (:259-284)."""
        if self.str_kmer_levels is None:
            return
        from scipy import stats

        ranked = [kmer for _, kmer in sorted((lvl, kmer) for kmer, lvl in self.str_kmer_levels.items())]
        self.kmer_idx_stats = []
        for pos in range(self.kmer_len):
            groups = [[rank for rank, kmer in enumerate(ranked) if kmer[pos] == base] for base in "ACGT"]
            self.kmer_idx_stats.append(stats.kruskal(*groups)[0])
        self.center_idx = np.argmax(self.kmer_idx_stats)

    def fix_gauge(self):
        """Median / MAD normalisation of the table (:340-349)."""
        lv = self.levels_array
        med = np.median(lv)
        mad = np.median(np.absolute(lv - med)) * 1.4826
        self._levels_array = (lv - med) / mad
        self.str_kmer_levels = {k: self._levels_array[index_from_kmer(k)] for k in self.kmers}
        self._dev = {}

    @property
    def levels_array(self):
        if self._levels_array is None:
            if self.str_kmer_levels is None:
                return None
            arr = np.empty(4**self.kmer_len, dtype=np.float32)
            for kmer, level in self.str_kmer_levels.items():
                arr[index_from_kmer(kmer)] = level
            self._levels_array = arr
        return self._levels_array

    def extract_levels(self, int_seq):
        """Level of every base from the k-mer around it; bases without a full k-mer get 0
        (refine_signal_map_core.pyx:87-101).  Host numpy (used by the scalar re-scaling
        estimators; the DP kernels compute their own copy on the device)."""
        seq = np.asarray(int_seq).astype(np.int64)
        n, k = seq.size, self.kmer_len
        levels = np.zeros(n, dtype=np.float32)
        if n >= k:
            idx = np.zeros(n - k + 1, dtype=np.int64)
            for j in range(k):
                idx = idx * 4 + seq[j : n - k + 1 + j]
            levels[self.center_idx : self.center_idx + n - k + 1] = np.asarray(self.levels_array, np.float32)[idx]
        return levels

    # ---- scalar re-scaling (host) -----------------------------------------------------------
    def rough_rescale(self, shift, scale, seq_to_sig_map, int_seq, dacs, quants=np.arange(0.05, 1, 0.05),
                      clip_bases=10, use_base_center=True):
        """Quantile match of the level of each base against its central sample (:366-408)."""
        levels = self.extract_levels(int_seq)
        if use_base_center:
            optim_dacs = dacs[(seq_to_sig_map[:-1] + seq_to_sig_map[1:]) // 2]
            if clip_bases > 0 and levels.size > clip_bases * 2:
                levels, optim_dacs = levels[clip_bases:-clip_bases], optim_dacs[clip_bases:-clip_bases]
        else:
            optim_dacs = dacs[seq_to_sig_map[0] : seq_to_sig_map[-1]]
        if self.rough_rescale_method == ROUGH_RESCALE_LEAST_SQUARES:
            return rough_rescale_lstsq(optim_dacs, levels, shift, scale, quants)
        if self.rough_rescale_method == ROUGH_RESCALE_THEIL_SEN:
            return rough_rescale_theil_sen(optim_dacs, levels, shift, scale, quants)
        raise RemoraError(f"Invalid rough re-scale method: {self.rough_rescale_method}")

    def rescale(self, levels, dacs, shift, scale, seq_to_sig_map, dwell_filter_pctls=(10, 90), min_abs_level=0.2,
                edge_filter_bases=10, min_levels=10):
        """Theil-Sen fit of per-base mean signal against levels over well-mapped bases (:410-470)."""
        with np.errstate(invalid="ignore", divide="ignore"):
            csum = np.empty(dacs.size + 1)
            csum[0] = 0
            csum[1:] = np.cumsum(dacs)
            dwells = np.diff(seq_to_sig_map)
            dac_means = np.diff(csum[seq_to_sig_map]) / dwells
        dwell_min, dwell_max = np.percentile(dwells, dwell_filter_pctls)
        inner = np.full(dwells.size, True, dtype=np.bool_)
        if edge_filter_bases > 0:
            inner[:edge_filter_bases] = False
            inner[-edge_filter_bases:] = False
        valid = np.logical_and.reduce((dwells > dwell_min, dwells < dwell_max,
                                       np.abs(levels - np.mean(levels)) > min_abs_level,
                                       np.logical_not(np.isnan(dac_means)), inner))
        if int(valid.sum()) < min_levels:
            raise RemoraError("Too few positions")
        return rescale_theil_sen(dac_means[valid], levels[valid], shift, scale)

    # ---- the DP on the GPU --------------------------------------------------------------------
    def _device_refiner(self, device=None):
        from .engine import get_engine

        eng = get_engine(device)
        dev = self._dev.get(eng.device)
        if dev is None:
            dev = _DeviceRefiner(eng, self.levels_array, self.center_idx, self.sd_arr, self.algo, self.half_bandwidth)
            self._dev[eng.device] = dev
        return dev

    def refine_maps(self, dacs_list, shifts, scales, maps, int_seqs, device=None):
        """One banded-DP pass for many reads.  Per read: dacs (int16), seq_to_sig_map, int_seq.
        Returns the list of refined maps; raises RemoraError for the first read whose band is
        invalid (validate_band, :686-737) unless `errors` are wanted per read (see refine_reads)."""
        outs, status, dev = self._refine_batch(dacs_list, shifts, scales, maps, int_seqs, device)
        for st in status:
            if st != 0:
                raise RemoraError(dev.status_message(st))
        return outs

    def _refine_batch(self, dacs_list, shifts, scales, maps, int_seqs, device=None):
        n = len(dacs_list)
        dev = self._device_refiner(device)
        if n == 0:
            return [], np.zeros(0, np.int32), dev
        sig_off = np.zeros(n + 1, np.int64)
        seq_off = np.zeros(n + 1, np.int64)
        np.cumsum([d.size for d in dacs_list], out=sig_off[1:])
        np.cumsum([s.size for s in int_seqs], out=seq_off[1:])
        for m, s in zip(maps, int_seqs):
            if m.size != s.size + 1:
                raise RemoraError("seq_to_sig_map must have one more entry than bases")
        dacs = np.ascontiguousarray(np.concatenate(dacs_list), np.int16)
        cmap = np.ascontiguousarray(np.concatenate(maps), np.int64)
        cseq = np.ascontiguousarray(np.concatenate(int_seqs), np.int8)
        out, status = dev.refine(dacs, sig_off, cmap, cseq, seq_off, np.ascontiguousarray(shifts, np.float64),
                                 np.ascontiguousarray(scales, np.float64))
        mo = seq_off + np.arange(n + 1)
        return [out[mo[i] : mo[i + 1]] for i in range(n)], status, dev

    def refine_sig_map(self, shift, scale, seq_to_sig_map, int_seq, dacs):
        """Per-read entry with the reference's signature and return value (:472-497)."""
        levels = None
        s2s = np.asarray(seq_to_sig_map)
        for _ in range(max(1, self.scale_iters)):
            s2s = self.refine_maps([dacs], [shift], [scale], [s2s], [int_seq])[0].astype(s2s.dtype, copy=False)
            if self.scale_iters > 0:
                if levels is None:
                    levels = self.extract_levels(int_seq)
                st = s2s[0]
                try:
                    shift, scale = self.rescale(levels, dacs[st : s2s[-1]], shift, scale, s2s - st)
                except RemoraError:
                    break
        return s2s, shift, scale

    def refine_reads(self, reads, check_read=False):
        """Batched RemoraRead.refine_signal_mapping (src/remora/data_chunks.py:267-308): scalar
        re-scaling per read on the host, one GPU call per DP round for all reads.  Mutates the
        reads; returns a list with None or the RemoraError of each read."""
        errs = [None] * len(reads)
        if not self.is_loaded:
            return errs
        if self.do_rough_rescale:
            for r in reads:
                r.shift, r.scale = self.rough_rescale(r.shift, r.scale, r.seq_to_sig_map, r.int_seq, r.dacs)
                r._sig = None
        if self.scale_iters >= 0:
            live = list(range(len(reads)))
            levels = {}
            for _ in range(max(1, self.scale_iters)):
                if not live:
                    break
                sub = [reads[i] for i in live]
                outs, status, dev = self._refine_batch([r.dacs for r in sub], [r.shift for r in sub],
                                                       [r.scale for r in sub], [np.asarray(r.seq_to_sig_map) for r in sub],
                                                       [r.int_seq for r in sub])
                nxt = []
                for i, r, o, st in zip(live, sub, outs, status):
                    if st != 0:
                        errs[i] = RemoraError(dev.status_message(st))
                        continue
                    r.seq_to_sig_map = o.astype(np.asarray(r.seq_to_sig_map).dtype, copy=False)
                    r._sig = None
                    if self.scale_iters > 0:
                        if i not in levels:
                            levels[i] = self.extract_levels(r.int_seq)
                        s0 = r.seq_to_sig_map[0]
                        try:
                            r.shift, r.scale = self.rescale(levels[i], r.dacs[s0 : r.seq_to_sig_map[-1]], r.shift,
                                                            r.scale, r.seq_to_sig_map - s0)
                        except RemoraError:
                            continue
                    nxt.append(i)
                live = nxt
        if check_read:
            for i, r in enumerate(reads):
                if errs[i] is None:
                    try:
                        r.check()
                    except RemoraError as e:
                        errs[i] = e
        return errs

    def rough_rescale_device(self, dr, reads, quants=np.arange(0.05, 1, 0.05), clip_bases=10):
        """`rough_rescale` for all reads of a `DeviceReads` batch: the per-base level lookup, the centre-sample
        gather, the float64 normalisation, the two sorts and numpy's quantile interpolation run in one hand-written
        kernel (one block per read, `rmr_rescale_quantiles`, csrc/k_refine.hip), the 19-point line fit of every
        read stays LAPACK / numpy on the host - so the new (shift, scale) equal the per-read host method bit for
        bit.  Updates the reads and `dr`."""
        sig_q, lvl_q = self._device_quantiles(dr, np.asarray(quants, np.float64), int(clip_bases))
        shifts, scales = [], []
        fits = _fit_lines(sig_q, lvl_q) if self.rough_rescale_method == ROUGH_RESCALE_LEAST_SQUARES else None
        for i, r in enumerate(reads):
            if fits is not None:
                inter, slope = fits[i]
                sh, sc = (r.shift, r.scale) if slope == 0 else (r.shift - (r.scale * inter / slope), r.scale / slope)
            else:
                sh, sc = theil_sen(sig_q[i], lvl_q[i], r.shift, r.scale)
            r.shift, r.scale = sh, sc
            r._sig = None
            shifts.append(float(sh))
            scales.append(float(sc))
        dr.set_scaling(shifts, scales)

    def _device_quantiles(self, dr, quants, clip_bases):
        """(sig_q, lvl_q) f64[n_reads][len(quants)] of a resident batch.  Reads longer than the kernel's in-LDS sort
        holds (16384 kept bases) send the batch through `_device_quantiles_general`."""
        import torch

        nr = dr.n_reads
        longest = int(np.diff(dr.seq_off).max()) if nr else 0
        if nr == 0 or longest < 1 or os.environ.get("RMR_RESCALE_GENERAL") == "1":
            return self._device_quantiles_general(dr, quants, clip_bases)
        dev = self._device_refiner(dr.engine.device)
        tdev = dr.s2s.device
        sig_q = torch.empty((nr, quants.size), dtype=torch.float64, device=tdev)
        lvl_q = torch.empty_like(sig_q)
        status = torch.empty(nr, dtype=torch.int32, device=tdev)
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        q = np.ascontiguousarray(quants, np.float64)
        L.check(dev._lib.rmr_rescale_quantiles(dev._h, nr, p(dr.dacs), p(dr.d_sig_off), p(dr.s2s), p(dr.iseq), p(dr.d_seq_off),
                                                p(dr.shift), p(dr.scale), longest, clip_bases, q.size,
                                                q.ctypes.data_as(ctypes.c_void_p), p(sig_q), p(lvl_q), p(status)))
        if status.cpu().numpy().any():
            return self._device_quantiles_general(dr, quants, clip_bases)
        return sig_q.cpu().numpy(), lvl_q.cpu().numpy()

    def _device_quantiles_general(self, dr, quants, clip_bases):
        """The same quantiles with torch tensor ops on the resident arrays (float64 / float32 exactly as numpy
        evaluates them): no bound on the read length; used for batches holding a read the kernel cannot sort in LDS."""
        import torch

        dev = dr.s2s.device
        nr = dr.n_reads
        seq_off = dr.d_seq_off
        n_i = seq_off[1:] - seq_off[:-1]
        total = int(dr.seq_off[-1])
        k = self.kmer_len
        base = torch.arange(total, device=dev)
        read_of = torch.searchsorted(seq_off, base, right=True) - 1
        local = base - seq_off[read_of]
        # levels (refine_signal_map_core.pyx:87-101): k-mer starting at local - center_idx, inside the read
        pos = base - int(self.center_idx)
        ok = (local >= int(self.center_idx)) & (local - int(self.center_idx) + k <= n_i[read_of])
        idx = torch.zeros(total, dtype=torch.int64, device=dev)
        seq = dr.iseq.to(torch.int64)
        for j in range(k):
            idx = idx * 4 + seq[(pos + j).clamp(0, total - 1)]
        table = torch.from_numpy(np.ascontiguousarray(self.levels_array, np.float32)).to(dev)
        levels = torch.where(ok, table[idx.clamp(0, table.numel() - 1)], torch.zeros((), dtype=torch.float32, device=dev))
        # centre sample of every base, normalised in float64
        mi = base + read_of  # index of the base's start in the concatenated maps (n + 1 entries per read)
        mid = (dr.s2s[mi] + dr.s2s[mi + 1]) // 2
        od = dr.dacs[dr.d_sig_off[read_of] + mid].to(torch.float64)
        norm = (od - dr.shift[read_of]) / dr.scale[read_of]
        # clip `clip_bases` at both ends of reads longer than 2 * clip_bases
        clip = (n_i > 2 * clip_bases)[read_of] if clip_bases > 0 else torch.zeros(total, dtype=torch.bool, device=dev)
        keep = ~clip | ((local >= clip_bases) & (local < n_i[read_of] - clip_bases))
        cnt = torch.where(n_i > 2 * clip_bases, n_i - 2 * clip_bases, n_i) if clip_bases > 0 else n_i
        col = torch.where(clip, local - clip_bases, local)
        width = int(cnt.max().item())
        q = torch.from_numpy(np.asarray(quants, np.float64)).to(dev)

        def quantiles(vals, dtype):
            mat = torch.full((nr, width), float("inf"), dtype=dtype, device=dev)
            mat[read_of[keep], col[keep]] = vals[keep]
            srt = torch.sort(mat, dim=1).values
            vi = (cnt - 1).to(torch.float64)[:, None] * q[None, :]
            prev = torch.floor(vi)
            gamma = vi - prev
            pi = prev.to(torch.int64)
            ni = pi + 1
            top = vi >= (cnt - 1).to(torch.float64)[:, None]
            last = (cnt - 1)[:, None].expand_as(pi)
            pi = torch.where(top, last, pi)
            ni = torch.where(top, last, ni)
            lo, hi = torch.gather(srt, 1, pi), torch.gather(srt, 1, ni)
            diff = hi - lo  # in the array's own dtype, as numpy's subtract(b, a)
            out = lo.to(torch.float64) + diff.to(torch.float64) * gamma
            alt = hi.to(torch.float64) - diff.to(torch.float64) * (1 - gamma)
            return torch.where(gamma >= 0.5, alt, out).cpu().numpy()

        return quantiles(norm, torch.float64), quantiles(levels, torch.float32)

    def refine_device_reads(self, dr, reads):
        """One DP pass (scale_iters 0 or 1 round of it) on reads that are already resident (`DeviceReads`):
        the refined mappings replace `dr.s2s` on the device and are copied back into the read objects.
        Raises the RemoraError of the first read whose band is invalid."""
        import torch

        dev = self._device_refiner(dr.engine.device)
        out = torch.empty_like(dr.s2s)
        status = torch.empty(max(dr.n_reads, 1), dtype=torch.int32, device=dr.s2s.device)  # written for every read
        p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
        L.check(dev._lib.rmr_refine_signal_maps(dev._h, dr.n_reads, p(dr.dacs), p(dr.d_sig_off), p(dr.s2s), p(dr.iseq),
                                                 p(dr.d_seq_off), p(dr.shift), p(dr.scale), p(out), p(status), L.MEM_DEVICE))
        for st in status[: dr.n_reads].cpu().numpy():
            if st != 0:
                raise RemoraError(dev.status_message(st))
        from .data_chunks import device_to_numpy

        dr.s2s = out
        host = device_to_numpy(out)
        mo = dr.seq_off + np.arange(dr.n_reads + 1)
        for i, r in enumerate(reads):
            r.seq_to_sig_map = host[mo[i] : mo[i + 1]].astype(np.asarray(r.seq_to_sig_map).dtype, copy=False)
            r._sig = None

    # ---- (de)serialisation (:499-587) -----------------------------------------------------------
    def asdict(self):
        return {
            "refine_kmer_levels": self._levels_array,
            "refine_kmer_center_idx": self.center_idx,
            "refine_do_rough_rescale": self.do_rough_rescale,
            "refine_scale_iters": self.scale_iters,
            "refine_algo": self.algo,
            "refine_half_bandwidth": self.half_bandwidth,
            "refine_sd_arr": self.sd_arr,
            "rough_rescale_method": self.rough_rescale_method,
        }

    @classmethod
    def load_from_metadata(cls, metadata):
        return cls(
            _levels_array=metadata.get("refine_kmer_levels"),
            center_idx=metadata.get("refine_kmer_center_idx"),
            do_rough_rescale=metadata.get("refine_do_rough_rescale"),
            scale_iters=metadata.get("refine_scale_iters"),
            algo=metadata.get("refine_algo"),
            half_bandwidth=metadata.get("refine_half_bandwidth"),
            sd_arr=metadata.get("refine_sd_arr"),
            rough_rescale_method=metadata.get("rough_rescale_method", ROUGH_RESCALE_LEAST_SQUARES),
        )

    @classmethod
    def load_from_dict(cls, data, do_rough_rescale=True, scale_iters=-1, algo=DEFAULT_REFINE_ALGO,
                       half_bandwidth=DEFAULT_REFINE_HBW, sd_params=None, do_fix_guage=False,
                       sd_arr=DEFAULT_REFINE_SHORT_DWELL_PEN, rough_rescale_method=DEFAULT_ROUGH_RESCALE_METHOD):
        return cls(do_rough_rescale=do_rough_rescale, scale_iters=scale_iters, algo=algo,
                   half_bandwidth=half_bandwidth, sd_params=sd_params, do_fix_guage=do_fix_guage, sd_arr=sd_arr,
                   str_kmer_levels=data, kmer_len=len(next(iter(data.keys()))),
                   rough_rescale_method=rough_rescale_method)

    def __eq__(self, other):
        if not isinstance(other, SigMapRefiner):
            return False
        if self.do_rough_rescale != other.do_rough_rescale or self.scale_iters != other.scale_iters:
            return False
        if not self.do_rough_rescale and self.scale_iters < 0:
            return True
        if self.rough_rescale_method != other.rough_rescale_method:
            return False
        if not np.array_equal(self._levels_array, other._levels_array) or self.center_idx != other.center_idx:
            return False
        if self.scale_iters < 0:
            return True
        return (self.algo == other.algo and self.half_bandwidth == other.half_bandwidth
                and np.array_equal(self.sd_arr, other.sd_arr))
