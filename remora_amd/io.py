"""Read-side helper of the hot path: move-table expansion (src/remora/io.py:394-407)."""
import ctypes

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
