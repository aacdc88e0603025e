"""GPU twin of remora.data_chunks_core (src/remora/data_chunks_core.pyx)."""
import numpy as np

from . import RemoraError
from . import _lib as L
from .engine import get_engine


def trim_sb_chunk_context_core(stored_cc_before, stored_cc_after, cc_before, cc_after, total_seq_context,
                               seqs, seq_mappings, seq_lens, engine=None):
    """In-place trim, same call as the reference (src/remora/data_chunks_core.pyx:10-45); the
    caller has already shifted `seq_mappings` by the start difference
    (src/remora/data_chunks.py:1555-1563)."""
    for a, dt in ((seqs, np.int8), (seq_mappings, np.int16), (seq_lens, np.int16)):
        if not (isinstance(a, np.ndarray) and a.dtype == dt and a.flags.c_contiguous and a.flags.writeable):
            raise RemoraError("trim_sb_chunk_context_core needs writable C-contiguous int8/int16/int16 arrays")
    eng = engine if engine is not None else get_engine()
    L.check(L.lib().rmr_trim_chunk_context(
        eng.handle, int(stored_cc_before), int(stored_cc_after), int(cc_before), int(cc_after),
        int(total_seq_context), seqs.ctypes.data, seqs.shape[1], seq_mappings.ctypes.data,
        seq_mappings.shape[1], seq_lens.ctypes.data, seq_lens.size, L.MEM_HOST))
