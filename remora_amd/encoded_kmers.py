"""GPU twin of remora.encoded_kmers (src/remora/encoded_kmers.pyx)."""
import numpy as np

from . import RemoraError
from . import _lib as L
from .engine import get_engine, _torch


def compute_encoded_kmer_batch(before_context_bases, after_context_bases, seqs, seq_mappings, seq_lens,
                               engine=None):
    """Same call as the reference's Cython function (src/remora/encoded_kmers.pyx:13-45):
    returns float32 [nchunks, 4*kmer_len, sig_len] with sig_len = seq_mappings[0, seq_lens[0]]
    (:23).  numpy in -> numpy out (staged through the GPU); CUDA tensors in -> CUDA tensor out."""
    torch = _torch()
    kb, ka = int(before_context_bases), int(after_context_bases)
    eng = engine if engine is not None else get_engine()
    on_dev = isinstance(seqs, torch.Tensor) and seqs.is_cuda
    if on_dev:
        seqs = seqs.to(torch.int8).contiguous()
        maps = seq_mappings.to(torch.int16).contiguous()
        lens = seq_lens.to(torch.int16).contiguous()
        n = int(lens.shape[0])
        if n == 0:
            raise RemoraError("empty batch")
        sig_len = int(maps[0, int(lens[0])])
        out = torch.empty((n, 4 * (kb + ka + 1), sig_len), dtype=torch.float32, device=seqs.device)
        L.check(L.lib().rmr_encode_kmers(eng.handle, kb, ka, seqs.data_ptr(), seqs.shape[1], maps.data_ptr(),
                                         maps.shape[1], lens.data_ptr(), n, sig_len, out.data_ptr(), L.MEM_DEVICE))
        return out
    seqs = np.ascontiguousarray(seqs, np.int8)
    maps = np.ascontiguousarray(seq_mappings, np.int16)
    lens = np.ascontiguousarray(seq_lens, np.int16)
    n = lens.size
    if n == 0:
        raise RemoraError("empty batch")
    sig_len = int(maps[0, lens[0]])
    out = np.empty((n, 4 * (kb + ka + 1), sig_len), np.float32)
    L.check(L.lib().rmr_encode_kmers(eng.handle, kb, ka, seqs.ctypes.data, seqs.shape[1], maps.ctypes.data,
                                     maps.shape[1], lens.ctypes.data, n, sig_len, out.ctypes.data, L.MEM_HOST))
    return out
