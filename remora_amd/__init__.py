"""remora_amd — MI355X (gfx950) engine for the per-read modified-base-call hot path of
nanoporetech/remora: chunk extraction -> k-mer encode -> ConvLSTM_w_ref / Conv_w_ref forward.

Host code is Python and mirrors the reference's interfaces for this path
(`model_util.load_model`, `inference.call_read_mods`, `data_chunks.RemoraRead`,
`encoded_kmers.compute_encoded_kmer_batch`, `io.parse_move_tag`, ...); every numeric step runs
in hand-written HIP kernels behind the C ABI in include/remora_hip.h (libremora_hip.so).
There is no CPU fallback: without the library or without a GPU the compute entry points raise
RemoraError.
"""

__version__ = "0.1.0"


class RemoraError(Exception):
    """Same role as remora.RemoraError (src/remora/__init__.py:4-7): the one exception type
    callers of the hot path catch (src/remora/inference.py:88, src/remora/io.py:502)."""
