"""Placeholder for the signal-mapping refiner object that model metadata carries
(src/remora/refine_signal_map.py:150-632).  Only the unloaded state is supported: models
without a k-mer level table get `SigMapRefiner()` whose `is_loaded` is False, which makes
`RemoraRead.refine_signal_mapping` a no-op exactly as in the reference
(src/remora/data_chunks.py:267-269).  The banded-DP refinement itself is a 'next' row."""
import dataclasses

import numpy as np


@dataclasses.dataclass
class SigMapRefiner:
    kmer_model_filename: str = None
    do_rough_rescale: bool = False
    scale_iters: int = -1
    algo: str = "dwell_penalty"
    half_bandwidth: int = 5
    sd_arr: np.ndarray = None
    do_fix_guage: bool = False
    _levels_array: np.ndarray = None
    center_idx: int = -1

    @property
    def is_loaded(self):
        return self._levels_array is not None
