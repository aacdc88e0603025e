"""Python API of the hot path — drop-in for `remora.inference.call_read_mods`
(src/remora/inference.py:661-712) and the per-label tally the multi-GPU runs reduce."""
import numpy as np

from . import RemoraError
from .constants import DEFAULT_BATCH_SIZE
from .util import Motif, format_mm_ml_tags, softmax_axis1


def call_read_mods(read, model, model_metadata, batch_size=DEFAULT_BATCH_SIZE, focus_offset=None,
                   return_mm_ml_tags=False, return_mod_probs=False):
    """Call modified bases on one read; arguments and return values as in the reference:
    (nn_out f32[N,num_out], labels i64[N], pos i64[N]) by default;
    (probs f64[N,num_mods], labels, pos) with return_mod_probs; (MM str, ML array('B')) with
    return_mm_ml_tags; three empty arrays when the read yields no chunk (:698-699)."""
    if focus_offset is None:
        read.set_motif_focus_bases([Motif(*m) for m in model_metadata["motifs"]])
    else:
        read.focus_bases = np.array([focus_offset])
    read.prepare_batches(model_metadata, batch_size)
    if len(read.batches) == 0:
        return np.array([]), np.array([]), np.array([])
    nn_out, labels, pos = read.run_model(model)
    if not return_mod_probs and not return_mm_ml_tags:
        return nn_out, labels, pos
    probs = softmax_axis1(nn_out)[:, 1:].astype(np.float64)
    if return_mm_ml_tags:
        return format_mm_ml_tags(seq=read.str_seq, poss=pos, probs=probs, mod_bases=model_metadata["mod_bases"],
                                 can_base=model_metadata["can_base"])
    return probs, labels, pos
