"""`remora validate from_remora_dataset` on the MI355X path: the chunks of a (mixed) RemoraDataset through the
fused extraction-free inference kernels and the reference's validation summary on top (accuracy, confusion
matrix, cross-entropy loss, confidence-filtered accuracy).  Mirrors src/remora/validate.py:17-293 for the
dataset flavour: VAL_METRICS, mat_to_str, compute_metrics (:42-66), add_unmodeled_labels (:69-99),
ValidationLogger (:168-293).  The host part is numpy on N x num_labels arrays; the model forward is
HipModel.infer_chunks on the stored rows (no one-hot tensor is materialised)."""
import json
from collections import namedtuple

import numpy as np

from . import constants
from .engine import HipModel, _torch
from .util import softmax_axis1

VAL_METRICS = namedtuple("VAL_METRICS", ("loss", "acc", "num_calls", "conf_mat", "filt_frac", "filt_acc", "filt_conf_mat",
                                         "filt_thresh"))


def mat_to_str(mat):
    return json.dumps(np.asarray(mat).tolist(), separators=(",", ":"))


def confusion_matrix(labels, preds):
    """Counts[true, predicted] over the sorted union of the classes that occur in either vector - what
    sklearn.metrics.confusion_matrix(labels, preds) returns with default arguments (validate.py:44)."""
    labels, preds = np.asarray(labels), np.asarray(preds)
    if labels.size and min(labels.min(), preds.min()) >= 0 and max(labels.max(), preds.max()) < 4096:
        k = int(max(labels.max(), preds.max())) + 1  # small non-negative class ids: one bincount, then keep what occurs
        full = np.bincount(labels.astype(np.int64) * k + preds, minlength=k * k).reshape(k, k)
        present = (full.sum(0) + full.sum(1)) > 0
        return full[present][:, present].astype(np.int64)
    classes = np.unique(np.concatenate([labels, preds]))
    n = classes.size
    idx = np.searchsorted(classes, labels) * n + np.searchsorted(classes, preds)
    return np.bincount(idx, minlength=n * n).reshape(n, n).astype(np.int64)


def compute_metrics(probs, labels, filt_frac):
    """(acc, conf_mat, filt_frac, filt_acc, filt_conf_mat, filt_thr): overall accuracy and the accuracy over the
    calls whose winning probability lies above the `filt_frac` quantile (validate.py:42-66)."""
    preds = np.argmax(probs, axis=1)
    conf = confusion_matrix(labels, preds)
    acc = (preds == labels).sum() / labels.size
    win = np.take_along_axis(probs, preds[:, None], -1)[:, 0]
    return (acc, conf) + filtered_metrics(win, preds, labels, filt_frac)


def filtered_metrics(win, preds, labels, filt_frac):
    """(filt_frac, filt_acc, filt_conf_mat, filt_thr) of the calls whose winning probability lies above the `filt_frac`
    quantile of all winning probabilities (validate.py:47-66)."""
    thr = np.quantile(win, filt_frac)
    if thr == win.max():  # everything would be filtered: nudge the threshold below the maximum
        thr *= 0.999999
    sure = win > thr
    n_sure = int(sure.sum())
    if n_sure == 0:  # all probabilities NaN
        return 1.0, np.nan, np.array([]), np.nan
    return (1 - n_sure / labels.size, int(((preds == labels) & sure).sum()) / n_sure,
            confusion_matrix(labels[sure], preds[sure]), thr)


def add_unmodeled_labels(output, unmodeled_labels):
    """Widen model outputs by the label columns the dataset has but the model does not; those columns get -1000
    so that they vanish under softmax (validate.py:69-99)."""
    unmodeled_labels = np.asarray(unmodeled_labels)
    if unmodeled_labels.size == 0:
        return output
    nobs, nlab = output.shape
    width = nlab + unmodeled_labels.size
    wide = np.full((nobs, width), -1000, dtype=output.dtype)
    wide[:, 0] = output[:, 0]
    skipped = 0
    for col in range(1, width):
        if col in unmodeled_labels:
            skipped += 1
        else:
            wide[:, col] = output[:, col - skipped]
    return wide


_RAW = ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths", "labels")


class ValidationLogger:
    """Tab-separated validation summary lines, same columns as the reference (validate.py:168-293)."""

    HEADER = "\t".join(("Val_Type", "Epoch", "Iteration", "Accuracy", "Confusion_Matrix", "Loss", "Num_Calls",
                        "Filtered_Fraction", "Filtered_Accuracy", "Filtered_Confusion_Matrix", "Filtered_Threshold"))
    FULL_HEADER = "\t".join(["label", "class_pred", "class_probs"])

    def __init__(self, fp, full_results_fh=None):
        self.fp = fp
        self.fp.write(self.HEADER + "\n")
        self.full_fh = full_results_fh
        if self.full_fh is not None:
            self.full_fh.write(self.FULL_HEADER + "\n")

    def write_full_results(self, output, labels):
        probs = softmax_axis1(output)
        for lab, pred, row in zip(labels.tolist(), output.argmax(axis=1), probs):
            self.full_fh.write(f"{lab}\t{pred}\t{','.join(map(str, row))}\n")

    def run_validation(self, model, model_mod_bases, criterion, dataset, filt_frac=constants.DEFAULT_FILT_FRAC,
                       full_results_fh=None, disable_pbar=False, world=1):
        """All batches of `dataset` (a finite RemoraDataset) through the model -> VAL_METRICS.  `criterion` is
        a torch loss on (float32 logits, int64 labels), or None for cross entropy; the loss is the mean of the
        per-batch means, as in the reference.

        `world` > 1 (one process per GPU, torch.distributed up, `dataset` = this rank's `RemoraDataset.shard`): the
        confusion matrix is tallied per rank over the full label set and summed by the ONE collective of the path
        (dist.allreduce_counts -> RCCL all-reduce of int64[k*k]); accuracy and num_calls follow from it.  The
        confidence-filtered columns need the global quantile of the winning probabilities, so the per-chunk (label,
        call, winning probability) triples are gathered as well (12 B per chunk, bookkeeping), and the loss is the
        mean over all ranks' batches.  Every rank returns the global metrics."""
        torch = _torch()
        if criterion is None:
            criterion = torch.nn.CrossEntropyLoss()
        md = dataset.metadata
        unmodeled = np.array([i + 1 for i, mb in enumerate(md.mod_bases) if mb not in model_mod_bases])
        fused = isinstance(model, HipModel)
        if fused and self.full_fh is None and type(criterion) is torch.nn.CrossEntropyLoss and criterion.weight is None \
                and criterion.reduction == "mean" and criterion.label_smoothing == 0.0 and criterion.ignore_index == -100 \
                and md.num_labels <= 16:
            return self._run_validation_device(model, unmodeled, dataset, filt_frac, world)
        names = _RAW if fused else ("enc_kmers", "signal", "labels")
        dataset._ds_iters = None
        all_out, all_lab, losses = [], [], []
        for batch in dataset.iter_numpy_batches(return_arrays=names, copy=not fused):
            b = dict(zip(names, batch))
            labels = np.asarray(b["labels"])
            if labels.size == 0:  # trailing empty batch of a short last super batch (the reference's loss turns NaN on it)
                continue
            if fused:  # stored rows straight into the fused kernels (uploaded from the memmaps when untouched)
                out = model.infer_chunks(b["signal"], b["sequence"], b["sequence_to_signal_mapping"], b["sequence_lengths"],
                                         md.kmer_context_bases)
                out = out.cpu().numpy() if hasattr(out, "cpu") else np.asarray(out)
            else:
                device = next(model.parameters()).device
                with torch.no_grad():
                    out = model(torch.from_numpy(b["signal"]).to(device), torch.from_numpy(b["enc_kmers"]).to(device))
                out = out.detach().cpu().numpy()
            out = add_unmodeled_labels(out, unmodeled)
            all_out.append(out)
            all_lab.append(labels)
            losses.append(criterion(torch.from_numpy(out), torch.from_numpy(np.array(labels))).detach().cpu().numpy())
            if self.full_fh is not None:
                self.write_full_results(out, labels)
        dataset._ds_iters = None
        out, labels = np.concatenate(all_out, axis=0), np.concatenate(all_lab)
        if world > 1:
            return self._global_metrics(softmax_axis1(out), labels, losses, filt_frac)
        acc, conf, ff, facc, fconf, thr = compute_metrics(softmax_axis1(out), labels, filt_frac)
        return VAL_METRICS(loss=np.mean(losses), acc=acc, num_calls=labels.size, conf_mat=conf, filt_frac=ff,
                           filt_acc=facc, filt_conf_mat=fconf, filt_thresh=thr)

    def _run_validation_device(self, model, unmodeled, dataset, filt_frac, world):
        """run_validation with the per-chunk work on the GPU (the fused model, the default loss, no per-chunk results file).
        A reader thread copies the batches' rows from the memmaps into a ring of pinned host slots (several copy threads: the
        page faults of a freshly mapped file are the slowest part of the host side) while the GPU works on the previous
        batch; the rows are uploaded on the engine's stream, inferred (fused kernels, logits stay on the device) and tallied
        there (rmr_validation_tally: widening by the unmodelled labels, float32 softmax, call, confusion counts, cross
        entropy); per chunk only the winning probability (4 B) and the call (1 B) come back, for the quantile of the
        confidence-filtered columns.  Same numbers as the host path (tests/test_gpu_parity.py)."""
        import ctypes
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor

        from . import _lib as L
        from .util import effective_cpu_count

        torch = _torch()
        md = dataset.metadata
        eng, dev = model.engine, model.engine.torch_device
        kf, km = md.num_labels, model.num_out
        col, label_of_column = 0, []
        for c in range(kf):
            if c in set(np.asarray(unmodeled).tolist()):
                label_of_column.append(-1)
            else:
                label_of_column.append(col)
                col += 1
        if col != km:
            from . import RemoraError

            raise RemoraError(f"model has {km} outputs, the dataset's labels minus the unmodelled ones {col}")
        colmap = (ctypes.c_int32 * kf)(*label_of_column)
        conf = torch.zeros(kf * kf, dtype=torch.int64, device=dev)
        loss_buf = torch.zeros(4096, dtype=torch.float64, device=dev)
        lib = L.lib()

        import os

        depth = 3
        nthreads = max(2, min(8, effective_cpu_count() // 2))
        free, ready = queue.Queue(), queue.Queue(maxsize=depth)
        slots = [None] * depth
        for i in range(depth):
            free.put(i)
        pool = ThreadPoolExecutor(max_workers=nthreads)
        dataset._ds_iters = None

        stop = threading.Event()  # set by the consumer when it leaves early: the reader must not stay parked on a queue

        def take_free():
            while not stop.is_set():
                try:
                    return free.get(timeout=0.1)
                except queue.Empty:
                    continue
            return None

        def hand_over(item):
            while not stop.is_set():
                try:
                    return ready.put(item, timeout=0.1)
                except queue.Full:
                    continue

        def produce():
            try:
                for batch in dataset.iter_numpy_batches(return_arrays=_RAW, copy=False):
                    n = int(np.asarray(batch[4]).shape[0])
                    if n == 0:
                        continue
                    i = take_free()
                    if i is None:
                        return
                    if slots[i] is None or any(t.shape[0] < n or t.shape[1:] != np.asarray(a).shape[1:] for t, a in zip(slots[i], batch)):
                        slots[i] = [torch.empty(np.asarray(a).shape, dtype=torch.from_numpy(np.empty(0, np.asarray(a).dtype)).dtype,
                                                pin_memory=True) for a in batch]
                    parts = [(lo, min(lo + (n + nthreads - 1) // nthreads, n)) for lo in range(0, n, (n + nthreads - 1) // nthreads)]
                    views = [t.numpy() for t in slots[i]]
                    list(pool.map(lambda p: [np.copyto(v[p[0] : p[1]], a[p[0] : p[1]]) for v, a in zip(views, batch)], parts))
                    hand_over((i, n))
                hand_over(None)
            except BaseException as e:  # noqa: BLE001 - handed to the consumer
                hand_over(e)

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        # the kernels run on the ENGINE's stream: that is the stream the uploads have to be ordered in front of (the caller's
        # current torch stream is the same one unless somebody changed it after the engine was made)
        main_stream = (torch.cuda.ExternalStream(eng.stream_ptr, device=dev) if getattr(eng, "stream_ptr", None)
                       else torch.cuda.current_stream(dev))
        up_stream = torch.cuda.Stream(dev)
        wins, preds, labs, sizes = [], [], [], []
        try:
            while True:
                item = ready.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                i, n = item
                # upload on a stream of its own: batch b + 1 crosses PCIe under the kernels of batch b
                with torch.cuda.stream(up_stream):
                    sig, seq, smap, lens, lab = (t[:n].to(dev, non_blocking=True) for t in slots[i])
                    up = torch.cuda.Event()
                    up.record(up_stream)
                main_stream.wait_event(up)
                for t in (sig, seq, smap, lens, lab):
                    t.record_stream(main_stream)
                labs.append(slots[i][4][:n].numpy().copy())
                logits = model.infer_chunks(sig, seq, smap, lens, md.kmer_context_bases)
                win = torch.empty(n, dtype=torch.float32, device=dev)
                pred = torch.empty(n, dtype=torch.uint8, device=dev)
                b = len(sizes)
                if b >= loss_buf.numel():
                    loss_buf = torch.cat([loss_buf, torch.zeros_like(loss_buf)])
                L.check(lib.rmr_validation_tally(eng.handle, logits.data_ptr(), lab.data_ptr(), n, km, kf, colmap, conf.data_ptr(),
                                                 win.data_ptr(), pred.data_ptr(), loss_buf.data_ptr() + 8 * b))
                wins.append(win)
                preds.append(pred)
                sizes.append(n)
                up.synchronize()  # the uploads of this slot are done: the reader may refill it under the kernels
                free.put(i)
        finally:
            stop.set()  # a consumer that raised leaves no reader behind holding pinned slots and open memmaps
            th.join(timeout=30.0)
            dataset._ds_iters = None
            pool.shutdown(wait=True)
        torch.cuda.synchronize(dev)
        win = torch.cat(wins).cpu().numpy()
        pred = torch.cat(preds).cpu().numpy().astype(np.int64)
        labels = np.concatenate(labs)
        losses = list((loss_buf[: len(sizes)].cpu().numpy() / np.asarray(sizes, np.float64)))
        full = conf.cpu().numpy().reshape(kf, kf)
        if world > 1:
            return self._global_metrics_from(kf, pred, win, labels, losses, filt_frac, local_conf=full.reshape(-1))
        present = (full.sum(0) + full.sum(1)) > 0
        ff, facc, fconf, thr = filtered_metrics(win, pred, labels, filt_frac)
        return VAL_METRICS(loss=np.mean(losses), acc=np.trace(full) / labels.size, num_calls=labels.size,
                           conf_mat=full[present][:, present], filt_frac=ff, filt_acc=facc, filt_conf_mat=fconf, filt_thresh=thr)

    @staticmethod
    def _global_metrics(probs, labels, losses, filt_frac):
        """This rank's calls -> the metrics of all ranks' calls (see run_validation)."""
        preds = np.argmax(probs, axis=1)
        win = np.take_along_axis(probs, preds[:, None], -1)[:, 0]
        return ValidationLogger._global_metrics_from(probs.shape[1], preds, win, labels, losses, filt_frac)

    @staticmethod
    def _global_metrics_from(k, preds, win, labels, losses, filt_frac, local_conf=None):
        from . import dist as rdist

        local = local_conf if local_conf is not None else np.bincount(labels.astype(np.int64) * k + preds, minlength=k * k)
        full = np.asarray(rdist.allreduce_counts(np.asarray(local, np.int64))).reshape(k, k)  # the collective: confusion counts over all GPUs
        present = (full.sum(0) + full.sum(1)) > 0
        conf = full[present][:, present]
        total = int(full.sum())
        acc = np.trace(full) / total
        lp = rdist.gather_arrays(np.stack([labels.astype(np.int64), preds.astype(np.int64)], axis=1))
        g_lab, g_pred, g_win = lp[:, 0], lp[:, 1], rdist.gather_arrays(win)  # win keeps its dtype: same quantile as one process
        loss = float(np.mean(rdist.gather_arrays(np.asarray(losses, np.float64).reshape(-1))))
        thr = np.quantile(g_win, filt_frac)
        if thr == g_win.max():
            thr *= 0.999999
        sure = g_win > thr
        n_sure = int(sure.sum())
        if n_sure == 0:
            return VAL_METRICS(loss=loss, acc=acc, num_calls=total, conf_mat=conf, filt_frac=1.0, filt_acc=np.nan,
                               filt_conf_mat=np.array([]), filt_thresh=np.nan)
        return VAL_METRICS(loss=loss, acc=acc, num_calls=total, conf_mat=conf, filt_frac=1 - n_sure / total,
                           filt_acc=(g_pred[sure] == g_lab[sure]).sum() / n_sure,
                           filt_conf_mat=confusion_matrix(g_lab[sure], g_pred[sure]), filt_thresh=thr)

    def validate_model(self, model, model_mod_bases, criterion, dataset, filt_frac=constants.DEFAULT_FILT_FRAC,
                       val_type="val", nepoch=0, niter=0, disable_pbar=False, world=1):
        ms = self.run_validation(model, model_mod_bases, criterion, dataset, filt_frac, disable_pbar=disable_pbar, world=world)
        self.fp.write(f"{val_type}\t{nepoch}\t{niter}\t{ms.acc:.6f}\t{mat_to_str(ms.conf_mat)}\t{ms.loss:.6f}\t"
                      f"{ms.num_calls}\t{ms.filt_frac:.4f}\t{ms.filt_acc:.6f}\t{mat_to_str(ms.filt_conf_mat)}\t"
                      f"{ms.filt_thresh}\n")
        return ms
