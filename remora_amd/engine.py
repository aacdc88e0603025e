"""Engine and model handles over the C ABI.  torch is used only as plumbing: device memory
(`data_ptr()`), the current stream, and `torch.jit.load` of model files."""
import ctypes
import threading

import numpy as np

from . import RemoraError
from . import _lib as L

_engines = {}
_engines_lock = threading.Lock()


def _torch():
    import torch

    return torch


class Engine:
    """One per (process, GPU).  Wraps rmr_engine."""

    def __init__(self, device=0, use_torch_stream=True, stream=None):
        lib = L.lib()
        torch = _torch()
        if not torch.cuda.is_available():
            raise RemoraError("no GPU visible: remora_amd has no CPU fallback")
        self.device = int(device)
        self.stream_ptr = None  # the borrowed HIP stream (None: the engine owns a private one)
        h = ctypes.c_void_p()
        if stream is not None:
            self.stream_ptr = int(stream)
            L.check(lib.rmr_engine_create(self.device, ctypes.c_void_p(stream), L.ENGINE_USE_STREAM, ctypes.byref(h)))
        elif use_torch_stream:
            s = torch.cuda.current_stream(self.device).cuda_stream
            self.stream_ptr = int(s)
            L.check(lib.rmr_engine_create(self.device, ctypes.c_void_p(s), L.ENGINE_USE_STREAM, ctypes.byref(h)))
        else:
            L.check(lib.rmr_engine_create(self.device, None, L.ENGINE_OWN_STREAM, ctypes.byref(h)))
        self._h = h
        self._lib = lib

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.rmr_engine_destroy(h)
            except Exception:
                pass

    @property
    def handle(self):
        return self._h

    @property
    def torch_device(self):
        return _torch().device("cuda", self.device)

    def synchronize(self):
        L.check(self._lib.rmr_engine_synchronize(self._h))

    def wait_for(self, producer):
        """Everything queued on `producer` (another Engine of this GPU) so far finishes before anything queued on this engine
        from now on starts: an event between the two streams, no host wait (rmr_engine_wait_for)."""
        L.check(self._lib.rmr_engine_wait_for(self._h, producer._h))

    def wait_submitted(self):
        """Block until the work submitted to the engine's stream SO FAR is done (an event, not a stream drain: work
        another thread queues behind it is not waited for).  A caller whose torch current stream is not the engine's
        needs this before it copies an asynchronous result."""
        torch = _torch()
        if self.stream_ptr is None:
            return self.synchronize()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.ExternalStream(self.stream_ptr, device=self.torch_device) if self.stream_ptr else
                  torch.cuda.default_stream(self.torch_device))
        ev.synchronize()

    def set_subbatch(self, chunks):
        L.check(self._lib.rmr_engine_set_subbatch(self._h, int(chunks)))

    # ---- profiling (HIP events around every kernel launch on the engine stream) ----
    def profile_enable(self, on=True):
        L.check(self._lib.rmr_profile_enable(self._h, int(bool(on))))

    def profile_reset(self):
        L.check(self._lib.rmr_profile_reset(self._h))

    def profile(self):
        """{kernel_name: (total_ms, launches)} for kernels launched since the last reset."""
        out = {}
        for i in range(self._lib.rmr_profile_num_kernels()):
            ms, n = ctypes.c_double(0), ctypes.c_int64(0)
            L.check(self._lib.rmr_profile_get(self._h, i, ctypes.byref(ms), ctypes.byref(n)))
            if n.value:
                out[self._lib.rmr_profile_kernel_name(i).decode()] = (ms.value, n.value)
        return out


def get_engine(device=None):
    """Process-wide engine for `device` (int / torch.device / None = current)."""
    torch = _torch()
    if not torch.cuda.is_available():
        raise RemoraError("no GPU visible: remora_amd has no CPU fallback")
    if device is None:
        idx = torch.cuda.current_device()
    elif isinstance(device, int):
        idx = device
    else:
        device = torch.device(device)
        if device.type != "cuda":
            raise RemoraError(f"remora_amd runs on the GPU only (got device {device})")
        idx = device.index if device.index is not None else torch.cuda.current_device()
    with _engines_lock:
        if idx not in _engines:
            _engines[idx] = Engine(idx, use_torch_stream=True)
        return _engines[idx]


_prep_engines = {}


def get_prep_engine(device):
    """A second engine (own non-blocking stream) on the GPU of `device`'s process-wide engine: read staging, motif scan
    and chunk extraction of one sub-batch run on it while the inference of another occupies the main engine's stream."""
    idx = get_engine(device).device
    with _engines_lock:
        if idx not in _prep_engines:
            _prep_engines[idx] = Engine(idx, use_torch_stream=False)
        return _prep_engines[idx]


_ingest_engines = {}


def get_ingest_engine(device=None):
    """A third engine (own stream, own mutex) for the ingest thread of a file-to-file run: the VBZ decode and the move-table
    expansion of the NEXT batches are small synchronous calls; on the model's engine they would wait, call by call, behind
    the inference kernels of the current batch (one mutex and one stream per engine)."""
    idx = get_engine(device).device
    with _engines_lock:
        if idx not in _ingest_engines:
            _ingest_engines[idx] = Engine(idx, use_torch_stream=False)
        return _ingest_engines[idx]


def _ptr(x):
    """(address, is_device, keepalive) for a torch tensor or numpy array (must be contiguous)."""
    torch = _torch()
    if isinstance(x, torch.Tensor):
        if not x.is_contiguous():
            raise RemoraError("tensor must be contiguous")
        return x.data_ptr(), x.is_cuda, x
    if isinstance(x, np.ndarray):
        if not x.flags.c_contiguous:
            raise RemoraError("array must be C-contiguous")
        return x.ctypes.data, False, x
    raise RemoraError(f"unsupported buffer type {type(x)}")


_CONV_ORDER = {
    "conv_lstm": [("sig_conv1", "sig_bn1"), ("sig_conv2", "sig_bn2"), ("sig_conv3", "sig_bn3"),
                  ("seq_conv1", "seq_bn1"), ("seq_conv2", "seq_bn2"), ("merge_conv1", "merge_bn")],
    "conv_only": [("sig_conv1", "sig_bn1"), ("sig_conv2", "sig_bn2"), ("sig_conv3", "sig_bn3"),
                  ("seq_conv1", "seq_bn1"), ("seq_conv2", "seq_bn2"), ("seq_conv3", "seq_bn3"),
                  ("merge_conv1", "merge_bn1"), ("merge_conv2", "merge_bn2"),
                  ("merge_conv3", "merge_bn3"), ("merge_conv4", "merge_bn4")],
}


def detect_arch(layer_names):
    """Architecture from the layer-name set, as the reference's exporter does
    (src/remora/model_util.py:231-263): lstm1 => conv_lstm, merge_conv4 => conv_only."""
    names = set(layer_names)
    lstm = {"sig_conv1", "sig_conv2", "sig_conv3", "seq_conv1", "seq_conv2", "merge_conv1", "lstm1", "lstm2", "fc"}
    conv = {"sig_conv1", "sig_conv2", "sig_conv3", "seq_conv1", "seq_conv2", "seq_conv3",
            "merge_conv1", "merge_conv2", "merge_conv3", "merge_conv4", "fc"}
    if lstm <= names and "merge_conv4" not in names:
        return "conv_lstm"
    if conv <= names and "lstm1" not in names:
        return "conv_only"
    raise RemoraError(f"unknown layer set in model: {sorted(names)}")


def state_to_blob(state):
    """{state_dict name: array} -> (arch, desc fields, flat fp32 blob) in the canonical order
    documented in include/remora_hip.h (rmr_model_create)."""
    get = lambda k: np.ascontiguousarray(np.asarray(state[k], dtype=np.float32)).ravel()
    arch = detect_arch({k.split(".")[0] for k in state})
    parts = []
    for conv, bn in _CONV_ORDER[arch]:
        for key in (f"{conv}.weight", f"{conv}.bias", f"{bn}.weight", f"{bn}.bias",
                    f"{bn}.running_mean", f"{bn}.running_var"):
            parts.append(get(key))
    if arch == "conv_lstm":
        for l in ("lstm1", "lstm2"):
            for key in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
                parts.append(get(f"{l}.{key}"))
    parts += [get("fc.weight"), get("fc.bias")]
    size = int(np.asarray(state["sig_conv3.weight"]).shape[0])
    ec = int(np.asarray(state["seq_conv1.weight"]).shape[1])
    num_out = int(np.asarray(state["fc.weight"]).shape[0])
    if ec % 4:
        raise RemoraError("seq_conv1 input channels not a multiple of 4")
    return arch, size, ec // 4, num_out, np.concatenate(parts)


class HipModel:
    """What `load_model` returns in place of the TorchScript module: callable
    `model(sigs, enc_kmers) -> logits`, `.parameters()`, `.eval()` — the surface the reference's
    callers use (src/remora/data_chunks.py:528-533, src/remora/inference.py:286,311-315,390,
    src/remora/model_util.py:559-562) — plus the fused `infer_chunks` fast path."""

    DTYPES = {"fp32": 0, "f32": 0, "bf16": 1, "bf16x3": 2, "bf16x6": 3, "f16": 4, "fp16": 4, "f16x3": 5}

    def __init__(self, state, chunk_len, device=None, engine=None, dtype="fp32"):
        torch = _torch()
        self.engine = engine if engine is not None else get_engine(device)
        arch, size, kmer_len, num_out, blob = state_to_blob(state)
        self.arch, self.size, self.kmer_len, self.num_out = arch, size, kmer_len, num_out
        self.chunk_len = int(chunk_len)
        if dtype not in self.DTYPES:
            raise RemoraError(f"unknown dtype {dtype!r}; choose from {sorted(self.DTYPES)}")
        self.dtype = dtype
        desc = L.ModelDesc(L.ARCH_CONV_LSTM if arch == "conv_lstm" else L.ARCH_CONV_ONLY, size,
                           kmer_len, num_out, self.chunk_len, self.DTYPES[dtype])
        lib = L.lib()
        want = lib.rmr_model_weight_count(ctypes.byref(desc))
        if want == 0:
            raise RemoraError(f"model not supported by the HIP engine: arch={arch} size={size} "
                              f"kmer_len={kmer_len} num_out={num_out} dtype={dtype} (size 1..256; the 16-bit dtypes need conv_lstm; "
                              "up to 64 channels f16 / bf16 on the fused kernels take 33..64 channels and a k-mer length of 9 or 6; "
                              "above 64 channels the dtypes are fp32, bf16 and f16 - the split dtypes stop at 64)")
        if want != blob.size:
            raise RemoraError(f"weight blob has {blob.size} floats, engine expects {want}")
        h = ctypes.c_void_p()
        L.check(lib.rmr_model_create(self.engine.handle, ctypes.byref(desc), blob.ctypes.data, blob.size, ctypes.byref(h)))
        self._h, self._lib = h, lib
        self.kernel_size = int(lib.rmr_model_padded_size(ctypes.byref(desc)))  # channels the kernels run at (zero-weight padding)
        # a device-resident token so that `next(model.parameters()).device` works
        self._param = torch.nn.Parameter(torch.zeros(1, device=self.engine.torch_device), requires_grad=False)
        self.training = False

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.rmr_model_destroy(h)
            except Exception:
                pass

    # ---- torch.nn.Module-like surface ----
    def parameters(self):
        yield self._param

    def eval(self):
        self.training = False
        return self

    def to(self, device):
        dev = _torch().device(device)
        if dev.type != "cuda" or (dev.index is not None and dev.index != self.engine.device):
            raise RemoraError("a HipModel lives on the GPU it was loaded on")
        return self

    @property
    def device(self):
        return self.engine.torch_device

    def __call__(self, sigs, seqs):
        return self.forward(sigs, seqs)

    def forward(self, sigs, seqs):
        """sigs f32[B,1,L], seqs f32[B,4K,L] (torch tensors; GPU or CPU, or numpy) -> logits
        f32[B,num_out] as a torch tensor on the inputs' device."""
        torch = _torch()
        if isinstance(sigs, np.ndarray):
            sigs = torch.from_numpy(np.ascontiguousarray(sigs, np.float32))
        if isinstance(seqs, np.ndarray):
            seqs = torch.from_numpy(np.ascontiguousarray(seqs, np.float32))
        if sigs.dim() != 3 or seqs.dim() != 3 or sigs.shape[1] != 1:
            raise RemoraError(f"expected sigs [B,1,L] and seqs [B,4K,L], got {tuple(sigs.shape)} {tuple(seqs.shape)}")
        n, _, Ls = sigs.shape
        if seqs.shape[0] != n or seqs.shape[2] != Ls or seqs.shape[1] != 4 * self.kmer_len or Ls != self.chunk_len:
            raise RemoraError(f"input shapes {tuple(sigs.shape)} {tuple(seqs.shape)} do not match model "
                              f"(chunk_len {self.chunk_len}, kmer_len {self.kmer_len})")
        if sigs.is_cuda != seqs.is_cuda:
            raise RemoraError("sigs and seqs must be on the same device")
        sigs = sigs.to(torch.float32).contiguous()
        seqs = seqs.to(torch.float32).contiguous()
        out = torch.empty((n, self.num_out), dtype=torch.float32, device=sigs.device)
        if n == 0:
            return out
        mem = L.MEM_DEVICE if sigs.is_cuda else L.MEM_HOST
        L.check(self._lib.rmr_forward(self._h, sigs.data_ptr(), seqs.data_ptr(), n, out.data_ptr(), mem))
        return out

    def infer_chunks(self, signal, sequence, mapping, lengths, kmer_context_bases, label_counts=None):
        """Fused hot path: chunk arrays (CoreRemoraDataset layout) -> logits.  Arrays may be
        torch CUDA tensors (zero-copy; returns a CUDA tensor, asynchronous on the engine stream)
        or numpy / CPU tensors (staged; returns numpy).  `label_counts` (int64[num_out], same
        residency) is incremented by the argmax histogram."""
        torch = _torch()
        kb, ka = (int(x) for x in kmer_context_bases)
        bufs = [signal, sequence, mapping, lengths]
        on_dev = isinstance(signal, torch.Tensor) and signal.is_cuda
        want = [(np.float32, torch.float32), (np.int8, torch.int8), (np.int16, torch.int16), (np.int16, torch.int16)]
        conv = []
        for b, (npd, tod) in zip(bufs, want):
            if on_dev:
                if not (isinstance(b, torch.Tensor) and b.is_cuda):
                    raise RemoraError("all chunk arrays must share the signal's residency")
                conv.append(b.to(tod).contiguous())
            else:
                if isinstance(b, torch.Tensor):
                    b = b.cpu().numpy()
                conv.append(np.ascontiguousarray(b, npd))
        signal, sequence, mapping, lengths = conv
        n = int(lengths.shape[0])
        if signal.numel() if on_dev else signal.size:
            sig_l = (signal.numel() if on_dev else signal.size) // max(n, 1)
            if sig_l != self.chunk_len:
                raise RemoraError(f"signal has {sig_l} samples per chunk, model expects {self.chunk_len}")
        seq_w, map_w = int(sequence.shape[1]), int(mapping.shape[1])
        if on_dev:
            out = torch.empty((n, self.num_out), dtype=torch.float32, device=signal.device)
            cptr = None
            if label_counts is not None:
                if not (isinstance(label_counts, torch.Tensor) and label_counts.is_cuda and label_counts.dtype == torch.int64):
                    raise RemoraError("label_counts must be a CUDA int64 tensor")
                cptr = label_counts.data_ptr()
            if n:
                L.check(self._lib.rmr_infer_chunks(self._h, signal.data_ptr(), sequence.data_ptr(), seq_w,
                                                   mapping.data_ptr(), map_w, lengths.data_ptr(), kb, ka, n,
                                                   out.data_ptr(), cptr, L.MEM_DEVICE))
            return out
        out = np.empty((n, self.num_out), np.float32)
        cptr = None
        if label_counts is not None:
            if not (isinstance(label_counts, np.ndarray) and label_counts.dtype == np.int64 and label_counts.flags.c_contiguous):
                raise RemoraError("label_counts must be a contiguous int64 numpy array")
            cptr = label_counts.ctypes.data
        if n:
            L.check(self._lib.rmr_infer_chunks(self._h, signal.ctypes.data, sequence.ctypes.data, seq_w,
                                               mapping.ctypes.data, map_w, lengths.ctypes.data, kb, ka, n,
                                               out.ctypes.data, cptr, L.MEM_HOST))
        return out
