#!/usr/bin/env python3
"""Export one golden model fixture (tests/golden/model_*.npz: reference weights, chunk arrays and the reference's
logits, made by tools/gen_golden.py from the imported reference) as ONE raw little-endian binary a C program can read
with fread — the input of tests/c/infer_from_c.c.  Needs numpy only (no torch, no library).

    python tools/export_c_fixture.py tests/golden/model_convlstm_s64_l100_o2.npz out.bin

Layout:  int32 magic 'RMRC' | int32 arch, size, kmer_len, num_out, chunk_len, kb, ka, seq_w, map_w | int64 n_chunks,
n_weights | f32 weights[n_weights] (state_dict tensors in the order include/remora_hip.h documents for
rmr_model_create) | f32 signal[n,L] | i8 sequence[n,seq_w] | i16 mapping[n,map_w] | i16 lengths[n] | f32 logits[n,num_out]."""
import os
import struct
import sys

import numpy as np

CONV_ORDER = {
    0: [("sig_conv1", "sig_bn1"), ("sig_conv2", "sig_bn2"), ("sig_conv3", "sig_bn3"), ("seq_conv1", "seq_bn1"), ("seq_conv2", "seq_bn2"),
        ("merge_conv1", "merge_bn")],
    1: [("sig_conv1", "sig_bn1"), ("sig_conv2", "sig_bn2"), ("sig_conv3", "sig_bn3"), ("seq_conv1", "seq_bn1"), ("seq_conv2", "seq_bn2"),
        ("seq_conv3", "seq_bn3"), ("merge_conv1", "merge_bn1"), ("merge_conv2", "merge_bn2"), ("merge_conv3", "merge_bn3"),
        ("merge_conv4", "merge_bn4")],
}


def export(npz_path, out_path):
    g = np.load(npz_path, allow_pickle=False)
    arch = 0 if str(g["arch"]) == "ConvLSTM_w_ref" else 1
    size, kb, ka, L, num_out = (int(x) for x in g["params"])
    w = lambda name: np.ascontiguousarray(g["w__" + name.replace(".", "__")], np.float32).ravel()
    parts = []
    for conv, bn in CONV_ORDER[arch]:
        parts += [w(f"{conv}.weight"), w(f"{conv}.bias"), w(f"{bn}.weight"), w(f"{bn}.bias"), w(f"{bn}.running_mean"), w(f"{bn}.running_var")]
    if arch == 0:
        for l in ("lstm1", "lstm2"):
            parts += [w(f"{l}.weight_ih_l0"), w(f"{l}.weight_hh_l0"), w(f"{l}.bias_ih_l0"), w(f"{l}.bias_hh_l0")]
    parts += [w("fc.weight"), w("fc.bias")]
    blob = np.concatenate(parts)
    sig, seqs, maps, lens, logits = g["sigs"], g["seqs"], g["maps"], g["lens"], g["logits"]
    n = sig.shape[0]
    with open(out_path, "wb") as fh:
        fh.write(struct.pack("<4s9i2q", b"RMRC", arch, size, kb + ka + 1, num_out, L, kb, ka, seqs.shape[1], maps.shape[1], n, blob.size))
        for a, dt in ((blob, "<f4"), (sig.reshape(n, L), "<f4"), (seqs, "i1"), (maps, "<i2"), (lens, "<i2"), (logits, "<f4")):
            fh.write(np.ascontiguousarray(a, dt).tobytes())
    return n, blob.size


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit(__doc__)
    n, nw = export(sys.argv[1], sys.argv[2])
    print(f"{os.path.basename(sys.argv[1])}: {n} chunks, {nw} weights -> {sys.argv[2]}")
