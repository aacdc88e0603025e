#!/usr/bin/env python3
"""Where does the plain-bf16 pipeline lose its accuracy?  CPU emulation (float64 arithmetic, round-to-nearest-even to
bf16 at chosen places) of ConvLSTM_w_ref on the synthetic benchmark data, one rounding site switched on at a time and
all together, against the float64 evaluation.  Mirrors the rounding sites of k_fused.hip / k_lstm_x16.hip:
  wconv  : BN-folded weights of the five MFMA convolutions (sig_conv2/3, seq_conv1/2, merge_conv1) rounded to bf16
  aconv  : activations between the convolutions (sig1, sig2, seq1, cat) rounded to bf16
  x      : merge_conv1's output (the LSTM input) rounded to bf16
  wlstm  : W_ih / W_hh of lstm1 and W_ih of lstm2 rounded to bf16
  h      : h_t rounded to bf16 before it feeds step t+1 / lstm2
`--split SITE[,SITE]`: those sites use a 2-part bf16 split (hi + lo, ~16 mantissa bits) instead.
Test infrastructure (imports oracle/): run by hand, CPU only.

    python tests/manual/bf16_error_sources.py [--cfg C100|C200] [--n 8192] [--split h,x]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import torch_ref  # noqa: E402
from remora_amd import synth  # noqa: E402


from oracle import lowp_emulation  # noqa: E402


def forward(net, sig, enc, sites, split=()):
    return lowp_emulation.forward(net, sig, enc, sites, split, FMT)


FMT = "bf16"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="C100")
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--split", default="")
    ap.add_argument("--fine", action="store_true", help="also one convolution's weights / one activation tensor at a time")
    ap.add_argument("--fmt", default="bf16", choices=["bf16", "fp16"], help="the 16-bit format emulated")
    ap.add_argument("--only-all", action="store_true", help="only the all-sites line")
    args = ap.parse_args()
    global FMT
    FMT = args.fmt
    cc, kcb, _, num_out, _ = synth.CONFIGS[args.cfg]
    state = synth.synth_state("conv_lstm", 64, 9, num_out, seed=0)
    probe = synth.synth_chunks_config(args.cfg, 8192, shard=0)
    penc = torch.from_numpy(O.compute_encoded_kmer_batch(kcb[0], kcb[1], probe["sequence"], probe["sequence_to_signal_mapping"], probe["sequence_lengths"]))
    with torch.no_grad():  # bench.py centres the class logits the same way (median of a fixed probe set)
        pl = torch_ref.from_state(state)(torch.from_numpy(probe["signal"]), penc).numpy()
    state["fc.bias"] = (state["fc.bias"].astype(np.float64) - np.median(pl, axis=0)).astype(np.float32)
    net = torch_ref.from_state(state)
    d = synth.synth_chunks_config(args.cfg, args.n)
    enc = torch.from_numpy(O.compute_encoded_kmer_batch(kcb[0], kcb[1], d["sequence"], d["sequence_to_signal_mapping"], d["sequence_lengths"])).double()
    sig = torch.from_numpy(d["signal"]).double()
    split = tuple(s for s in args.split.split(",") if s)
    with torch.no_grad():
        ref = forward(net, sig, enc, ())
        chk = net.double()(sig, enc)
        print(f"emulation vs torch module (float64): {float((ref - chk).abs().max()):.2e}; logit scale: std {float(ref.std()):.3f}")
        srt = ref.sort(1).values
        clear = (srt[:, -1] - srt[:, -2]) > 2e-2
        allsites = ("wconv", "aconv", "x", "wlstm", "h")
        fine = [("wconv." + k,) for k in ("sig2", "sig3", "seq1", "seq2", "merge1")] + [("aconv." + k,) for k in ("sig1", "sig2", "seq1", "cat")]
        for sites in ([] if args.only_all else [(s,) for s in allsites] + (fine if args.fine else []) + [("wconv", "aconv", "x"), ("wlstm", "h")]) + [allsites]:
            out = forward(net, sig, enc, sites, split)
            e = (out - ref).abs()
            agree = out.argmax(1) == ref.argmax(1)
            print(f"{'+'.join(sites):28s} max {float(e.max()):.2e}  mean {float(e.mean()):.2e}  argmax agree {float(agree.double().mean()):.5f}"
                  f"  (margin>2e-2: {float(agree[clear].double().mean()):.5f})")


if __name__ == "__main__":
    main()
