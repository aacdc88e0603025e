"""Where does x of the role-specialised fused front kernel (V2) differ from V1's?  Experiment tool.

Runs the bf16 pipeline on the same chunks with RMR_FUSED_V=1 and with each listed V2 setting, dumps x (RMR_FUSED_DUMP_X) and
reports, per setting: mismatching elements, how they spread over (chunk % cb, position, channel), their size, and
whether two runs of the same setting agree with each other.

    python tools/debug_fused2.py [--cfg C100] [--n 8192] [--sets "cb4:RMR_FUSED2_CB=4;cb4w:RMR_FUSED2_CB=4,RMR_FUSED2_DBG=1"]
    (REMORA_HIP_LIB selects an experiment build)"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="C100")
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--pattern", default="", help="identical: every chunk = chunk 0;  alternate: chunk content by parity of (chunk // (cb * 256)) "
                    "(the tick of a 256-block grid), so that data left over from a block's previous or next iteration is recognisable")
    ap.add_argument("--cb", type=int, default=4)
    ap.add_argument("--sets", default="cb4:RMR_FUSED2_CB=4;cb4w:RMR_FUSED2_CB=4,RMR_FUSED2_DBG=1;cb3:RMR_FUSED2_CB=3;cb3w:RMR_FUSED2_CB=3,RMR_FUSED2_DBG=1")
    args = ap.parse_args()
    import torch

    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    cc, kcb, msl, num_out, _ = synth.CONFIGS[args.cfg]
    state = synth.synth_state("conv_lstm", 64, 9, num_out, seed=0)
    model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=kcb), device=0, dtype="bf16")
    d = synth.synth_chunks_config(args.cfg, args.n)
    keys4 = ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")
    if args.pattern:
        sel = np.zeros(args.n, dtype=np.int64) if args.pattern == "identical" else (np.arange(args.n) // (args.cb * 256)) % 2
        for k in keys4:
            d[k] = np.ascontiguousarray(d[k][sel])
    dev = [torch.from_numpy(d[k]).cuda() for k in keys4]
    L = d["signal"].shape[-1]
    T = ((L - 8 - 9) // 3 + 1) - 4

    def run(env):
        keys = dict(item.split("=", 1) for item in env.split(",") if item)
        with tempfile.NamedTemporaryFile(suffix=".x16", delete=False) as f:
            path = f.name
        os.unlink(path)
        old = {k: os.environ.get(k) for k in list(keys) + ["RMR_FUSED_DUMP_X"]}
        os.environ.update(keys)
        os.environ["RMR_FUSED_DUMP_X"] = path
        try:
            lg = model.infer_chunks(*dev, kcb)
            torch.cuda.synchronize()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        sys.stderr.flush()
        x = np.fromfile(path, dtype=np.uint16).reshape(args.n, T, 64)
        os.unlink(path)
        return x, lg.cpu().numpy() if hasattr(lg, "cpu") else np.asarray(lg)

    def f32(x):
        return (x.astype(np.uint32) << 16).view(np.float32)

    x1, l1 = run("RMR_FUSED_V=1")
    x1b, _ = run("RMR_FUSED_V=1")
    print(f"{args.cfg} n={args.n} lib={os.environ.get('REMORA_HIP_LIB', 'default')}: V1 twice equal: {np.array_equal(x1, x1b)}")
    for st in args.sets.split(";"):
        label, _, env = st.rpartition(":")
        cb = int(dict(item.split("=", 1) for item in env.split(",")).get("RMR_FUSED2_CB", 4))
        xs = [run("RMR_FUSED_V=2," + env)[0] for _ in range(args.reps)]
        same = all(np.array_equal(xs[0], x) for x in xs[1:])
        for r, x in enumerate(xs):
            bad = x != x1
            nb = int(bad.sum())
            msg = f"  {label or env} run {r}: {nb} of {bad.size} elements differ from V1"
            if nb:
                ch, pos, c = np.nonzero(bad)
                dv = np.abs(f32(x[bad]) - f32(x1[bad]))
                msg += (f"; chunks {np.unique(ch).size}; chunk%cb hist {np.bincount(ch % cb, minlength=cb).tolist()}; "
                        f"channel/16 hist {np.bincount(c // 16, minlength=4).tolist()}; "
                        f"|d| max {dv.max():.3g} median {np.median(dv):.3g}")
            print(msg)
            if nb:
                for c_ in np.unique(ch)[:6]:
                    pb = np.nonzero(bad[c_].any(axis=1))[0]
                    print(f"      chunk {c_} (iteration {c_ // cb}, slot {c_ % cb}): positions {pb.tolist()}, elements per position {bad[c_].sum(axis=1)[pb].tolist()}")
                    if args.pattern == "alternate":
                        other = x1[(c_ + cb * 256) % args.n if (c_ + cb * 256) < args.n else c_ - cb * 256]
                        print(f"          of the differing elements, equal to V1's x of the OTHER content at the same place: "
                              f"{int((x[c_][bad[c_]] == other[bad[c_]]).sum())} of {int(bad[c_].sum())}")
        print(f"  {label or env}: runs agree with each other: {same}")


if __name__ == "__main__":
    main()
