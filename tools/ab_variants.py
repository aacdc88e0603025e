"""A/B of library builds (tools/build_variant.sh) on the bf16 pipeline: for each libremora_hip_<name>.so (and the shipped
one) a fresh process measures ns/chunk of every kernel of a dtype's pipeline (HIP events, engine profile) on C100 and
C200 and hashes the logits - builds that only re-schedule work must agree bit for bit.

    python tools/ab_variants.py [--libs default,r2fused,...] [--dtype bf16] [--n 524288]"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(dtype, n, cfgs, arch):
    sys.path.insert(0, ROOT)
    import torch

    from remora_amd import synth
    from remora_amd.engine import get_engine
    from remora_amd.model_util import model_from_state

    out = {}
    for cfg in cfgs:
        cc, kcb, msl, num_out, _ = synth.CONFIGS[cfg]
        state = synth.synth_state(arch, 64, 9, num_out, seed=0)
        model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=kcb), device=0, dtype=dtype)
        d = synth.synth_chunks_config(cfg, n)
        dev = [torch.from_numpy(d[k]).cuda() for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
        eng = get_engine(0)
        for _ in range(2):
            lg = model.infer_chunks(*dev, kcb)
        torch.cuda.synchronize()
        eng.profile_reset()
        eng.profile_enable(True)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        t0 = time.perf_counter()
        for _ in range(4):
            lg = model.infer_chunks(*dev, kcb)
        eng.synchronize()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        eng.profile_enable(False)
        prof = eng.profile()
        out[cfg] = {"wall_ns_per_chunk": wall * 1e9 / (4 * n), "M_chunks_per_s": 4 * n / wall / 1e6,
                    "kernels_ns_per_chunk": {k: v[0] * 1e6 / (4 * n) for k, v in prof.items()},
                    "logits_sha": hashlib.sha256(lg.cpu().numpy().tobytes()).hexdigest()[:16]}
    print("RESULT " + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="default")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--arch", default="conv_lstm")
    ap.add_argument("--n", type=int, default=524288)
    ap.add_argument("--cfgs", default="C100,C200")
    ap.add_argument("--envs", default="", help="';'-separated env sets applied on top of each lib, each 'label:K=V,K=V' (label optional)")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args.dtype, args.n, args.cfgs.split(","), args.arch)
    runs = []
    for lib in args.libs.split(","):
        for es in (args.envs.split(";") if args.envs else [""]):
            label, _, kv = es.rpartition(":")
            extra = dict(item.split("=", 1) for item in kv.split(",") if item)
            runs.append((lib + ("/" + (label or kv) if es else ""), lib, extra))
    for name, lib, extra in runs:
        env = dict(os.environ)
        env.update(extra)
        if lib != "default":
            env["REMORA_HIP_LIB"] = os.path.join(ROOT, "remora_amd", f"libremora_hip_{lib}.so")
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--dtype", args.dtype, "--n", str(args.n), "--cfgs", args.cfgs,
                            "--arch", args.arch], env=env, capture_output=True, text=True, timeout=240)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        if p.returncode != 0 or not line:
            print(f"{name}: FAILED rc={p.returncode}\n{p.stderr[-1500:]}")
            continue
        res = json.loads(line[-1][7:])
        clk = [ln for ln in p.stderr.splitlines() if ln.startswith("[fused2 clock]") or ln.startswith("[stage clock]")]
        for ln in clk[-8:]:  # the last launch's (C200 when both configurations run)
            print("    " + ln)
        for cfg, r in res.items():
            ks = "  ".join(f"{k} {v:.3f}" for k, v in sorted(r["kernels_ns_per_chunk"].items(), key=lambda kv: -kv[1]))
            print(f"{name:24s} {cfg}: {r['M_chunks_per_s']:7.2f} M chunks/s  wall {r['wall_ns_per_chunk']:.3f} ns/chunk | {ks} | sha {r['logits_sha']}", flush=True)


if __name__ == "__main__":
    main()
