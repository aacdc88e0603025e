"""Is the 16-bit pipeline bit-stable under load?  P processes share one GPU; each runs the same chunks through
model.infer_chunks REPS times and hashes the logits (and, with RMR_FUSED_DUMP_X unset, nothing else; the dump switches
RMR_FUSED_DUMP_X / RMR_DUMP_CAT / RMR_DEBUG_SKIP_LSTM exist in the experiment build only: make abl, REMORA_HIP_LIB=.../libremora_hip_abl.so): every hash of every
process must be the same.  (Round 4: a two-rank bf16 bench on one GPU once disagreed with the single-rank run by one argmax.)

    python tools/stress_determinism.py [--procs 3] [--reps 30] [--n 200000] [--dtype bf16] [--cfg C100]"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch

    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    cc, kcb, _, num_out, _ = synth.CONFIGS[args.cfg]
    state = synth.synth_state(args.arch, 64, 9, num_out, seed=0)
    model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=kcb), device=0, dtype=args.dtype)
    d = synth.synth_chunks_config(args.cfg, args.n, shard=7)
    dev = [torch.from_numpy(d[k]).cuda() for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
    hashes, first, bad, xh = {}, None, [], {}
    cat0, catinfo, dbg0 = None, [], {}
    xpath = f"/dev/shm/stress_x_{os.getpid()}.bin" if args.hash_x else None
    for r in range(int(os.environ.get("STRESS_REPS", args.reps))):  # (a --mix entry can run longer than the others: dtype@STRESS_REPS=N)
        if xpath:
            if os.path.exists(xpath):
                os.unlink(xpath)
            os.environ["RMR_FUSED_DUMP_X"] = xpath
        lg = model.infer_chunks(*dev, kcb).cpu().numpy()
        if os.environ.get("RMR_DEBUG_SKIP_LSTM"):
            lg = np.zeros_like(lg)  # (the LSTM did not run: only x counts)
        h = hashlib.sha256(lg.tobytes()).hexdigest()[:16]
        if os.environ.get("RMR_DUMP_CAT"):  # fp32 two-branch fold: where in cat [n][P3][128] (signal half | sequence half) the damage sits
            cat = np.fromfile(os.environ["RMR_DUMP_CAT"], dtype=np.float32).reshape(args.n, -1, 128)
            if r == 0:
                cat0 = cat
            elif not np.array_equal(cat0, cat) and len(catinfo) < 4:
                rows = np.nonzero((cat0 != cat).any(axis=(1, 2)))[0]
                ent = {"rep": r, "n_chunks": int(rows.size), "chunks": []}
                for ch in rows[:6]:
                    dm = cat0[ch] != cat[ch]
                    pos, chan = np.nonzero(dm.any(axis=1))[0], np.nonzero(dm.any(axis=0))[0]
                    ent["chunks"].append({"chunk": int(ch), "positions": pos.tolist()[:40], "n_pos": int(pos.size), "channels": chan.tolist()[:70], "n_chan": int(chan.size),
                                          "max_abs": float(np.abs(cat0[ch] - cat[ch]).max()), "elems": int(dm.sum())})
                catinfo.append(ent)
        if os.environ.get("RMR_SIG3_DEBUG_DUMP"):  # sig3_front_kernel's LDS intermediates: sig1 [n][P1][4], sig2 [n][P2][16]
            for name, width in (("sig1", 4), ("sig2", 16)):
                cur = np.fromfile(os.environ["RMR_SIG3_DEBUG_DUMP"] + "." + name, dtype=np.float32).reshape(args.n, -1, width)
                if r == 0:
                    dbg0[name] = cur
                elif not np.array_equal(dbg0[name], cur) and len(catinfo) < 8:
                    rows = np.nonzero((dbg0[name] != cur).any(axis=(1, 2)))[0]
                    ent = {"rep": r, "what": name, "n_chunks": int(rows.size), "chunks": []}
                    for ch in rows[:8]:
                        dm = dbg0[name][ch] != cur[ch]
                        pos, chan = np.nonzero(dm.any(axis=1))[0], np.nonzero(dm.any(axis=0))[0]
                        ent["chunks"].append({"chunk": int(ch), "positions": pos.tolist()[:40], "channels": chan.tolist(), "max_abs": float(np.abs(dbg0[name][ch] - cur[ch]).max()),
                                              "was": dbg0[name][ch][dm][:4].tolist(), "is": cur[ch][dm][:4].tolist()})
                    catinfo.append(ent)
        if xpath:  # which kernel moved: x (fused_front's output) or only the logits (the LSTM's)
            hx = hashlib.sha256(open(xpath, "rb").read()).hexdigest()[:8]
            xh[hx] = xh.get(hx, 0) + 1
            h = h + "/x:" + hx
        hashes[h] = hashes.get(h, 0) + 1
        if first is None:
            first = lg
        elif not np.array_equal(first, lg):
            rows = np.nonzero((first != lg).any(axis=1))[0]
            info = {"rep": r, "chunks": rows[:8].tolist(), "n": int(rows.size), "max_abs": float(np.abs(first - lg).max())}
            if args.detail and len(bad) < 3:
                # structure of the damage: runs of consecutive chunks, and whether a wrong row is another chunk's right row
                runs, start = [], int(rows[0])
                for a_, b_ in zip(rows[:-1], rows[1:]):
                    if b_ != a_ + 1:
                        runs.append((start, int(a_) - start + 1))
                        start = int(b_)
                runs.append((start, int(rows[-1]) - start + 1))
                lookup = {first[i].tobytes(): i for i in range(first.shape[0])}
                moved = [(int(i), lookup.get(lg[i].tobytes())) for i in rows[:12]]
                info.update(runs=runs[:24], same_as_chunk=moved, nan=int(np.isnan(lg).sum()))
            bad.append(info)
    print("RESULT " + json.dumps({"hashes": hashes, "differing_reps": bad[:6], "cat": catinfo}))


def child_specs(args):
    """Every pipeline 'dtype[:cfg[:arch]]' of --child-specs in turn: REPS calls each, the logits of every call hashed."""
    sys.path.insert(0, ROOT)
    import time

    import torch

    from remora_amd import synth
    from remora_amd.model_util import model_from_state

    out = {}
    for spec in args.child_specs.split(","):
        parts = spec.split(":")
        dt, cfg, arch = parts[0], (parts[1] if len(parts) > 1 else "C100"), (parts[2] if len(parts) > 2 else "conv_lstm")
        size = int(parts[3]) if len(parts) > 3 else 64   # 'dtype[:cfg[:arch[:size[:chunks]]]]': > 64 channels = the streamed kernels,
        n_spec = int(parts[4]) if len(parts) > 4 else args.n  # a few hundred chunks = the small-batch LSTM (k_stream.hip, k_lstm.hip)
        cc, kcb, _, num_out, _ = synth.CONFIGS[cfg]
        state = synth.synth_state(arch, size, 9, num_out, seed=0)
        model = model_from_state(state, dict(chunk_context=cc, kmer_context_bases=kcb), device=0, dtype=dt)
        d = synth.synth_chunks_config(cfg, n_spec, shard=7)
        dev = [torch.from_numpy(d[k]).cuda() for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
        hashes = {}
        t0 = time.perf_counter()
        for _ in range(args.reps):
            lg = model.infer_chunks(*dev, kcb).cpu().numpy()
            h = hashlib.sha256(lg.tobytes()).hexdigest()[:16]
            hashes[h] = hashes.get(h, 0) + 1
        out[spec] = {"hashes": hashes, "ms_per_call": (time.perf_counter() - t0) / args.reps * 1e3}
        print(f"  [{os.environ.get('REMORA_HIP_LIB', 'shipped library').split('/')[-1]}] {spec}: {args.reps} calls, {out[spec]['ms_per_call']:.2f} ms each, "
              f"{len(hashes)} distinct result(s)", file=sys.stderr, flush=True)
        del model, dev
    print("RESULT " + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=3)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--cfg", default="C100")
    ap.add_argument("--hash-x", action="store_true", help="process 0 also hashes x, the fused front kernel's output (RMR_FUSED_DUMP_X: a sync + copy between the two kernels); the others hammer on")
    ap.add_argument("--mix", default="", help="comma list of dtypes, one per process (overrides --procs / --dtype): who disturbs whom")
    ap.add_argument("--detail", action="store_true")
    ap.add_argument("--json", action="store_true", help="--mix: one JSON line with the verdict per process (tests/test_gpu_shared_gpu.py)")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--arch", default="conv_lstm", choices=["conv_lstm", "conv_only"])
    ap.add_argument("--child-specs", default="", help="internal: run these pipelines in turn in this process")
    ap.add_argument("--jitter", default="", help="comma list of pipelines 'dtype[:cfg[:arch]]': each runs ALONE on the GPU, two calls with the shipped "
                    "library and --reps calls with the jitter build (make jitter: random per-wave sleeps around every barrier); every call "
                    "of both must return the same bits")
    args = ap.parse_args()
    if args.child:
        return child(args)
    if args.child_specs:
        return child_specs(args)
    if args.jitter:
        # two processes, one per library, each running every pipeline in turn (alone on the GPU at any moment of its run)
        jit = os.environ.get("RMR_JITTER_LIB") or os.path.join(ROOT, "remora_amd", "libremora_hip_jitter.so")
        got = []
        for lib, reps in ((None, 2), (jit, args.reps)):
            env = dict(os.environ)
            env.pop("REMORA_HIP_LIB", None)
            if lib:
                env["REMORA_HIP_LIB"] = lib
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child-specs", args.jitter, "--reps", str(reps), "--n", str(args.n)],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=1500)
            sys.stderr.write(p.stderr)  # the child's progress lines
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
            if not line:
                print(f"  jitter: child with {lib or 'the shipped library'} FAILED rc={p.returncode}: {p.stderr[-800:]}", flush=True)
                if args.json:
                    print("JSON " + json.dumps([{"pipeline": "*", "ok": False, "error": f"child rc {p.returncode}: {p.stderr[-800:]}"}]), flush=True)
                return
            got.append(json.loads(line[-1][7:]))
        verdicts = []
        for spec in args.jitter.split(","):
            ref, jh = got[0][spec]["hashes"], got[1][spec]["hashes"]
            ok = len(ref) == 1 and set(jh) == set(ref)
            nbad = sum(c for h, c in jh.items() if h not in ref)
            verdicts.append({"pipeline": spec, "ok": ok, "shipped_hashes": ref, "jitter_hashes": jh, "differing_runs": nbad, "runs": sum(jh.values()),
                             "ms_per_call_shipped": got[0][spec]["ms_per_call"], "ms_per_call_jitter": got[1][spec]["ms_per_call"]})
            print(f"  jitter {spec}: {'same bits as the shipped build in' if ok else 'DIFFERS:'} {sum(jh.values()) - nbad} of {sum(jh.values())} calls "
                  f"({got[0][spec]['ms_per_call']:.2f} -> {got[1][spec]['ms_per_call']:.2f} ms per call)" + ("" if ok else f" {jh} vs {ref}"), flush=True)
        if args.json:
            print("JSON " + json.dumps(verdicts), flush=True)
        return
    if args.mix:
        dts = args.mix.split(",")
        def env_of(spec):  # "fp32@KEY=VAL@KEY2=VAL2"
            e = dict(os.environ)
            e.update(kv.split("=", 1) for kv in spec.split("@")[1:])
            return e

        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", "--reps", str(args.reps), "--n", str(args.n), "--dtype", dt.split("@")[0],
                                "--cfg", args.cfg] + (["--detail"] if args.detail else []) + (["--hash-x"] if "RMR_DEBUG_SKIP_LSTM" in dt else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env_of(dt)) for dt in dts]
        verdicts = []
        for dt, p in zip(dts, ps):
            out, err = p.communicate(timeout=900)
            line = [ln for ln in out.splitlines() if ln.startswith("RESULT ")]
            res = json.loads(line[-1][7:]) if line else {"hashes": {"FAILED": 1}, "differing_reps": []}
            nbad = sum(c for h, c in res["hashes"].items()) - max(res["hashes"].values())
            verdicts.append({"process": dt, "failed": not line, "runs": sum(res["hashes"].values()), "differing_runs": nbad, "stderr": "" if line else err[-500:]})
            print(f"  mix {args.mix}: process {dt}: {nbad} of {args.reps} runs differ; (first chunk, chunks, max |d|) "
                  f"{[(d['chunks'][0], d['n'], round(d['max_abs'], 4)) for d in res['differing_reps']]}", flush=True)
            for ent in res.get("cat", []):
                print(f"      {ent.get('what', 'cat')} rep {ent['rep']}: {ent['n_chunks']} chunks differ", flush=True)
                for c in ent["chunks"]:
                    print(f"         {c}", flush=True)
            for d in res["differing_reps"]:
                if "runs" in d:
                    print(f"      rep {d['rep']}: runs (start, length) {d['runs']}; wrong row == right row of chunk: {d['same_as_chunk']}; NaNs {d['nan']}", flush=True)
        if args.json:
            print("JSON " + json.dumps(verdicts), flush=True)
        return
    for procs in sorted({1, args.procs}):
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", "--reps", str(args.reps), "--n", str(args.n), "--dtype", args.dtype,
                                "--cfg", args.cfg] + (["--hash-x"] if (args.hash_x and i == 0) else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(procs)]
        allh = {}
        for i, p in enumerate(ps):
            out, err = p.communicate(timeout=900)
            line = [ln for ln in out.splitlines() if ln.startswith("RESULT ")]
            if not line:
                print(f"  process {i}: FAILED rc={p.returncode} {err[-400:]}")
                continue
            res = json.loads(line[-1][7:])
            for h, c in res["hashes"].items():
                allh[h] = allh.get(h, 0) + c
            if res["differing_reps"]:
                print(f"  process {i}: {res['differing_reps']}")
        print(f"{args.cfg} {args.dtype} n={args.n}: {procs} process(es) x {args.reps} runs -> logits hashes {allh}", flush=True)


if __name__ == "__main__":
    main()
