#!/usr/bin/env python3
"""bench.py — chunks/sec of the fused per-read modified-base-call hot path on MI355X.

A step = one pass of the hot path (chunk arrays -> class logits + per-label counts) over the
rank's batch of synthetic chunks, inputs already resident in HBM.  One process per GPU
(torchrun); chunks shard across ranks with no data-path collective; the only exchange is one
all-reduce of the per-label counts (RCCL over xGMI) at the end of the timed region.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement; `roofline` and `cpu_baseline`
objects included).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (arch, synth config, description)
    "convlstm_c100": ("conv_lstm", "C100", "synthetic 100-sig-pt CG chunks, ConvLSTM_w_ref size 64 k-mer (4,4) 2-class fp32 (BASELINE configs[2])"),
    "conv_c100": ("conv_only", "C100", "synthetic 100-sig-pt CG chunks, Conv_w_ref size 64 fp32 (BASELINE configs[1])"),
    "convlstm_c200": ("conv_lstm", "C200", "synthetic 200-sig-pt all-context chunks, ConvLSTM_w_ref 3-class fp32 (BASELINE configs[4] shape, fp32)"),
}

# algorithmic HBM bytes per chunk of the MFMA kernels (fp32 channel-last in + out), C100 ConvLSTM
ALG_BYTES = {"conv_merge1": 28 * 128 * 4 + 24 * 64 * 4, "conv_sig3": 92 * 16 * 4 + 28 * 64 * 4,
             "conv_seq2": 96 * 16 * 4 + 28 * 64 * 4, "lstm_head": 24 * 64 * 4 + 8}
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 256 CUs @ 2.4 GHz
PEAK_HBM_GBS = 8000.0


def kernel_flops_per_chunk(arch, L, size=64, K=9, num_out=2):
    """ALGORITHMIC flops (2 per MAC) per chunk of each MFMA kernel (no credit for
    lstm2's dead steps or for multiplying one-hot zeros)."""
    kw1 = 5 if arch == "conv_lstm" else 11
    P1 = L - kw1 + 1
    P2 = P1 - kw1 + 1
    P3 = (P2 - 9) // 3 + 1
    f = {}
    f["conv_sig3"] = 2 * size * 16 * 9 * P3
    # front: sig_conv1, sig_conv2 MACs + seq_conv1 as K*kw1 gather-adds x 16 ch
    f["front_sig"] = 2 * (4 * kw1 * P1 + 16 * 4 * kw1 * P2)
    f["front_seq"] = 16 * K * kw1 * P1  # seq_conv1 as K*kw1 gather-adds x 16 channels
    if arch == "conv_lstm":
        T = P3 - 4
        f["conv_seq2"] = 2 * size * 16 * 13 * P3
        f["conv_merge1"] = 2 * size * 2 * size * 5 * T
        f["lstm_head"] = 2 * (T * 2 * 4 * size * size + 4 * size * size + num_out * size)
        # the fused bf16 front kernel (k_fused.hip) does the work of the five kernels above it
        f["fused_front"] = f["front_sig"] + f["front_seq"] + f["conv_sig3"] + f["conv_seq2"] + f["conv_merge1"]
    else:
        PQ2 = P1 - 10
        T, T2 = P3 - 4, P3 - 8
        T3 = (T2 - 3) // 2 + 1
        T4 = (T3 - 3) // 2 + 1
        f["conv_seq2"] = 2 * 32 * 16 * 11 * PQ2
        f["conv_seq3"] = 2 * size * 32 * 9 * P3
        f["conv_merge1"] = 2 * size * 2 * size * 5 * T
        f["conv_merge2"] = 2 * size * size * 5 * T2
        f["conv_merge3"] = 2 * size * size * 3 * T3
        f["conv_merge4"] = 2 * size * size * 3 * T4
        f["fc_head"] = 2 * num_out * size * T4
    return f


def precision_check(state, data, kcb, gpu_logits):
    """Max |logit error| of each GPU path against a float64 CPU evaluation of the same network on
    the first 512 chunks (and the fp32 CPU reference's own error, for scale)."""
    import torch

    from oracle import oracle as O
    from oracle import torch_ref

    k = 512
    enc = O.compute_encoded_kmer_batch(kcb[0], kcb[1], data["sequence"][:k], data["sequence_to_signal_mapping"][:k],
                                       data["sequence_lengths"][:k])
    sig = torch.from_numpy(data["signal"][:k])
    with torch.no_grad():
        ref64 = torch_ref.from_state(state).double()(sig.double(), torch.from_numpy(enc).double()).numpy()
        ref32 = torch_ref.from_state(state)(sig, torch.from_numpy(enc)).numpy()
    out = {"chunks": k, "cpu_fp32_reference_vs_fp64": float(np.abs(ref32 - ref64).max())}
    for name, lg in gpu_logits.items():
        if lg is not None:
            out[f"{name}_vs_fp64"] = float(np.abs(lg[:k] - ref64).max())
            out[f"{name}_vs_cpu_fp32_reference"] = float(np.abs(lg[:k] - ref32).max())
    return out


def cpu_baseline(state, data, kcb, budget_s=12.0):
    """Reference CPU path timed on this box's host cores: single-thread C restatement of the
    Cython encode + torch.nn restatement of the network (all cores, eager, batch 2048)."""
    import torch

    from oracle import oracle as O
    from oracle import torch_ref

    cores = os.cpu_count() or 1
    net = torch_ref.from_state(state)
    # torch's intra-op pool does not scale to every core on this small network: pick the fastest thread
    # count on a probe of one full batch (best of 3 timings each; reported as `cores`)
    npb = min(2048, data["sequence_lengths"].shape[0])
    probe_enc = O.compute_encoded_kmer_batch(kcb[0], kcb[1], data["sequence"][:npb],
                                             data["sequence_to_signal_mapping"][:npb], data["sequence_lengths"][:npb])
    probe = (torch.from_numpy(data["signal"][:npb]), torch.from_numpy(probe_enc))
    best = (None, 0.0)
    tuned = {}
    with torch.no_grad():
        for nt in sorted({cores, max(cores // 2, 1), 64, 32, 16, 8}, reverse=True):
            if nt > cores:
                continue
            torch.set_num_threads(nt)
            net(*probe)
            rate = 0.0
            for _ in range(3):
                t0 = time.perf_counter()
                net(*probe)
                rate = max(rate, npb / (time.perf_counter() - t0))
            tuned[nt] = rate
            if rate > best[1]:
                best = (nt, rate)
    threads = best[0]
    torch.set_num_threads(threads)
    B = 2048
    n_avail = data["sequence_lengths"].shape[0]
    t_enc = t_net = 0.0
    done = 0
    it = 0
    with torch.no_grad():
        while True:
            st = (it * B) % max(n_avail - B, 1)
            sl = slice(st, st + B)
            t0 = time.perf_counter()
            enc = O.compute_encoded_kmer_batch(kcb[0], kcb[1], data["sequence"][sl],
                                               data["sequence_to_signal_mapping"][sl], data["sequence_lengths"][sl])
            t1 = time.perf_counter()
            net(torch.from_numpy(data["signal"][sl]), torch.from_numpy(enc))
            t2 = time.perf_counter()
            if it >= 2:  # 2 warm-up batches
                t_enc += t1 - t0
                t_net += t2 - t1
                done += B
            it += 1
            if it >= 7 and (t_enc + t_net) >= budget_s:
                break
            if it >= 400:
                break
    return {
        "value": done / (t_enc + t_net),
        "unit": "chunks/s",
        "cores": threads,
        "host_cores": cores,
        "kind": "port",
        "sample": f"{done} chunks in batches of {B}: C port of compute_encoded_kmer_batch (1 thread) + torch.nn "
                  f"restatement of the network, eager fp32, {threads} torch threads (best of {sorted(tuned)} on a "
                  f"one-batch probe; box has {cores} cores)",
        "thread_probe_chunks_per_s": {str(k): v for k, v in tuned.items()},
        "encode_chunks_per_s": done / t_enc,
        "model_chunks_per_s": done / t_net,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="convlstm_c100", choices=sorted(WORKLOADS))
    ap.add_argument("--chunks", type=int, default=1_000_000, help="chunks per GPU per step")
    ap.add_argument("--subbatch", type=int, default=0)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16x6", "bf16x3", "bf16"],
                    help="GEMM arithmetic: fp32 MFMA (default) or bf16 MFMA with split operands")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encode", action="store_true", help="skip the standalone encode-kernel roofline leg")
    ap.add_argument("--no-reads", action="store_true", help="skip the measured reads/sec leg (extract + infer from whole reads)")
    ap.add_argument("--no-alt", action="store_true", help="skip the bf16x6 (fp32-class split bf16 MFMA) comparison leg")
    ap.add_argument("--no-refine", action="store_true", help="skip the signal-mapping refinement (banded DP) leg")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--dist-backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--force-device", type=int, default=None, help="testing: put every rank on this GPU")
    args = ap.parse_args()

    import torch

    from remora_amd import dist as rdist
    from remora_amd import synth
    from remora_amd.engine import get_engine
    from remora_amd.model_util import model_from_state

    rank, world, local = rdist.init_process_group(args.dist_backend, set_device=args.force_device is None)
    if args.force_device is not None:
        local = args.force_device
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    torch.cuda.set_device(local)
    arch, cfg, desc = WORKLOADS[args.workload]
    cc, kcb, msl, num_out, _ = synth.CONFIGS[cfg]
    L = sum(cc)
    state = synth.synth_state(arch, 64, sum(kcb) + 1, num_out, seed=0)
    eng = get_engine(local)
    if args.subbatch:
        eng.set_subbatch(args.subbatch)
    md = dict(chunk_context=cc, kmer_context_bases=kcb)
    model = model_from_state(state, md, device=local, dtype=args.dtype)

    n = args.chunks
    data = synth.synth_chunks_config(cfg, n, shard=rank)
    dev = [torch.from_numpy(data[k]).cuda(local) for k in
           ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
    # centre the class logits on a sample (random weights otherwise call one class for every
    # chunk): shift fc.bias by the per-class median, identically on every rank and for the CPU leg
    probe = model.infer_chunks(*[t[:8192] for t in dev], kcb).cpu().numpy() if rank == 0 else None
    shift = torch.zeros(num_out, dtype=torch.float64)
    if rank == 0:
        shift = torch.from_numpy(np.median(probe, axis=0).astype(np.float64))
    if world > 1:
        on_gpu = torch.distributed.get_backend() == "nccl"
        shift = shift.cuda(local) if on_gpu else shift
        torch.distributed.broadcast(shift, src=0)
        shift = shift.cpu()
    state["fc.bias"] = (state["fc.bias"].astype(np.float64) - shift.numpy()).astype(np.float32)
    model = model_from_state(state, md, device=local, dtype=args.dtype)
    counts = torch.zeros(num_out, dtype=torch.int64, device=f"cuda:{local}")

    def step():
        return model.infer_chunks(*dev, kcb, label_counts=counts)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    counts.zero_()
    eng.profile_reset()
    eng.profile_enable(True)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logits = step()
    rdist.allreduce_counts(counts)  # the one collective of the job
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t1 = time.perf_counter()
    eng.profile_enable(False)
    elapsed = rdist.allreduce_max_float(t1 - t0)
    prof = eng.profile()

    total_chunks = n * world * args.steps
    value = total_chunks / elapsed

    # ---- the same job handed over as HOST buffers (numpy, pageable): PCIe-inclusive rate of the C-ABI boundary.
    #      Never `value`; reported beside it (pinned double-buffered upload under the kernels) ----
    host_leg = None
    if rank == 0 and world == 1 and not args.no_reads:
        host = [data[k] for k in ("signal", "sequence", "sequence_to_signal_mapping", "sequence_lengths")]
        hc = np.zeros(num_out, np.int64)
        model.infer_chunks(*host, kcb)  # warm-up (staging arenas, pinned slots)
        th0 = time.perf_counter()
        for _ in range(2):
            lg_h = model.infer_chunks(*host, kcb, label_counts=hc)
        th1 = time.perf_counter()
        host_leg = {"chunks_per_s": 2 * n / (th1 - th0), "ms_per_step": (th1 - th0) / 2 * 1e3,
                    "bytes_per_chunk_over_pcie": int(sum(a[0:1].nbytes for a in host) + 4 * num_out),
                    "max_abs_diff_vs_device_path": float(np.abs(lg_h[:4096] - logits[:4096].cpu().numpy()).max()),
                    "note": "numpy (pageable) chunk arrays in, logits + label counts out on the host; includes the host "
                            "copy into pinned slots, H2D, kernels, D2H"}

    # ---- E1 standalone (materialised one-hot): HBM-write roofline, outside the timed region ----
    enc_roof = None
    if rank == 0 and not args.no_encode:
        from remora_amd.encoded_kmers import compute_encoded_kmer_batch

        blk = min(n, 250_000)
        eng.profile_reset()
        eng.profile_enable(True)
        for rep in range(3):
            enc = compute_encoded_kmer_batch(kcb[0], kcb[1], dev[1][:blk], dev[2][:blk], dev[3][:blk])
        torch.cuda.synchronize()
        eng.profile_enable(False)
        ms, launches = eng.profile()["encode_kmers"]
        K = sum(kcb) + 1
        bytes_per_chunk = dev[1].shape[1] + 2 * dev[2].shape[1] + 2 + 4 * 4 * K * L
        gbs = bytes_per_chunk * blk * launches / (ms * 1e-3) / 1e9
        assert float(enc[:4096].sum()) == 4096 * K * L
        enc_roof = {"kernel": "encode_kmers", "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": gbs / PEAK_HBM_GBS, "bytes_per_chunk": bytes_per_chunk, "chunks_per_launch": blk,
                    "avg_launch_ms": ms / launches, "chunks_per_s": blk * launches / (ms * 1e-3)}
        del enc
    # ---- reads/sec measured end to end from whole reads (motif scan on the host, chunk extraction +
    #      fused inference on the GPU, logits back on the host), outside the timed region ----
    reads_leg = None
    if rank == 0 and not args.no_reads and arch == "conv_lstm" and cfg == "C100":
        from remora_amd.data_chunks import RemoraRead
        from remora_amd.inference import call_read_mods, call_reads_mods

        mdr = dict(md, motifs=[("CG", 0)], mod_bases=["m"], mod_long_names=["5mC"], can_base="C",
                   base_start_justify=False, offset=0, sig_map_refiner=None)
        nreads = 2048
        rs = []
        for i in range(nreads):
            r = synth.synth_read(5000, idx=i)
            rs.append(RemoraRead(dacs=r["dacs"], shift=r["shift"], scale=r["scale"], seq_to_sig_map=r["seq_to_sig_map"],
                                 int_seq=r["int_seq"], read_id=f"syn{i}"))
        res = call_reads_mods(rs, model, mdr)  # warm-up
        nchunks = sum(r[2].size for r in res)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(3):
            call_reads_mods(rs, model, mdr)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        t1a = time.perf_counter()
        for r in rs[:32]:
            call_read_mods(r, model, mdr)
        t1b = time.perf_counter()
        # the same reads as a stream of 8 batches of 512: staging of batch k+1 under the GPU work of batch k
        from remora_amd.inference import iter_call_reads_mods

        stream_batches = [rs[i : i + 512] for i in range(0, nreads, 512)] * 2
        for _ in iter_call_reads_mods(stream_batches[:2], model, mdr):
            pass
        torch.cuda.synchronize()
        tsa = time.perf_counter()
        for _ in iter_call_reads_mods(stream_batches, model, mdr):
            pass
        torch.cuda.synchronize()
        tsb = time.perf_counter()
        reads_leg = {"reads": nreads, "bases_per_read": 5000, "chunks_per_read": nchunks / nreads,
                     "batched_reads_per_s": 3 * nreads / (tb - ta), "batched_chunks_per_s": 3 * nchunks / (tb - ta),
                     "streamed_reads_per_s": 512 * len(stream_batches) / (tsb - tsa),
                     "single_read_api_reads_per_s": 32 / (t1b - t1a),
                     "note": f"call_reads_mods: one upload of the reads, GPU motif scan + geometry/fill + fused inference, logits back on "
                             f"the host, per batch of {nreads} reads"}

    # ---- signal-mapping refinement: banded DP kernels on resident reads, and the reads/sec of
    #      the whole per-read path for a model that carries a k-mer level table (rough re-scale + DP + calls) ----
    refine_leg = None
    if rank == 0 and world == 1 and not args.no_refine:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_refine

        refine_leg = bench_refine.measure(n_reads=8192, n_bases=5000, steps=2, warmup=1, cpu_reads=4, device=local)
        if reads_leg is not None:
            from remora_amd.refine_signal_map import SigMapRefiner

            table, center, base = bench_refine.synth_reads(64, 5000, seed=5)
            refiner = SigMapRefiner(_levels_array=table, center_idx=center, do_rough_rescale=True, scale_iters=0)
            mdf = dict(mdr, sig_map_refiner=refiner)

            nref = 2048  # the banded DP of a batch takes one read's latency (~20 ms) up to ~8 k reads: amortise it

            def fresh():
                return [RemoraRead(dacs=base[i % 64][0], shift=400.0, scale=60.0, seq_to_sig_map=base[i % 64][1].copy(),
                                   int_seq=base[i % 64][2], read_id=f"lv{i}") for i in range(nref)]

            call_reads_mods(fresh(), model, mdf)  # warm-up (also creates the device refiner)
            rs2 = fresh()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            call_reads_mods(rs2, model, mdf)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            refine_leg["reads_pipeline_with_refiner"] = {
                "reads": nref, "batched_reads_per_s": nref / (tb - ta),
                "note": "call_reads_mods with a loaded SigMapRefiner (do_rough_rescale, scale_iters=0, dwell_penalty): one "
                        "upload, GPU rough re-scale inputs (sorts) + host 19-point fits, GPU banded DP, motif scan, "
                        "extraction, inference"}

    # ---- POD5 signal decompression: the VBZ layer below zstd on resident rows ----
    vbz_leg = None
    if rank == 0 and world == 1 and not args.no_refine:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_vbz

        vbz_leg = bench_vbz.measure(n_rows=4096, row_samples=102400, steps=3, warmup=1, cpu_rows=8, device=local)

    # ---- dataset ETL: validate from an on-disk dataset, dataset prepare from aligned reads ----
    dataset_leg = None
    if rank == 0 and world == 1 and not args.no_refine:
        import bench_dataset

        dataset_leg = bench_dataset.measure(n_chunks=1 << 20, n_reads=2048, n_bases=5000, device=local)

    # ---- comparison leg: same job on the bf16 matrix cores with 3-part split operands (bf16x6) ----
    alt, alt_head = None, None
    if rank == 0 and world == 1 and args.dtype == "fp32" and arch == "conv_lstm" and not args.no_alt:
        model6 = model_from_state(state, md, device=local, dtype="bf16x6")
        c6 = torch.zeros(num_out, dtype=torch.int64, device=f"cuda:{local}")
        for _ in range(max(args.warmup, 1)):
            lg6 = model6.infer_chunks(*dev, kcb)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(args.steps):
            lg6 = model6.infer_chunks(*dev, kcb, label_counts=c6)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        alt = {"dtype": "bf16x6 (3-part exact split of fp32 operands, 6 bf16 MFMA products, fp32 accumulate)",
               "value": n * args.steps / (tb - ta), "unit": "chunks/s", "ms_per_step": (tb - ta) / args.steps * 1e3,
               "max_abs_logit_diff_vs_fp32_mfma_path": float((lg6 - logits).abs().max().item()),
               "label_counts": [int(x) for x in c6.tolist()]}
        alt_head = lg6[:512].cpu().numpy()
        del model6
    if rank != 0:
        return
    assert int(counts.sum().item()) == total_chunks, "label counts do not add up"

    flops = kernel_flops_per_chunk(arch, L, 64, sum(kcb) + 1, num_out)
    kern = {}
    for name, (ms, launches) in prof.items():
        fl = flops.get(name)
        kern[name] = {"ms_total": ms, "launches": launches, "avg_ms": ms / launches,
                      "tflops": (fl * n * args.steps / (ms * 1e-3) / 1e12) if fl else None}
    dom = max((k for k in kern if flops.get(k) and not k.startswith("front_")), key=lambda k: kern[k]["ms_total"])
    chunks_per_launch = n * args.steps / kern[dom]["launches"]
    achieved = flops[dom] * chunks_per_launch / (kern[dom]["avg_ms"] * 1e-3) / 1e12
    # HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes of this workload
    # (profiles/traffic.json, written by tools/summarize_profile.py --traffic; MI355X_MICROARCH.md
    # corrections applied there); scaled to this run's chunks per launch
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        ent = tj.get(args.dtype, {}).get(dom)
        if ent:
            traffic = ent["bytes_per_chunk"] * chunks_per_launch
    except (OSError, ValueError, KeyError):
        pass
    # fp32 MFMA: peak 157.3 TF.  bf16 MFMA with split operands executes NPROD bf16 products per
    # algorithmic MAC, so the matrix-pipe ceiling for algorithmic flops is 2.5 PF / NPROD.
    nprod = {"fp32": None, "bf16": 1, "bf16x3": 3, "bf16x6": 6}[args.dtype]
    peak = PEAK_FP32_MFMA_TFLOPS if nprod is None else PEAK_BF16_MFMA_TFLOPS / nprod
    roofline = {
        "kernel": dom, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
        "peak_note": ("v_mfma_f32_16x16x4_f32 dense peak" if nprod is None else
                      f"bf16 dense peak 2500 / {nprod} part products per algorithmic MAC"),
        "frac": achieved / peak, "traffic": float(traffic) if traffic else None,
        "algorithmic_bytes": float(ALG_BYTES.get(dom, 0) * chunks_per_launch) if dom in ALG_BYTES else None,
        "flop_per_chunk": flops[dom], "chunks_per_launch": chunks_per_launch, "avg_launch_ms": kern[dom]["avg_ms"],
    }
    gpu_ms = sum(k["ms_total"] for k in kern.values())
    total_flops = sum(flops.values())
    out = {
        "metric": "chunks/sec, 5mC CG ConvLSTM_w_ref inference (fused chunk arrays -> logits + label counts)",
        "value": value, "unit": "chunks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"fp32": "f32"}.get(args.dtype, args.dtype), "data": "synthetic",
        "config": {"workload": desc, "chunks_per_gpu_per_step": n, "chunk_len": L, "kmer_context_bases": list(kcb),
                   "num_out": num_out, "sharding": f"chunks sharded over {world} GPU(s), 1 count all-reduce"},
        "reads_per_sec": value / 312.0,
        "roofline": roofline,
        "whole_pipeline": {"algorithmic_tflops": total_flops * total_chunks / world / (gpu_ms * 1e-3) / 1e12,
                           "frac_of_fp32_mfma_peak": total_flops * total_chunks / world / (gpu_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                           "kernel_ms_sum": gpu_ms, "wall_ms": elapsed * 1e3},
        "kernels": kern,
        "encode_roofline": enc_roof,
        "alt_bf16x6": alt,
        "reads_pipeline": reads_leg,
        "refine_signal_map": refine_leg,
        "vbz_decode": vbz_leg,
        "dataset_etl": dataset_leg,
        "host_buffers_pcie_inclusive": host_leg,
        "label_counts": [int(x) for x in counts.tolist()],
    }
    if world == 1 and not args.no_cpu_baseline:
        nb = min(n, 1 << 17)
        sample = {k: v[:nb] for k, v in data.items() if isinstance(v, np.ndarray)}
        out["cpu_baseline"] = cpu_baseline(state, sample, kcb, args.cpu_budget)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        out["precision"] = precision_check(state, sample, kcb, {args.dtype + "_path": logits[:512].cpu().numpy(),
                                                                  "bf16x6_path": alt_head})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
