"""`python -m remora_amd infer from_pod5_and_bam POD5 BAM --model MODEL.pt --out-bam OUT.bam`
— the sub-command of the reference CLI that sits on the hot path (src/remora/parsers.py:1294-1417,
:1582-1612), same positional arguments and the same meaning of --model / --out-bam / --device /
--num-reads; `python -m remora_amd validate from_remora_dataset DATASET --model MODEL.pt` (:1800-1960):
the stored chunks of an on-disk dataset (directory or config) through the model with the model's chunk /
k-mer contexts, the reference's validation summary line against the stored labels; and `python -m remora_amd dataset prepare`
(:64-337), the ETL into the on-disk chunk format, with the bookkeeping verbs around that format (`dataset inspect |
make_config | merge | head | copy`, :359-727: host code over `CoreRemoraDataset` / `RemoraDataset`, no kernel involved)."""
import argparse
import os
import sys

from . import RemoraError, constants


def _validate(args):
    """src/remora/parsers.py:1890-1960: the dataset is loaded with the model's contexts and without extra arrays,
    every chunk goes through the model, one summary line in the reference's format."""
    import torch

    from .data_chunks import CoreRemoraDataset, RemoraDataset, load_dataset
    from .model_util import load_torchscript_model
    from .validate import ValidationLogger

    from . import dist as rdist

    rank, world, dev = rdist.setup_ranks(args.gpus, args.procs_per_gpu)
    model, md = load_torchscript_model(args.model, device=args.device if dev is None else dev, eval_only=True, dtype=args.dtype)
    over = {"extra_arrays": {}, "kmer_context_bases": md["kmer_context_bases"], "chunk_context": md["chunk_context"]}
    paths, props, hashes = load_dataset(args.remora_dataset_path)
    dataset = RemoraDataset([CoreRemoraDataset(p, override_metadata=dict(over), infinite_iter=False,
                                               do_check_super_batches=True) for p in paths],
                            props, hashes, batch_size=args.batch_size)
    if world > 1:  # this rank's contiguous share of every core dataset's rows
        dataset = dataset.shard(rank, world)
    # rank 0 writes the summary (every rank holds the same global metrics); the per-chunk table is per rank
    sink = open(os.devnull, "w") if rank != 0 else None
    out_fp = sink if sink is not None else (sys.stdout if args.out_file is None else open(args.out_file, "w", buffering=1))
    full_name = args.full_results_filename
    if full_name is not None and world > 1:
        full_name = f"{full_name}.rank{rank:03d}"
    full_fp = None if full_name is None else open(full_name, "w", buffering=1)
    try:
        ValidationLogger(out_fp, full_fp).validate_model(model, md["mod_bases"], torch.nn.CrossEntropyLoss(), dataset,
                                                         args.pct_filt / 100, world=world)
    finally:
        for fp in (out_fp, full_fp):
            if fp not in (None, sys.stdout):
                fp.close()
    rdist.barrier()
    return 0


def _dataset_prepare(args):
    """src/remora/parsers.py:281-337."""
    from .engine import get_engine
    from .io import parse_bed
    from .prepare_train_data import extract_chunk_dataset
    from .refine_signal_map import SigMapRefiner
    from .util import Motif, prepare_out_dir

    if args.mod_base is None and not args.mod_base_control:
        raise RemoraError("Must specify either --mod-base or --mod-base-control")
    from . import dist as rdist

    rank, world, dev = rdist.setup_ranks(args.gpus, args.procs_per_gpu)
    if rank == 0:
        prepare_out_dir(args.output_path, args.overwrite)
    rdist.barrier()
    refiner = SigMapRefiner(kmer_model_filename=args.refine_kmer_level_table, do_rough_rescale=args.refine_rough_rescale,
                            scale_iters=args.refine_scale_iters, algo=args.refine_algo,
                            half_bandwidth=args.refine_half_bandwidth, sd_params=args.refine_short_dwell_parameters,
                            do_fix_guage=True, rough_rescale_method=args.rough_rescale_method)
    if not refiner.is_valid:
        raise RemoraError("Invalid signal mapping refiner settings.")
    dataset, errs = extract_chunk_dataset(
        bam_path=args.bam, pod5_path=args.pod5, out_path=args.output_path, mod_base=args.mod_base,
        mod_base_control=args.mod_base_control, motifs=[Motif(*m) for m in args.motif],
        focus_ref_pos=None if args.focus_reference_positions is None else parse_bed(args.focus_reference_positions),
        chunk_context=args.chunk_context, min_samps_per_base=args.min_samples_per_base,
        max_chunks_per_read=args.max_chunks_per_read, pa_scaling=None, sig_map_refiner=refiner,
        kmer_context_bases=args.kmer_context_bases, base_start_justify=args.base_start_justify, offset=args.offset,
        num_reads=args.num_reads, basecall_anchor=args.basecall_anchor, rev_sig=args.reverse_signal,
        save_every=args.save_every, skip_shuffle=args.skip_shuffle, reads_per_batch=args.reads_per_batch,
        engine=get_engine(args.device if dev is None else dev), rank=rank, world=world)
    if rank != 0:
        return 0
    if dataset is None:
        print("no reads with signal and alignment")
        return 0
    for reason, cnt in sorted(errs.items(), key=lambda kv: -kv[1]):
        print(f"{cnt:>7,} : {reason}")
    print(f"Extracted {dataset.size:,} chunks -> {args.output_path}")
    print(f"Label distribution: {dataset.label_summary}")
    return 0


def _dataset_inspect(args):
    """src/remora/parsers.py:359-376."""
    import json

    from .data_chunks import CoreRemoraDataset, RemoraDataset, load_dataset

    paths, props, hashes = load_dataset(args.remora_dataset_path)
    dataset = RemoraDataset([CoreRemoraDataset(p, do_check_super_batches=True) for p in paths], props, hashes)
    print(f"Dataset summary:\n{dataset.summary}")
    if args.out_path is not None:
        with open(args.out_path, "w") as fh:
            json.dump(dataset.get_config(), fh)
    return 0


def _dataset_make_config(args):
    """src/remora/parsers.py:414-458: default weights are the dataset sizes (chunks drawn uniformly overall)."""
    import json

    import numpy as np

    from .data_chunks import CoreRemoraDataset, RemoraDataset, load_dataset

    if args.dataset_weights is not None:
        if len(args.dataset_weights) != len(args.dataset_paths):
            raise RemoraError("Weights must be same length as input datasets.")
        if any(w <= 0 for w in args.dataset_weights):
            raise RemoraError("Weights must be positive.")
    core_paths, core_weights, core_hashes = [], [], []
    for i, ds_path in enumerate(args.dataset_paths):
        paths, weights, hashes = load_dataset(ds_path)
        core_paths.extend(paths)
        scale = sum(CoreRemoraDataset(p).size for p in paths) if args.dataset_weights is None else args.dataset_weights[i]
        core_weights.extend(weights * scale)
        if hashes is None or core_hashes is None:
            core_hashes = None
        else:
            core_hashes.extend(hashes)
    core_weights = np.array(core_weights)
    dataset = RemoraDataset([CoreRemoraDataset(p) for p in core_paths], core_weights / core_weights.sum(), core_hashes)
    with open(args.out_path, "w") as fh:
        json.dump(dataset.get_config(), fh)
    print(dataset.summary)
    return 0


def _dataset_merge(args):
    """src/remora/parsers.py:493-572: all rows of several datasets (labels converted to the merged label set)
    copied into one new dataset, optionally capped at --max-size in proportion, then shuffled."""
    import numpy as np

    from .data_chunks import CoreRemoraDataset, RemoraDataset, compute_best_split, load_dataset
    from .util import prepare_out_dir

    prepare_out_dir(args.out_path, args.overwrite)
    paths = [sub for ds_path in args.dataset_paths for sub in load_dataset(ds_path)[0]]
    dataset = RemoraDataset([CoreRemoraDataset(p, infinite_iter=False, do_check_super_batches=True) for p in paths],
                            np.ones(len(paths)) / len(paths))
    sizes = np.array([ds.size for ds in dataset.datasets])
    if args.max_size is not None and sizes.sum() > args.max_size:
        sizes = compute_best_split(args.max_size, sizes / sizes.sum())
    md = dataset.metadata.copy()
    md.allocate_size, md.max_seq_len = int(sizes.sum()), max(ds.metadata.max_seq_len for ds in dataset.datasets)
    md.dataset_start = md.dataset_end = 0
    merged = CoreRemoraDataset(data_path=args.out_path, mode="w", metadata=md)
    for ds, size in zip(dataset.datasets, sizes):
        ds.metadata.dataset_end = ds.metadata.dataset_start + int(size)
        ds.adjust_batch_params()
        for sb in ds.iter_super_batches():
            merged.write_batch(sb)
        merged.flush()
    merged.shuffle()
    merged.flush()
    print(f"Saved core dataset:\n{merged.summary}")
    return 0


def _dataset_head(args):
    """src/remora/parsers.py:604-655: the first `num_chunks` rows of a core dataset as a new (shuffled) dataset."""
    from .data_chunks import CoreRemoraDataset
    from .util import prepare_out_dir

    prepare_out_dir(args.out_path, args.overwrite)
    src = CoreRemoraDataset(args.in_path, infinite_iter=False, do_check_super_batches=True)
    md = src.metadata.copy()
    md.allocate_size, md.dataset_start, md.dataset_end = args.num_chunks, 0, 0
    head = CoreRemoraDataset(data_path=args.out_path, mode="w", metadata=md)
    src.adjust_batch_params()
    for sb in src.iter_super_batches():
        room = args.num_chunks - head.metadata.dataset_end
        if sb["labels"].size >= room:
            head.write_batch({n: a[:room] for n, a in sb.items()})
            break
        head.write_batch(sb)
    head.flush()
    head.shuffle()
    head.flush()
    print(f"Saved core dataset:\n{head.summary}")
    return 0


def _dataset_copy(args):
    """src/remora/parsers.py:684-727: every core dataset of a dataset / config copied to OUT/dataset_NNN, with
    OUT/dataset.cfg pointing at the copies and OUT/sources.txt recording where they came from."""
    import json
    import os
    import shutil

    from .data_chunks import CoreRemoraDataset, RemoraDataset, load_dataset
    from .util import prepare_out_dir

    prepare_out_dir(args.out_path, args.overwrite)
    paths, props, hashes = load_dataset(args.in_path)
    out_dirs = []
    with open(os.path.join(args.out_path, "sources.txt"), "w") as src_fh:
        for i, src in enumerate(paths):
            if any(os.path.isdir(os.path.join(src, item)) for item in os.listdir(src)):
                raise RemoraError(f"Source dataset has nested directory: {src}")
            dst = os.path.join(args.out_path, f"dataset_{i:03}")
            src_fh.write(f"{src}\t{dst}\n")
            shutil.copytree(src, dst)
            out_dirs.append(dst)
    dataset = RemoraDataset([CoreRemoraDataset(d) for d in out_dirs], props, hashes)
    with open(os.path.join(args.out_path, "dataset.cfg"), "w") as fh:
        json.dump(dataset.get_config(), fh)
    print(dataset.summary)
    return 0


def _infer(args):
    from . import dist as rdist
    from .inference import infer_from_pod5_and_bam
    from .model_util import load_torchscript_model

    # this rank's share of the BAM, found in a thread under the model load.  By byte range (default): where the share and
    # the next one begin is read off the bytes there (io.bam_byte_shard - no pass over the file, and the rank in front
    # verifies the guess); REMORA_AMD_BAM_SHARD=scan: by record count from one exact pass (the launcher's, or this rank's own)
    erank, eworld, _ = rdist.env_rank_world()
    shard_future = None
    if eworld > 1:
        from concurrent.futures import ThreadPoolExecutor

        from . import io as rio

        shard_future = ThreadPoolExecutor(max_workers=1).submit(rio.shard_of, args.in_bam, erank, eworld)
    rank, world, dev = rdist.setup_ranks(args.gpus, args.procs_per_gpu)
    loaded = [load_torchscript_model(m, device=args.device if dev is None else dev, eval_only=True, dtype=args.dtype)
              for m in args.model]
    model, md = [x[0] for x in loaded], [x[1] for x in loaded]
    if len({m["can_base"] for m in md}) != len(md):
        raise RemoraError("Only one model per canonical base allowed.")
    import time

    label_counts = {}
    rdist.barrier()  # every rank has its model: the clock below covers the file-to-file work, not interpreter start-up
    t0 = time.perf_counter()
    stats = infer_from_pod5_and_bam(args.pod5, args.in_bam, model, md, args.out_bam, num_reads=args.num_reads,
                                    reads_per_batch=args.reads_per_batch, ref_anchored=args.reference_anchored,
                                    rank=rank, world=world, label_counts_out=label_counts, bam_level=args.bam_level,
                                    shard_future=shard_future)
    dt = time.perf_counter() - t0
    if rank != 0:
        return 0
    ok = stats.pop(None, 0)
    nrec = ok + sum(stats.values())
    print(f"{nrec} records in {dt:.2f} s = {nrec / max(dt, 1e-9):.0f} reads/s (models loaded; parts joined)", file=sys.stderr)
    print(f"called {ok} reads -> {args.out_bam}" + (f" ({args.gpus} GPUs" + (f" x {args.procs_per_gpu} processes" if args.procs_per_gpu > 1 else "") + ")"
                                                  if world > 1 else ""))
    for reason, cnt in sorted(stats.items(), key=lambda kv: -kv[1]):
        print(f"{cnt:>7} : {reason}")
    for m in md:
        names = ["canonical"] + list(m["mod_long_names"])
        print(f"calls per label ({m['can_base']}): " + "; ".join(f"{n}:{int(c)}" for n, c in zip(names, label_counts[m["can_base"]])))
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(prog="remora_amd")
    sub = ap.add_subparsers(dest="cmd", required=True)
    infer = sub.add_parser("infer").add_subparsers(dest="sub", required=True)
    p = infer.add_parser("from_pod5_and_bam", help="Infer modified bases from POD5 + BAM on an MI355X")
    p.add_argument("pod5")
    p.add_argument("in_bam")
    p.add_argument("--model", required=True, action="append",
                   help="TorchScript model file (with meta.txt); repeat for one model per canonical base")
    p.add_argument("--out-bam", required=True)
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--gpus", type=int, default=1,
                   help="one process per GPU, each takes a contiguous share of the alignments (ranks are started here unless "
                        "already under torchrun); the parts are joined into --out-bam in input order")
    p.add_argument("--procs-per-gpu", type=int, default=1,
                   help="processes per GPU: the host side (BAM / POD5 parsing, per-read arithmetic, MM/ML formatting, BGZF) is "
                        "Python and scales with processes; each takes its own contiguous share of the alignments")
    p.add_argument("--bam-level", type=int, default=None, help="zlib level of the output BAM (default 6, as htslib; 1 = fast: Huffman coding only, about a fifth larger, half the CPU time of zlib's level 1)")
    p.add_argument("--num-reads", type=int, default=None)
    p.add_argument("--reads-per-batch", type=int, default=512)
    p.add_argument("--dtype", default=None, help="fp32 (default) | f16x3 (fp32-class accuracy inside IEEE half's range: inputs and folded weights beyond +-65504 overflow) | "
                        "bf16x6 (fp32 class, fp32 range) | bf16x3 | bf16 | f16")
    p.add_argument("--reference-anchored", action="store_true",
                   help="call at reference positions; output records become <len>M with the reference sequence")
    p.set_defaults(func=_infer)

    val = sub.add_parser("validate").add_subparsers(dest="sub", required=True)
    v = val.add_parser("from_remora_dataset", help="Validate a model on an on-disk Remora dataset (directory or config)")
    v.add_argument("remora_dataset_path")
    v.add_argument("--model", required=True)
    v.add_argument("--out-file", help="validation summary (default stdout)")
    v.add_argument("--full-results-filename", help="per-chunk label, call and probabilities (TSV)")
    v.add_argument("--pct-filt", type=float, default=10.0)
    v.add_argument("--device", type=int, default=0)
    v.add_argument("--gpus", type=int, default=1,
                   help="one process per GPU, each validates a contiguous share of the rows; confusion counts are all-reduced")
    v.add_argument("--procs-per-gpu", type=int, default=1)
    v.add_argument("--batch-size", type=int, default=131072)
    v.add_argument("--dtype", default=None)
    v.set_defaults(func=_validate)

    dset = sub.add_parser("dataset").add_subparsers(dest="sub", required=True)
    d = dset.add_parser("prepare", help="POD5 + BAM -> labelled chunk dataset directory")
    d.add_argument("pod5")
    d.add_argument("bam")
    d.add_argument("--output-path", default="remora_training_dataset")
    d.add_argument("--overwrite", action="store_true")
    d.add_argument("--motif", nargs=2, action="append", metavar=("MOTIF", "FOCUS_POSITION"), required=True)
    d.add_argument("--focus-reference-positions")
    d.add_argument("--chunk-context", default=list(constants.DEFAULT_CHUNK_CONTEXT), type=int, nargs=2)
    d.add_argument("--min-samples-per-base", type=int, default=constants.DEFAULT_MIN_SAMPLES_PER_BASE)
    d.add_argument("--kmer-context-bases", nargs=2, default=list(constants.DEFAULT_KMER_CONTEXT_BASES), type=int)
    d.add_argument("--max-chunks-per-read", type=int, default=15)
    d.add_argument("--base-start-justify", action="store_true")
    d.add_argument("--offset", default=0, type=int)
    d.add_argument("--num-reads", type=int)
    d.add_argument("--basecall-anchor", action="store_true")
    d.add_argument("--reverse-signal", action="store_true")
    d.add_argument("--save-every", default=100_000, type=int)
    d.add_argument("--skip-shuffle", action="store_true")
    d.add_argument("--refine-kmer-level-table")
    d.add_argument("--refine-rough-rescale", action="store_true")
    d.add_argument("--refine-scale-iters", default=-1, type=int)
    d.add_argument("--refine-half-bandwidth", default=5, type=int)
    d.add_argument("--refine-algo", default="dwell_penalty", choices=("Viterbi", "dwell_penalty"))
    d.add_argument("--refine-short-dwell-parameters", default=[4, 3, 0.5], type=float, nargs=3)
    d.add_argument("--rough-rescale-method", default="least_squares", choices=("least_squares", "theil_sen"))
    d.add_argument("--mod-base", nargs=2, metavar=("SHORT_NAME", "LONG_NAME"))
    d.add_argument("--mod-base-control", action="store_true")
    d.add_argument("--reads-per-batch", type=int, default=512)
    d.add_argument("--device", type=int, default=0)
    d.add_argument("--gpus", type=int, default=1,
                   help="one process per GPU, each extracts the chunks of its own share of the BAM; the parts become one dataset")
    d.add_argument("--procs-per-gpu", type=int, default=1)
    d.set_defaults(func=_dataset_prepare)
    di = dset.add_parser("inspect", help="Summary of a dataset directory or config")
    di.add_argument("remora_dataset_path")
    di.add_argument("--out-path", help="write the expanded config (with hashes) here")
    di.set_defaults(func=_dataset_inspect)
    dm = dset.add_parser("make_config", help="Config drawing from several datasets at fixed proportions (no data copied)")
    dm.add_argument("out_path")
    dm.add_argument("dataset_paths", nargs="+")
    dm.add_argument("--dataset-weights", type=float, nargs="+")
    dm.set_defaults(func=_dataset_make_config)

    dg = dset.add_parser("merge", help="Copy several datasets into one new core dataset (shuffled)")
    dg.add_argument("out_path")
    dg.add_argument("dataset_paths", nargs="+")
    dg.add_argument("--max-size", type=int)
    dg.add_argument("--overwrite", action="store_true")
    dg.set_defaults(func=_dataset_merge)
    dh = dset.add_parser("head", help="New core dataset from the first chunks of another")
    dh.add_argument("out_path")
    dh.add_argument("in_path")
    dh.add_argument("num_chunks", type=int)
    dh.add_argument("--overwrite", action="store_true")
    dh.set_defaults(func=_dataset_head)
    dc = dset.add_parser("copy", help="Copy a dataset (all its core datasets + a new config) to a new location")
    dc.add_argument("in_path")
    dc.add_argument("out_path")
    dc.add_argument("--overwrite", action="store_true")
    dc.set_defaults(func=_dataset_copy)

    args = ap.parse_args(argv)
    nranks = getattr(args, "gpus", 1) * max(getattr(args, "procs_per_gpu", 1), 1)
    if nranks > 1 and "WORLD_SIZE" not in os.environ:
        from .dist import launch_ranks

        by_scan = os.environ.get("REMORA_AMD_BAM_SHARD", "bytes") == "scan"
        scan_bam = args.in_bam if args.func is _infer and by_scan else None
        return launch_ranks(sys.argv[1:] if argv is None else list(argv), nranks, scan_bam=scan_bam)
    try:
        return args.func(args)
    except RemoraError as e:
        print(f"remora_amd: {e}", file=sys.stderr)
        return 1


if __name__ == "__main__":
    sys.exit(main())
