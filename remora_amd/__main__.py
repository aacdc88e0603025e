"""`python -m remora_amd infer from_pod5_and_bam POD5 BAM --model MODEL.pt --out-bam OUT.bam`
— the sub-command of the reference CLI that sits on the hot path (src/remora/parsers.py:1294-1417,
:1582-1612), same positional arguments and the same meaning of --model / --out-bam / --device /
--num-reads; and `python -m remora_amd validate from_remora_dataset DATASET_DIR --model MODEL.pt`
(:1800-1960): the stored chunks of an on-disk dataset through the model with the model's chunk / k-mer
contexts, accuracy and confusion matrix against the stored labels."""
import argparse
import sys

from . import RemoraError


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
    p.add_argument("--num-reads", type=int, default=None)
    p.add_argument("--reads-per-batch", type=int, default=256)
    p.add_argument("--dtype", default=None, help="fp32 (default) | bf16x6 | bf16x3 | bf16")
    p.add_argument("--reference-anchored", action="store_true",
                   help="call at reference positions; output records become <len>M with the reference sequence")
    val = sub.add_parser("validate").add_subparsers(dest="sub", required=True)
    v = val.add_parser("from_remora_dataset", help="Validate a model on an on-disk Remora chunk dataset")
    v.add_argument("remora_dataset_path")
    v.add_argument("--model", required=True)
    v.add_argument("--device", type=int, default=0)
    v.add_argument("--batch-size", type=int, default=131072)
    v.add_argument("--dtype", default=None)
    args = ap.parse_args(argv)
    if args.cmd == "validate":
        from .data_chunks import CoreRemoraDataset, validate_dataset
        from .model_util import load_torchscript_model

        try:
            model, md = load_torchscript_model(args.model, device=args.device, eval_only=True, dtype=args.dtype)
            ds = CoreRemoraDataset(args.remora_dataset_path, batch_size=args.batch_size,
                                   override_metadata={"kmer_context_bases": md["kmer_context_bases"],
                                                      "chunk_context": md["chunk_context"]})
            res = validate_dataset(ds, model)
        except RemoraError as e:
            print(f"remora_amd: {e}", file=sys.stderr)
            return 1
        print(f"chunks {ds.size}\tacc {res['acc']:.6f}")
        print("predicted label counts\t" + "\t".join(str(int(c)) for c in res["pred_counts"]))
        print("confusion (rows = stored label, columns = call)")
        for row in res["confusion"]:
            print("\t".join(str(int(c)) for c in row))
        return 0

    from .inference import infer_from_pod5_and_bam
    from .model_util import load_torchscript_model

    try:
        loaded = [load_torchscript_model(m, device=args.device, eval_only=True, dtype=args.dtype) for m in args.model]
        model, md = [x[0] for x in loaded], [x[1] for x in loaded]
        if len({m["can_base"] for m in md}) != len(md):
            raise RemoraError("Only one model per canonical base allowed.")
        stats = infer_from_pod5_and_bam(args.pod5, args.in_bam, model, md, args.out_bam, num_reads=args.num_reads,
                                        reads_per_batch=args.reads_per_batch, ref_anchored=args.reference_anchored)
    except RemoraError as e:
        print(f"remora_amd: {e}", file=sys.stderr)
        return 1
    ok = stats.pop(None, 0)
    print(f"called {ok} reads -> {args.out_bam}")
    for reason, cnt in sorted(stats.items(), key=lambda kv: -kv[1]):
        print(f"{cnt:>7} : {reason}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
