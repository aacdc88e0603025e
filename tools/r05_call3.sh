#!/bin/bash
# Round-5 third GPU call: reproducer variants, the calibrated jitter build, single-read entry with host geometry, BAM reader
# with the inflate running ahead of the parser, bench with per-config parity.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
U=tools/ubench/bin
{
  for V in "" _scalar _splat _reg _nop0 _nop7; do timeout 30 $U/pk_lds_repro$V 3000 1; done
  for V in "" _splat _reg _nop7; do timeout 30 $U/pk_lds_repro$V 2000 0; done
} > $O/pk_lds_repro_variants.log 2>&1
cat $O/pk_lds_repro_variants.log
timeout 240 python tools/stress_determinism.py --jitter "fp32,bf16,f16,f16x3,bf16x3,bf16x6,fp32:C100:conv_only,bf16:C200,fp32:C200" --reps 12 --n 20000 > $O/jitter_all_pipelines.log 2>&1; tail -24 $O/jitter_all_pipelines.log | cut -c1-220
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -m gpu -q -x -k "call_read or extract_chunk or specified or batched" > $O/pytest_call3.txt 2>&1; echo "rc=$?" >> $O/pytest_call3.txt; tail -3 $O/pytest_call3.txt
timeout 300 python tools/timeline_reads.py --out $O/timeline_host_call3.md > $O/timeline_host_call3.log 2>&1; tail -7 $O/timeline_host_call3.log
export RMR_BAM_LEVEL=1 RMR_INFER_TIMING=1
( timeout 600 python tests/manual/prof_infer_cli.py 24000 1,6 fp32 1 ) 2>&1 | grep -E 'procs/gpu|infer rank 0|identical|records' > $O/infer_cli_prefetch.log; cat $O/infer_cli_prefetch.log
( time timeout 600 python bench.py --steps 20 --warmup 5 --details $O/bench_call3_details.json > $O/bench_call3.json 2> $O/bench_call3.err ) 2> $O/bench_call3.time; tail -3 $O/bench_call3.time; grep "^\[bench" $O/bench_call3.err | tail -14; cut -c1-600 $O/bench_call3.json
