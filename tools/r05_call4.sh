#!/bin/bash
# Round-5 fourth GPU call: jitter build (fixed), aligned-pair reproducer variant, reads pipeline after the reorder / taper /
# streaming gather, single-read entry with the overlapped host geometry.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
U=tools/ubench/bin
{ timeout 30 $U/pk_lds_repro_pairadd 3000 1; timeout 30 $U/pk_lds_repro_pairadd 2000 0; timeout 30 $U/pk_lds_repro 2000 1; } > $O/pk_lds_repro_pairadd.log 2>&1; cat $O/pk_lds_repro_pairadd.log
timeout 300 python tools/stress_determinism.py --jitter "fp32,bf16,f16,f16x3,bf16x3,bf16x6,fp32:C100:conv_only,bf16:C200,fp32:C200" --reps 12 --n 20000 > $O/jitter_all_pipelines.log 2>&1; tail -22 $O/jitter_all_pipelines.log | cut -c1-230
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_ingest.py -m gpu -q -k "call_read or extract_chunk or specified or batched or streamed or subbatch or ingest or real_read or infer" > $O/pytest_call4.txt 2>&1; echo "rc=$?" >> $O/pytest_call4.txt; tail -5 $O/pytest_call4.txt | cut -c1-300
timeout 300 python tools/timeline_reads.py --out $O/timeline_host_call4.md > $O/timeline_host_call4.log 2>&1; tail -7 $O/timeline_host_call4.log
RMR_PACK_STREAM=0 timeout 300 python tools/timeline_reads.py --single 0 --out $O/timeline_host_call4_memcpy.md > $O/timeline_host_call4_memcpy.log 2>&1; tail -5 $O/timeline_host_call4_memcpy.log
RMR_PACK_THREADS=12 timeout 300 python tools/timeline_reads.py --single 0 --out $O/timeline_host_call4_t12.md > $O/timeline_host_call4_t12.log 2>&1; tail -5 $O/timeline_host_call4_t12.log
export RMR_BAM_LEVEL=1 RMR_INFER_TIMING=1
( timeout 600 python tests/manual/prof_infer_cli.py 24000 1 fp32 1 ) 2>&1 | grep -E 'procs/gpu|infer rank 0|records' > $O/infer_cli_prefetch2.log; cat $O/infer_cli_prefetch2.log
