#!/bin/bash
# Round-5 sixth GPU call: two stagers, side-stream sequence branch for small batches, motif scan four windows per round,
# ingest with the translated base codes; the build without SLP vectorisation is the shipped one now.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_conv_front.py tests/test_gpu_ingest.py tests/test_gpu_jitter.py -m gpu -q -k "call_read or call_reads or extract or specified or batched or streamed or subbatch or ingest or real_read or infer or motif or focus or side_stream or folded or jitter" > $O/pytest_call6.txt 2>&1; echo "rc=$?" >> $O/pytest_call6.txt; tail -5 $O/pytest_call6.txt | cut -c1-300
timeout 300 python tools/timeline_reads.py --out $O/timeline_host_call6.md > $O/timeline_host_call6.log 2>&1; tail -7 $O/timeline_host_call6.log
RMR_READS_STAGERS=1 timeout 300 python tools/timeline_reads.py --single 0 --dtypes bf16 --out $O/timeline_host_call6_1stager.md > $O/timeline_host_call6_1stager.log 2>&1; tail -3 $O/timeline_host_call6_1stager.log
timeout 200 python tools/prof_ingest_batches.py 6000 > $O/prof_ingest_batches2.log 2>&1; sed -n 1,14p $O/prof_ingest_batches2.log | cut -c1-200
export RMR_BAM_LEVEL=1 RMR_INFER_TIMING=1
( timeout 600 python tests/manual/prof_infer_cli.py 24000 1,6 fp32 1 ) 2>&1 | grep -E 'procs/gpu|infer rank 0|identical|records' > $O/infer_cli_call6.log; cat $O/infer_cli_call6.log
