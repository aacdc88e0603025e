#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_parity.py tests/test_gpu_shard.py -m gpu -q -k "ingest or prepare or dataset or infer" > $O/pytest_call16.txt 2>&1; echo "rc=$?" >> $O/pytest_call16.txt; tail -3 $O/pytest_call16.txt | cut -c1-250
timeout 300 python tools/prof_ingest_batches.py 4000 ref 512 > $O/prof_ingest_ref_b512_scratch.log 2>&1; grep -v amdgpu $O/prof_ingest_ref_b512_scratch.log | head -3
timeout 300 python tools/prof_ingest_batches.py 6000 can 512 > $O/prof_ingest_can_b512_scratch.log 2>&1; grep -v amdgpu $O/prof_ingest_can_b512_scratch.log | head -3
export RMR_INFER_TIMING=1
( timeout 400 python tests/manual/prof_prepare_cli.py 12000 1,6 ) > $O/prepare_cli_batch_168k_b512.log 2>&1; grep -v amdgpu $O/prepare_cli_batch_168k_b512.log | grep -E "procs/gpu|rank 0"
( timeout 600 python tests/manual/prof_infer_cli.py 24000 1,6 fp32 1 "--reference-anchored" ) 2>&1 | grep -E 'procs/gpu|infer rank 0|identical|records' > $O/infer_cli_ref_anchored_b512.log; cat $O/infer_cli_ref_anchored_b512.log
( timeout 600 python tests/manual/prof_infer_cli.py 24000 1,6 fp32 1 ) 2>&1 | grep -E 'procs/gpu|infer rank 0|identical|records' > $O/infer_cli_b512.log; cat $O/infer_cli_b512.log
