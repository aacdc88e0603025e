#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py -m gpu -q -k "prepare or dataset" > $O/pytest_call10.txt 2>&1; echo "rc=$?" >> $O/pytest_call10.txt; tail -4 $O/pytest_call10.txt | cut -c1-250
export RMR_INFER_TIMING=1
( timeout 600 python tests/manual/prof_prepare_cli.py 12000 1,6 ) > $O/prepare_cli_batch_168k.log 2>&1; grep -v amdgpu $O/prepare_cli_batch_168k.log | tail -4
( RMR_PREPARE_BATCH_INGEST=0 timeout 600 python tests/manual/prof_prepare_cli.py 12000 1,6 ) > $O/prepare_cli_per_read_168k.log 2>&1; grep -v amdgpu $O/prepare_cli_per_read_168k.log | tail -4
