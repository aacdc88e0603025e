#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py -m gpu -q -k "prepare or dataset" > $O/pytest_call11.txt 2>&1; echo "rc=$?" >> $O/pytest_call11.txt; tail -4 $O/pytest_call11.txt | cut -c1-250
export RMR_INFER_TIMING=1
( timeout 400 python tests/manual/prof_prepare_cli.py 12000 1,6 ) > $O/prepare_cli_batch_168k_setorder.log 2>&1; grep -v amdgpu $O/prepare_cli_batch_168k_setorder.log | tail -12
( RMR_PY_GLUE=0 timeout 400 python tests/manual/prof_prepare_cli.py 12000 1 ) > $O/prepare_cli_batch_168k_interp_set.log 2>&1; grep -v amdgpu $O/prepare_cli_batch_168k_interp_set.log | tail -4
