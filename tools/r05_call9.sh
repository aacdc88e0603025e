#!/bin/bash
# Round-5 ninth GPU call: dataset prepare on the batch ingest (goldens + A/B + the rank test), the jitter test with the
# rebuilt library, the prepare CLI's rate with and without the batch ingest.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py tests/test_gpu_jitter.py tests/test_gpu_refine.py -m gpu -q -k "prepare or dataset or jitter or extract_chunks_reference" > $O/pytest_call9.txt 2>&1; echo "rc=$?" >> $O/pytest_call9.txt; tail -25 $O/pytest_call9.txt | cut -c1-250
( timeout 600 python tests/manual/prof_prepare_cli.py 3000 1,6 ) > $O/prepare_cli_batch.log 2>&1; grep -v amdgpu $O/prepare_cli_batch.log | tail -4
( RMR_PREPARE_BATCH_INGEST=0 timeout 600 python tests/manual/prof_prepare_cli.py 3000 1,6 ) > $O/prepare_cli_per_read.log 2>&1; grep -v amdgpu $O/prepare_cli_per_read.log | tail -4
