#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 300 python tools/prof_ingest_batches.py 4000 ref 256 > $O/prof_ingest_ref_b256.log 2>&1; grep -v amdgpu $O/prof_ingest_ref_b256.log | head -45
timeout 300 python tools/prof_ingest_batches.py 4000 ref 1024 > $O/prof_ingest_ref_b1024.log 2>&1; grep -v amdgpu $O/prof_ingest_ref_b1024.log | head -4
export RMR_INFER_TIMING=1
( timeout 400 python tests/manual/prof_prepare_cli.py 12000 1 ) > $O/prepare_cli_batch_168k_lightcount.log 2>&1; grep -v amdgpu $O/prepare_cli_batch_168k_lightcount.log | tail -4
