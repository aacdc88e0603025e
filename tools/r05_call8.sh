#!/bin/bash
# Round-5 eighth GPU call: the reference-anchored batch ingest (tests, then file-to-file rates with and without it).
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_parity.py -m gpu -q -k "ingest or reference_anchored or real_read or infer or ref_anch" > $O/pytest_call8.txt 2>&1; echo "rc=$?" >> $O/pytest_call8.txt; tail -30 $O/pytest_call8.txt | cut -c1-250
export RMR_BAM_LEVEL=1 RMR_INFER_TIMING=1
( timeout 600 python tests/manual/prof_infer_cli.py 6000 1,6 fp32 1 "--reference-anchored" ) 2>&1 | grep -E 'procs/gpu|infer rank 0|identical|records' > $O/infer_cli_ref_anchored_batch.log; cat $O/infer_cli_ref_anchored_batch.log
( RMR_INFER_BATCH_INGEST=0 timeout 600 python tests/manual/prof_infer_cli.py 6000 1,6 fp32 1 "--reference-anchored" ) 2>&1 | grep -E 'procs/gpu|infer rank 0|identical|records' > $O/infer_cli_ref_anchored_per_read.log; cat $O/infer_cli_ref_anchored_per_read.log
timeout 300 python tools/ab_reads.py > $O/ab_reads_final.log 2>&1; grep -v amdgpu $O/ab_reads_final.log
