#!/bin/bash
# File-to-file rates of the host-bound pipelines on a GPU box (via gpurun): what profiles/NOTES_r04.md section 8 lists as
# "changed after the GPU minutes were spent".  Usage: tools/measure_host_paths.sh <tag> [records/14 = 24000]
#   - infer from_pod5_and_bam, 1 and 6 processes on one GPU, --bam-level 1, with the library's BAM codecs and with zlib
#   - infer --reference-anchored on the same file
#   - dataset prepare (reference-anchored), 1 and 6 processes, with and without the batch ingest
# Logs under gpurun_out/<tag>_*.log (copy what is to be kept into profiles/).
set -u
TAG=${1:-r05}; REP=${2:-24000}
mkdir -p gpurun_out
export RMR_BAM_LEVEL=1 RMR_INFER_TIMING=1
( time timeout 900 python tests/manual/prof_infer_cli.py $REP 6,1 fp32 1 ) 2>&1 | grep -E 'procs/gpu|infer rank 0|identical|records|real' > gpurun_out/${TAG}_infer_cli.log
( RMR_FAST_INFLATE=0 RMR_BGZF_NATIVE=0 timeout 900 python tests/manual/prof_infer_cli.py $REP 6,1 fp32 1 ) 2>&1 | grep -E 'procs/gpu|infer rank 0|records' > gpurun_out/${TAG}_infer_cli_zlib.log
( timeout 900 python tests/manual/prof_infer_cli.py $REP 6,1 fp32 1 "--reference-anchored" ) 2>&1 | grep -E 'procs/gpu|infer rank 0|identical|records' > gpurun_out/${TAG}_infer_cli_ref_anchored.log
( timeout 900 python tests/manual/prof_prepare_cli.py $((REP / 2)) 1,6 ) 2>&1 | grep -v amdgpu | tail -12 > gpurun_out/${TAG}_prepare_cli.log
( RMR_PREPARE_BATCH_INGEST=0 timeout 900 python tests/manual/prof_prepare_cli.py $((REP / 8)) 1,6 ) 2>&1 | grep -v amdgpu | tail -4 > gpurun_out/${TAG}_prepare_cli_per_read.log
tail -n 4 gpurun_out/${TAG}_infer_cli.log gpurun_out/${TAG}_infer_cli_zlib.log gpurun_out/${TAG}_infer_cli_ref_anchored.log gpurun_out/${TAG}_prepare_cli.log gpurun_out/${TAG}_prepare_cli_per_read.log
