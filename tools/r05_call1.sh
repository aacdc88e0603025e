#!/bin/bash
# Round-5 first GPU call: where the reads pipeline and the file-to-file paths stand at the round's start.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
TAG=${1:-base}
timeout 300 python tools/timeline_reads.py --out $O/timeline_host_$TAG.md > $O/timeline_host_$TAG.log 2>&1
timeout 600 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $O/tl_trace_$TAG -- python tools/timeline_reads.py --out $O/timeline_prof_$TAG.md > $O/timeline_prof_$TAG.log 2>&1
python tools/timeline_summary.py $O/tl_trace_$TAG $O/timeline_prof_${TAG}_windows.json > $O/timeline_gpu_$TAG.md 2> $O/timeline_gpu_$TAG.err
ls -la $O/tl_trace_$TAG/*/ > $O/tl_trace_$TAG.ls 2>&1; head -3 $O/tl_trace_$TAG/*/*hip_api_trace.csv $O/tl_trace_$TAG/*/*memory_copy_trace.csv >> $O/tl_trace_$TAG.ls 2>&1
rm -rf $O/tl_trace_$TAG
timeout 300 python tools/prof_single_read.py > $O/single_read_$TAG.txt 2>&1
tail -5 $O/timeline_host_$TAG.log; tail -30 $O/timeline_gpu_$TAG.md | head -5
