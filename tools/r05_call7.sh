#!/bin/bash
# Round-5 seventh GPU call: interleaved A/B of the reads paths (side stream, stagers), results fetched with one wait.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_refine.py -m gpu -q -k "call_reads or batched or streamed or subbatch or pipelined" > $O/pytest_call7.txt 2>&1; echo "rc=$?" >> $O/pytest_call7.txt; tail -3 $O/pytest_call7.txt | cut -c1-300
timeout 300 python tools/ab_reads.py > $O/ab_reads_2stagers.log 2>&1; cat $O/ab_reads_2stagers.log | grep -v amdgpu
RMR_READS_STAGERS=1 timeout 300 python tools/ab_reads.py > $O/ab_reads_1stager.log 2>&1; cat $O/ab_reads_1stager.log | grep -v amdgpu
RMR_READS_STAGERS=2 RMR_PACK_THREADS=4 timeout 300 python tools/ab_reads.py --dtypes bf16 > $O/ab_reads_2stagers_4threads.log 2>&1; cat $O/ab_reads_2stagers_4threads.log | grep -v amdgpu
timeout 300 python tools/timeline_reads.py --single 0 --out $O/timeline_host_call7.md > $O/timeline_host_call7.log 2>&1; tail -5 $O/timeline_host_call7.log
