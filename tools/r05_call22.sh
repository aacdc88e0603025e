#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "device_reads" > $O/pytest_call22.txt 2>&1; echo "rc=$?" >> $O/pytest_call22.txt; tail -3 $O/pytest_call22.txt | cut -c1-200
: > $O/ab_reads_narrow_maps2.log
for rep in 1 2 3; do
for N in 0 1; do
  echo "== RMR_READS_NARROW_MAPS=$N" >> $O/ab_reads_narrow_maps2.log
  RMR_READS_NARROW_MAPS=$N timeout 300 python tools/ab_reads.py --dtypes bf16 --calls 21 2>&1 | grep -E "batched" >> $O/ab_reads_narrow_maps2.log
done
done
cat $O/ab_reads_narrow_maps2.log | cut -c1-120
