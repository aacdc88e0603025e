#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
: > $O/ab_reads_stagers_final.log
for rep in 1 2; do
for S in 2 1 3; do
  echo "== RMR_READS_STAGERS=$S" >> $O/ab_reads_stagers_final.log
  RMR_READS_STAGERS=$S timeout 200 python tools/ab_reads.py --dtypes bf16 --calls 15 2>&1 | grep -E "batched" | cut -c1-110 >> $O/ab_reads_stagers_final.log
done
done
cat $O/ab_reads_stagers_final.log
