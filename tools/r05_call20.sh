#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
: > $O/ab_reads_lead_cuts.log
for rep in 1 2; do
for C in "" "64,192" "64,128,256" "32,96,256"; do
  echo "== lead cuts '$C'" >> $O/ab_reads_lead_cuts.log
  RMR_READS_LEAD_CUTS=$C timeout 300 python tools/ab_reads.py --calls 9 2>&1 | grep -E "batched" >> $O/ab_reads_lead_cuts.log
done
done
cat $O/ab_reads_lead_cuts.log
