#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "device_reads or reads_mods or call_reads or motif or streamed or refiner" > $O/pytest_call21.txt 2>&1; echo "rc=$?" >> $O/pytest_call21.txt; tail -3 $O/pytest_call21.txt | cut -c1-200
: > $O/ab_reads_narrow_maps.log
for rep in 1 2; do
for N in 0 1; do
  echo "== RMR_READS_NARROW_MAPS=$N" >> $O/ab_reads_narrow_maps.log
  RMR_READS_NARROW_MAPS=$N timeout 300 python tools/ab_reads.py --calls 9 2>&1 | grep -E "batched" >> $O/ab_reads_narrow_maps.log
done
done
cat $O/ab_reads_narrow_maps.log
