#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ingest.py -m gpu -q -k "motif or reads or prepare or ingest" > $O/pytest_call19.txt 2>&1; echo "rc=$?" >> $O/pytest_call19.txt; tail -3 $O/pytest_call19.txt | cut -c1-200
: > $O/ab_reads_motif_block.log
for rep in 1 2; do
  for L in libremora_hip_motif1.so libremora_hip.so; do
    echo "== $L" >> $O/ab_reads_motif_block.log
    REMORA_HIP_LIB=$PWD/remora_amd/$L timeout 300 python tools/ab_reads.py --calls 7 2>&1 | grep -E "batched|single" >> $O/ab_reads_motif_block.log
  done
done
cat $O/ab_reads_motif_block.log
