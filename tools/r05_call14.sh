#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "native_call_read or call_read" > $O/pytest_call14.txt 2>&1; echo "rc=$?" >> $O/pytest_call14.txt; tail -4 $O/pytest_call14.txt | cut -c1-250
: > $O/ab_single_read_zero_copy.log
for z in 0 7 1 3 4 0 7; do RMR_CALL_READ_ZERO_COPY=$z timeout 200 python tools/ab_single_read.py 2>&1 | grep -E "median|Error|error" >> $O/ab_single_read_zero_copy.log; done
cat $O/ab_single_read_zero_copy.log
