#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + HBM counters for the data-side kernels
# (normalise / geometry / fill / motif scan / VBZ decode / encode) on the reads pipeline and the VBZ bench.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_data
mkdir -p $OUT
W1="python tools/prof_reads_kernels.py"
W2="python tools/bench_vbz.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_reads -- $W1 > $OUT/reads.log 2> $OUT/trace_reads.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_vbz -- $W2 > $OUT/vbz.log 2> $OUT/trace_vbz.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_reads_$C -- $W1 > /dev/null 2> $OUT/pmc_reads_$C.err
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_vbz_$C -- $W2 > /dev/null 2> $OUT/pmc_vbz_$C.err
done
find $OUT -name "*.csv" -size +20M -delete
du -sh $OUT
