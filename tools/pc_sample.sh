#!/bin/bash
# (Round 3: the boxes of this pool list NO agent with PC-sampling support - `rocprofv3-avail list --pc-sampling` is empty and
# every configuration is refused - so this has produced nothing yet; tools/isa_blocks.py took its place.)
# Run on the GPU box (via gpurun): rocprofv3 PC sampling of bench.py (no counters, no traces in the same pass), then
# tools/pc_sample_agg.py folds the samples per instruction and stall reason.  Usage: tools/pc_sample.sh <tag> [bench args...]
set -u
TAG=${1:-pcs}; shift || true
export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
OUT=gpurun_out/pcs_$TAG
mkdir -p $OUT
BENCH="python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-reads --no-others --no-refine $*"
echo "$BENCH" > $OUT/command.txt
timeout 120 rocprofv3 -L 2>&1 | grep -i -B2 -A12 "pc.sampl\|PC Sampl" | head -60 > $OUT/avail.txt
for METHOD in stochastic host_trap; do
  if [ $METHOD = stochastic ]; then UNIT=cycles; IV=65536; else UNIT=time; IV=50; fi
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $IV \
      --kernel-trace --output-format csv -d /tmp/pcs_$METHOD -- $BENCH > $OUT/bench_$METHOD.json 2> $OUT/$METHOD.err
  echo "$METHOD rc=$?" >> $OUT/command.txt
  find /tmp/pcs_$METHOD -type f | head -20 >> $OUT/command.txt
  python tools/pc_sample_agg.py /tmp/pcs_$METHOD $OUT/$METHOD > $OUT/agg_$METHOD.log 2>&1
done
du -sh $OUT
