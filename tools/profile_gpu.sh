#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py (MI355X_MICROARCH.md: counters
# in their own passes, never combined with sys/hip traces).
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r2}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reads --no-others --no-refine $*"
echo "$BENCH" > $OUT/command.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- $BENCH > $OUT/bench_$C.json 2> $OUT/pmc_$C.err
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_mfma -- $BENCH > $OUT/bench_mfma.json 2> $OUT/pmc_mfma.err
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_lds -- $BENCH > $OUT/bench_lds.json 2> $OUT/pmc_lds.err
python tools/summarize_profile.py $OUT > $OUT/summary.md 2> $OUT/summary.err
# profiles/traffic.json (what bench.py's roofline.traffic is scaled from): rewritten for this workload:dtype from the two PMC
# passes above, with the hash of every kernel source.  TRAFFIC_KEY = the table's key (dtype, or workload:dtype), e.g.
#   COMMIT=$(git rev-parse --short HEAD) TRAFFIC_KEY=bf16 tools/profile_gpu.sh r04_bf16 --dtype bf16   (then copy gpurun_out/traffic.json to profiles/)
if [ -n "${TRAFFIC_KEY:-}" ]; then
  CPL=$(python - "$OUT/bench_FETCH_SIZE.json" <<'PY'
import json, sys
line = [ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1]
print(json.loads(line)["roofline"]["chunks_per_launch"])
PY
)
  [ -f gpurun_out/traffic.json ] || cp profiles/traffic.json gpurun_out/traffic.json
  python tools/summarize_profile.py $OUT --traffic "$TRAFFIC_KEY" "$CPL" gpurun_out/traffic.json "${COMMIT:-worktree}"
fi
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
du -sh $OUT
