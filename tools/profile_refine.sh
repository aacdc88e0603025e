#!/bin/bash
# Run on the GPU box (via gpurun): two batch sizes, kernel-trace stats and PMC passes for the
# signal-mapping refinement kernels (tools/bench_refine.py).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_refine
mkdir -p $OUT
timeout 300 python tools/bench_refine.py --reads 16384 2>/dev/null | tail -1 >> $OUT/sweep.log
timeout 300 python tools/bench_refine.py --reads 2048 2>/dev/null | tail -1 >> $OUT/sweep.log
BENCH="python tools/bench_refine.py --reads 16384 --steps 2"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_inst -- $BENCH > $OUT/bench_inst.json 2> $OUT/pmc_inst.err
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/pmc_act -- $BENCH > $OUT/bench_act.json 2> $OUT/pmc_act.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- $BENCH > $OUT/bench_$C.json 2> $OUT/pmc_$C.err
done
find $OUT -name "*.csv" -size +20M -delete
du -sh $OUT
