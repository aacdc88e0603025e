set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/prof_fused2
mkdir -p $OUT
BENCH="python bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-encode --no-reads --no-others --no-refine"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_lds -- $BENCH > $OUT/bench_lds.json 2> $OUT/pmc_lds.err
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_mfma -- $BENCH > $OUT/bench_mfma.json 2> $OUT/pmc_mfma.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS --kernel-trace --output-format csv -d $OUT/pmc_misc -- $BENCH > $OUT/bench_misc.json 2> $OUT/pmc_misc.err
python tools/summarize_profile.py $OUT > gpurun_out/prof_fused2.md 2>&1
cat gpurun_out/prof_fused2.md
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('gpurun_out/prof_fused2/pmc_misc/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][-40:]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items(): print(k, dict(v))
PY
tail -2 $OUT/pmc_misc.err
