set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
REMORA_HIP_LIB=$PWD/remora_amd/libremora_hip_abl.so timeout 300 python tools/abl_fused.py C100 262144 > gpurun_out/abl_c100.txt 2>&1
cat gpurun_out/abl_c100.txt
OUT=gpurun_out/prof_fused1
mkdir -p $OUT
BENCH="python bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-encode --no-reads --no-alt --no-refine"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_lds -- $BENCH > $OUT/bench_lds.json 2> $OUT/pmc_lds.err
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_mfma -- $BENCH > $OUT/bench_mfma.json 2> $OUT/pmc_mfma.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc_misc -- $BENCH > $OUT/bench_misc.json 2> $OUT/pmc_misc.err
python tools/summarize_profile.py $OUT > gpurun_out/prof_fused1.md 2>&1
cat gpurun_out/prof_fused1.md
tail -3 $OUT/pmc_misc.err
