#!/bin/bash
# Round-5 second GPU call: the new native single-read entry and the C-API staging walk (tests + timeline), the jitter build
# on every pipeline, and the packed-fp32 experiments beside a bf16-MFMA tenant.
set -u
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "call_read or call_reads or batched or streamed or real_read" > $O/pytest_call2.txt 2>&1; echo "rc=$?" >> $O/pytest_call2.txt
tail -4 $O/pytest_call2.txt
timeout 300 python tools/stress_determinism.py --jitter "fp32,bf16,f16,f16x3,bf16x3,bf16x6,fp32:C100:conv_only,bf16:C200,fp32:C200" --reps 60 --n 40000 > $O/jitter_all_pipelines.log 2>&1; tail -12 $O/jitter_all_pipelines.log
timeout 300 python tools/timeline_reads.py --out $O/timeline_host_glue.md > $O/timeline_host_glue.log 2>&1; tail -7 $O/timeline_host_glue.log
# ---- packed fp32 fed from LDS: stand-alone candidate reproducer
U=tools/ubench/bin
N=$U/neighbour
[ -x $N ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/neighbour.hip -o $N
{
  for P in 0 1; do $U/pk_lds_repro 4000 $P; $U/pk_lds_repro_scalar 4000 $P; done
  $N mfma16 30000 2 32 > /dev/null & P1=$!
  $N mfma16 30000 2 32 > /dev/null & P2=$!
  sleep 1
  echo "beside 2 x neighbour mfma16:"
  $U/pk_lds_repro 8000 0; $U/pk_lds_repro_scalar 6000 0; $U/pk_lds_repro 6000 1
  kill $P1 $P2 2>/dev/null; wait $P1 $P2 2>/dev/null
} > $O/pk_lds_repro.log 2>&1
cat $O/pk_lds_repro.log
# ---- the VALU signal producer (RMR_SIG3_MFMA_LSTM=0) in three builds beside the tenant: compiler-scheduled packed, volatile-asm packed, shipped scalar pair
export RMR_SIG3_MFMA_LSTM=0
R=/root/repo/remora_amd
bash tools/ubench/neighbour_matrix.sh mfma16 200 "fp32@REMORA_HIP_LIB=$R/libremora_hip_pk.so" "fp32@REMORA_HIP_LIB=$R/libremora_hip_pk2.so" "fp32" "bf16x6@REMORA_HIP_LIB=$R/libremora_hip_pk.so" > $O/neighbour_mfma16_pk_builds.log 2>&1
cat $O/neighbour_mfma16_pk_builds.log
# ---- the packed build with jittered barriers, ALONE on the GPU: a race would not need the neighbour
RMR_JITTER_LIB=$R/libremora_hip_jitpk.so timeout 300 python tools/stress_determinism.py --jitter "fp32,bf16x6,bf16x3" --reps 100 --n 40000 > $O/jitter_packed_build.log 2>&1; tail -5 $O/jitter_packed_build.log
