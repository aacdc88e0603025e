#!/bin/bash
# Round-5 profile passes on the round's final kernels: rocprofv3 kernel-trace stats + separate PMC passes, fp32 (headline) and bf16
set -u
cp profiles/traffic.json gpurun_out/traffic.json
COMMIT=0c78894 TRAFFIC_KEY=fp32 bash tools/profile_gpu.sh r05_fp32
COMMIT=0c78894 TRAFFIC_KEY=bf16 bash tools/profile_gpu.sh r05_bf16 --dtype bf16
rm -rf gpurun_out/prof_r05_fp32/trace gpurun_out/prof_r05_fp32/pmc_* gpurun_out/prof_r05_bf16/trace gpurun_out/prof_r05_bf16/pmc_*
ls gpurun_out/prof_r05_fp32 gpurun_out/prof_r05_bf16
head -30 gpurun_out/prof_r05_fp32/summary.md
