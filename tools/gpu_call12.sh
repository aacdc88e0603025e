set -x
bash tools/profile_gpu.sh r02_fp32 --dtype fp32
bash tools/profile_gpu.sh r02_bf16 --dtype bf16
bash tools/profile_gpu.sh r02_c200_bf16 --workload convlstm_c200_bf16 --no-encode
ls gpurun_out/prof_r02_fp32 gpurun_out/prof_r02_bf16
cat gpurun_out/prof_r02_bf16/summary.md | head -60
