# Round-6 rocprofv3 passes (kernel trace + the separate PMC passes of tools/profile_gpu.sh) for the configurations the bench
# reports, rewriting profiles/traffic.json for each.  Usage (on the GPU box): COMMIT=<sha> bash tools/r06_profiles.sh
set -u
cp profiles/traffic.json gpurun_out/traffic.json
TRAFFIC_KEY=fp32 tools/profile_gpu.sh r06_fp32
TRAFFIC_KEY=bf16 tools/profile_gpu.sh r06_bf16 --dtype bf16
TRAFFIC_KEY=f16 tools/profile_gpu.sh r06_f16 --dtype f16
TRAFFIC_KEY=f16x3 tools/profile_gpu.sh r06_f16x3 --dtype f16x3
TRAFFIC_KEY=conv_c100:fp32 tools/profile_gpu.sh r06_conv_c100 --workload conv_c100
TRAFFIC_KEY=convlstm_c200_bf16:bf16 tools/profile_gpu.sh r06_c200_bf16 --workload convlstm_c200_bf16
TRAFFIC_KEY=convlstm_c100_s128:fp32 tools/profile_gpu.sh r06_s128 --workload convlstm_c100_s128
for t in fp32 bf16 f16 f16x3 conv_c100 c200_bf16 s128; do
  d=gpurun_out/prof_r06_$t
  rm -rf $d/trace $d/pmc_* 2>/dev/null  # the raw traces stay on the box: the summaries and the kernel stats come back
done
du -sh gpurun_out
