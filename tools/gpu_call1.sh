set -x
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_rates tools/ubench/valu_rates.hip && /tmp/valu_rates > gpurun_out/ubench_valu.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.txt
for dt in bf16 bf16x3 fp32; do
  timeout 300 python bench.py --dtype $dt --no-cpu-baseline --no-encode --no-reads --no-alt --no-refine --steps 5 --warmup 2 > gpurun_out/base_$dt.json 2> gpurun_out/base_$dt.err
done
timeout 300 python bench.py --dtype bf16 --workload convlstm_c200 --no-cpu-baseline --no-encode --no-reads --no-alt --no-refine --steps 5 --warmup 2 > gpurun_out/base_c200_bf16.json 2> gpurun_out/base_c200_bf16.err
tail -3 gpurun_out/pytest_gpu.txt; cat gpurun_out/ubench_valu.txt
