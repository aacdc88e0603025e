set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_refine.py tests/test_gpu_bench.py -x -q 2>&1 | tail -4
timeout 300 python tools/prof_reads_kernels.py 2>&1 | grep "batch of"
