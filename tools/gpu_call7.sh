set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err ) 2> gpurun_out/bench_full.time
tail -3 gpurun_out/bench_full.time; grep "^\[bench" gpurun_out/bench_full.err; wc -l gpurun_out/bench_full.json
timeout 600 python -m pytest tests/test_gpu_bench.py -x -q -k "refuses or two_ranks" 2>&1 | tail -3
