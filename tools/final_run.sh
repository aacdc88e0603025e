set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_final.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_final.txt
tail -4 gpurun_out/pytest_gpu_final.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; tail -2 gpurun_out/smoke.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02_final.json 2> gpurun_out/bench_r02_final.err ) 2> gpurun_out/bench_r02_final.time
tail -3 gpurun_out/bench_r02_final.time; grep "^\[bench" gpurun_out/bench_r02_final.err
timeout 600 python bench.py --gpus 2 --force-device 0 --dist-backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-encode --no-reads --no-others --no-refine > gpurun_out/bench_r02_2ranks_1gpu.json 2> gpurun_out/bench_r02_2ranks_1gpu.err; tail -c 600 gpurun_out/bench_r02_2ranks_1gpu.json
