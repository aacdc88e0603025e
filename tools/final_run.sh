set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_final.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_final.txt
tail -4 gpurun_out/pytest_gpu_final.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; tail -2 gpurun_out/smoke.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r02_final.json 2> gpurun_out/bench_r02_final.err ) 2> gpurun_out/bench_r02_final.time
tail -3 gpurun_out/bench_r02_final.time; grep "^\[bench" gpurun_out/bench_r02_final.err
timeout 600 python bench.py --gpus 2 --force-device 0 --dist-backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-encode --no-reads --no-others --no-refine > gpurun_out/bench_r02_2ranks_1gpu.json 2> gpurun_out/bench_r02_2ranks_1gpu.err; tail -c 600 gpurun_out/bench_r02_2ranks_1gpu.json
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r02_reads; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02_reads -- python tools/prof_reads_kernels.py > gpurun_out/prof_r02_reads.log 2> gpurun_out/prof_r02_reads.err
grep -E "batch of|kernel sum" gpurun_out/prof_r02_reads.log
cp gpurun_out/prof_r02_reads/*/*_kernel_stats.csv gpurun_out/prof_r02_reads_kernel_stats.csv
python tools/prof_single_read.py 2>&1 | grep -E "ms per read|us per read|kernel sum" > gpurun_out/single_read.txt; head -3 gpurun_out/single_read.txt
