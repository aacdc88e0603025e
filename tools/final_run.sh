# Round-end validation on the GPU box (via gpurun): the GPU suite, smoke(), the default bench line + details, a 2-rank bench on
# one GPU.  Usage: tools/final_run.sh <tag>   (outputs under gpurun_out/)
set -x
TAG=${1:-r03}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_final.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_final.txt
grep -E "passed|failed|rc=" gpurun_out/pytest_gpu_final.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; tail -2 gpurun_out/smoke.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 --details gpurun_out/bench_${TAG}_details.json > gpurun_out/bench_${TAG}_final.json 2> gpurun_out/bench_${TAG}_final.err ) 2> gpurun_out/bench_${TAG}_final.time
tail -3 gpurun_out/bench_${TAG}_final.time; grep "^\[bench" gpurun_out/bench_${TAG}_final.err | tail -12; wc -c gpurun_out/bench_${TAG}_final.json
timeout 600 python bench.py --gpus 2 --force-device 0 --dist-backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-encode --no-reads --no-others --no-refine --details gpurun_out/bench_${TAG}_2ranks_details.json > gpurun_out/bench_${TAG}_2ranks_1gpu.json 2> gpurun_out/bench_${TAG}_2ranks_1gpu.err; tail -c 400 gpurun_out/bench_${TAG}_2ranks_1gpu.json
timeout 900 python bench.py --gpus 8 --force-device 0 --dist-backend gloo --chunks 125000 --steps 5 --warmup 2 --no-cpu-baseline --no-encode --no-reads --no-others --no-refine --details gpurun_out/bench_${TAG}_8ranks_details.json > gpurun_out/bench_${TAG}_8ranks_1gpu.json 2> gpurun_out/bench_${TAG}_8ranks_1gpu.err; tail -c 300 gpurun_out/bench_${TAG}_8ranks_1gpu.json; grep "bench rank" gpurun_out/bench_${TAG}_8ranks_1gpu.err | head -16
python tools/time_single_read.py > gpurun_out/single_read_${TAG}.log 2>&1; tail -2 gpurun_out/single_read_${TAG}.log
