#!/bin/bash
# Round-5 closing bench on the final code: the default bench line + details, the 2-rank leg on one GPU
set -u
TAG=r05
mkdir -p gpurun_out
( time timeout 900 python bench.py --steps 20 --warmup 5 --details gpurun_out/bench_${TAG}_details.json > gpurun_out/bench_${TAG}_final.json 2> gpurun_out/bench_${TAG}_final.err ) 2> gpurun_out/bench_${TAG}_final.time
tail -3 gpurun_out/bench_${TAG}_final.time; wc -c gpurun_out/bench_${TAG}_final.json
timeout 600 python bench.py --gpus 2 --force-device 0 --dist-backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-encode --no-reads --no-others --no-refine --details gpurun_out/bench_${TAG}_2ranks_details.json > gpurun_out/bench_${TAG}_2ranks_1gpu.json 2> gpurun_out/bench_${TAG}_2ranks_1gpu.err; tail -c 300 gpurun_out/bench_${TAG}_2ranks_1gpu.json
