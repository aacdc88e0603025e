set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.txt
tail -5 gpurun_out/pytest_gpu.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-encode --no-others --no-refine > gpurun_out/bench_reads.json 2> gpurun_out/bench_reads.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_reads.json'))
print('value', d['value']/1e6, 'reads', d['reads_pipeline'])
PY
