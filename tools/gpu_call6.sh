set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_bench.py -x -q > gpurun_out/pytest_bench.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_bench.txt
tail -30 gpurun_out/pytest_bench.txt
( time timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err ) 2> gpurun_out/bench_full.time
tail -3 gpurun_out/bench_full.time; tail -5 gpurun_out/bench_full.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_full.json'))
print('headline', d['value']/1e6, d['dtype'], 'roofline', d['roofline']['kernel'], round(d['roofline']['frac'],3))
print('cpu', {k:(round(v['chunks_per_s']) if 'chunks_per_s' in v else v) for k,v in d['cpu_baseline']['forms'].items()}, d['cpu_baseline']['headline_form'], d['cpu_baseline']['cores'])
print('reads', d.get('reads_per_sec'), d.get('reads_per_sec_derived'))
print('cabi', d.get('allreduce_counts_c_abi'))
for k,v in d.get('other_configs',{}).items():
    print(k, v.get('error') or ('%.1fM %s roofline %s %.3f'%(v['value']/1e6, v['dtype'], v['roofline']['kernel'], v['roofline']['frac'])))
print('precision', d.get('precision'))
PY
