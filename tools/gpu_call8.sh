set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -k "fused or bf16" > gpurun_out/pytest_fused.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_fused.txt
tail -5 gpurun_out/pytest_fused.txt
B="python bench.py --dtype bf16 --no-cpu-baseline --no-encode --no-reads --no-others --no-refine --steps 5 --warmup 2"
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 $B > gpurun_out/f_$name.json 2> gpurun_out/f_$name.err
  python - $name <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/f_%s.json'%sys.argv[1]))
    print(sys.argv[1], 'value %.1fM'%(d['value']/1e6), ' '.join('%s=%.3f'%(k,v['avg_ms']) for k,v in d['kernels'].items()))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
}
run default RMR_X=0
run lx2 RMR_LSTMX_BLOCKS_PER_CU=2
run lx8 RMR_LSTMX_BLOCKS_PER_CU=8
run sb64k RMR_FUSED_SUBBATCH=65536
run sb128k RMR_FUSED_SUBBATCH=131072
B="python bench.py --workload convlstm_c200_bf16 --no-cpu-baseline --no-encode --no-reads --no-others --no-refine --steps 5 --warmup 2"
run c200 RMR_X=0
