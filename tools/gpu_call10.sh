set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q > gpurun_out/pytest_fused.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_fused.txt
tail -5 gpurun_out/pytest_fused.txt
REMORA_HIP_LIB=$PWD/remora_amd/libremora_hip_abl.so timeout 300 python tools/abl_fused.py C100 262144 0,1,64,63,56,62,57,55,47,31 > gpurun_out/abl_c100.txt 2>&1
cat gpurun_out/abl_c100.txt
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
run sb64k RMR_FUSED_SUBBATCH=65536
run fb2 RMR_FUSED_BLOCKS_PER_CU=2 RMR_FUSED_SUBBATCH=65536
B="python bench.py --workload convlstm_c200_bf16 --no-cpu-baseline --no-encode --no-reads --no-others --no-refine --steps 5 --warmup 2"
run c200 RMR_FUSED_SUBBATCH=65536
