set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_cycles tools/ubench/valu_cycles.hip && /tmp/valu_cycles > gpurun_out/ubench_cycles.txt 2>&1
cat gpurun_out/ubench_cycles.txt
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q > gpurun_out/pytest_fused.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_fused.txt
tail -5 gpurun_out/pytest_fused.txt
REMORA_HIP_LIB=$PWD/remora_amd/libremora_hip_abl.so timeout 300 python tools/abl_fused.py C100 262144 0,1,64,63,56,31 > gpurun_out/abl_c100.txt 2>&1
cat gpurun_out/abl_c100.txt
B="python bench.py --dtype bf16 --no-cpu-baseline --no-encode --no-reads --no-alt --no-refine --steps 5 --warmup 2"
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
run res00 RMR_X=0
# residency variants of the fused kernel, built here
for v in "0 1" "1 0"; do
  set -- $v
  mkdir -p /tmp/v$1$2 && cp -r remora_amd/csrc /tmp/v$1$2/ && cp -r include /tmp/v$1$2/
  (cd /tmp/v$1$2/csrc && sed -i 's#../../include#../include#g' rmr_internal.h Makefile && rm -f k_fused.o && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -DRMR_FUSED_RES_SMALL=$1 -DRMR_FUSED_RES_MID=$2" OUT=/tmp/v$1$2/lib.so > /tmp/v$1$2/build.log 2>&1; tail -2 /tmp/v$1$2/build.log)
  run res$1$2 REMORA_HIP_LIB=/tmp/v$1$2/lib.so
done
