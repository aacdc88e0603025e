#!/bin/bash
# Experiment builds: libremora_hip_<name>.so = the shipped objects with the listed sources recompiled under extra -D flags
# (selected at run time with REMORA_HIP_LIB; *.so are git-ignored but travel with gpurun).
#   tools/build_variant.sh r2fused "k_fused.hip" "-DRMR_FUSED_WAVES_EU=2 -DRMR_FUSED_STREAM_M1=0 -DRMR_FUSED_S3_SPLIT=0 -DRMR_FUSED_S4_LEAN=0"
set -e
name=$1; srcs=$2; flags=$3
root=$(cd "$(dirname "$0")/.." && pwd)
cs=$root/remora_amd/csrc
make -s -j8 -C "$cs"
out=$cs/var_$name
mkdir -p "$out"
objs=""
for o in "$cs"/*.o; do
  b=$(basename "$o" .o)
  if [[ " $srcs " == *" $b.hip "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function $flags -c "$cs/$b.hip" -o "$out/$b.o"
    objs="$objs $out/$b.o"
  else
    objs="$objs $o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/remora_amd/libremora_hip_$name.so" $objs -lz -ldl
echo "built remora_amd/libremora_hip_$name.so"
