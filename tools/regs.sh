#!/bin/bash
# Register / spill / scratch report of every kernel in one .hip file (cross-compiled for gfx950, no GPU needed):
#   tools/regs.sh remora_amd/csrc/k_fused.hip [-DNAME=VALUE ...]
set -e
src=$(realpath "$1"); shift
tmp=$(mktemp -d /tmp/regs.XXXXXX)
cd "$tmp"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -I/root/repo/include -I"$(dirname "$src")" "$@" -c "$src" -o out.o -save-temps 2>&1 | grep -v "^$" | head -20
awk '/^    \.name:/ {n=$2} /sgpr_spill_count|vgpr_count|vgpr_spill_count|private_segment_fixed_size|agpr_count/ {printf "%s %s %s\n", n, $1, $2}' *gfx950.s | \
  awk '{k[$1]=k[$1]" "$2$3} END {for (n in k) print n": "k[n]}' | sed 's/_ZN3rmr//' | sort
rm -rf "$tmp"
