# packed fp32 VALU results beside synthetic co-tenants (tools/ubench/neighbour.hip): which load disturbs them?
B=tools/ubench/bin
mkdir -p $B
[ -x $B/neighbour ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/neighbour.hip -o $B/neighbour
[ -x $B/pk_f32_check ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/ubench/pk_f32_check.hip -o $B/pk_f32_check
echo "--- alone"; $B/pk_f32_check 4000
for kind in valu lds mfma32 mfma16; do
  $B/neighbour $kind 14000 2 32 > /dev/null & P1=$!
  $B/neighbour $kind 14000 2 32 > /dev/null & P2=$!
  sleep 2
  echo "--- beside 2 x $kind"; $B/pk_f32_check 8000
  kill $P1 $P2 2>/dev/null; wait $P1 $P2 2>/dev/null
done
