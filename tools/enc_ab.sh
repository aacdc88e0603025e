# A/B of the encode kernel's forms on one box: shipped library vs libremora_hip_sel.so (per-element selects), twice each
for lib in "" sel "" sel; do
  if [ -n "$lib" ]; then export REMORA_HIP_LIB=$PWD/remora_amd/libremora_hip_$lib.so; else unset REMORA_HIP_LIB; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-reads --no-others --no-refine --details /tmp/d.json > /dev/null 2>/tmp/e.err
  python - <<PY
import json
r=json.load(open('/tmp/d.json'))['encode_roofline']
print("lib=${lib:-shipped}", round(r['achieved']), "GB/s", round(r['frac'],3), round(r['avg_launch_ms'],4), "ms")
PY
done
