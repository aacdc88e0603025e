for v in "1 1" "0 0" "1 0" "1 1" "0 0"; do set -- $v; RMR_ENCODE_PREFETCH=$1 RMR_ENCODE_UNROLL=$2 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-reads --no-others --no-refine --details /tmp/d.json > /dev/null 2>/tmp/e.err; python - <<PY
import json
d=json.load(open('/tmp/d.json'))
r=d.get('encode_roofline') or d.get('details',{}).get('encode_roofline')
print("prefetch=$1 unroll=$2", r)
PY
done
