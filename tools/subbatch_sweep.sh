# headline fp32 pipeline against the sub-batch size (do the intermediates of a sub-batch stay in the 256 MB Infinity Cache?)
for sb in 4096 8192 16384 65536 131072 262144 524288; do
  RMR_SUBBATCH=$sb python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-reads --no-others --no-refine --no-encode --details /tmp/d.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('RMR_SUBBATCH=$sb', round(d['value']/1e6,2), 'M chunks/s', round(d['roofline']['frac'],3))"
done
