( for i in $(seq 1 60); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/clk_during_bench.log &
SP=$!
sleep 3
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-encode --no-reads --no-others --no-refine --details gpurun_out/clk_bench.json > gpurun_out/clk_line.json 2>/dev/null
kill $SP 2>/dev/null
python -c "
import json;d=json.load(open('gpurun_out/clk_line.json'));print(d['value'], d['ms_per_step'])"
head -3 gpurun_out/clk_during_bench.log; echo ...; sed -n 20,40p gpurun_out/clk_during_bench.log
