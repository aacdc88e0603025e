import json, sys
d = json.load(open(sys.argv[1]))
print(f"value {d['value']/1e6:.2f} M chunks/s  ms/step {d['ms_per_step']:.2f}  roofline {d['roofline']['kernel']} {d['roofline']['frac']:.3f}  pipeline frac {d['whole_pipeline']['frac_of_fp32_mfma_peak']:.3f}")
for k, v in d["kernels"].items():
    print(f"  {k:14s} {v['ms_total']:8.1f} ms  {v['launches']:5d} launches  {v['avg_ms']*1e3:8.1f} us  {v['tflops'] if v['tflops'] is None else round(v['tflops'],1)} TF")
if "cpu_baseline" in d:
    print("  cpu:", d["cpu_baseline"])
