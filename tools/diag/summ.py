import json,sys
r=json.load(open(sys.argv[1]))
print("images/s", round(r["value"],1), "whole-path frac", round(r["frac_fp32_mfma_peak_whole_path"],3))
if "roofline" in r: print("roofline frac", round(r["roofline"]["frac"],3))
for k,v in r.get("kernels",{}).items(): print(f'{k:24s} n={v["launches"]:4d} avg_us={v["avg_us"]:8.1f} share={v["share"]:.3f} TF={v["tflops"]:6.1f} GB/s={v["alg_gbps"]:7.1f}')
if "cpu_baseline" in r: print(r["cpu_baseline"], r.get("check_max_abs_vs_cpu_oracle"))
