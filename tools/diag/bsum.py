# one-screen summary of a bench.py JSON line (stdin or file argument)
import json, sys
d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read())
print("images/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "roofline.frac", round(d["roofline"]["frac"], 3))
for k, v in d["kernels"].items(): print("   ", k, v["launches"], round(v["avg_us"], 1), "us")
c = d.get("configs", {})
for k, v in c.items():
    if isinstance(v, dict) and "images_per_s" in v: print("   ", k, round(v["images_per_s"], 1), "img/s check", v.get("check_max_abs_vs_cpu_oracle"))
