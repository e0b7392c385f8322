# dm_sample's parallel sub-batch branches (DM_SPLIT = 1 | 2 | 4): rates of the eta>0 B=64 chain and of DDIM-50 at B=1024
mkdir -p gpurun_out/r3s
for sp in 1 2 4; do
  DM_SPLIT=$sp timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-roofline --sampler ancestral --ddim-steps 1000 --batch 64 > gpurun_out/r3s/cfg1_$sp.json 2> gpurun_out/r3s/err1_$sp.log || { tail -5 gpurun_out/r3s/err1_$sp.log; exit 1; }
  DM_SPLIT=$sp timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-roofline > gpurun_out/r3s/cfg2_$sp.json 2> gpurun_out/r3s/err2_$sp.log || { tail -5 gpurun_out/r3s/err2_$sp.log; exit 1; }
  python - <<PY
import json
a=json.load(open("gpurun_out/r3s/cfg1_$sp.json")); b=json.load(open("gpurun_out/r3s/cfg2_$sp.json"))
print("split $sp: cfg1", round(a["value"],1), "check", a.get("check_max_abs_vs_cpu_oracle"), "| cfg2", round(b["value"],1), "check", b.get("check_max_abs_vs_cpu_oracle"))
PY
done
