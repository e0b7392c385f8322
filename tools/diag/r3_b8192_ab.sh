mkdir -p gpurun_out/r3h
for lr in 1 0 1 0; do
  DM_LOWRES=$lr timeout -k 10 200 python bench.py --steps 2 --warmup 1 --batch 8192 --no-cpu-baseline --no-configs > gpurun_out/r3h/b8192_lr$lr.json 2> gpurun_out/r3h/err.log
  python - <<PY
import json
j=json.load(open("gpurun_out/r3h/b8192_lr$lr.json"))
print("lowres=$lr B=8192", round(j["value"],1), {k:(round(v["avg_us"],1), v["launches"]) for k,v in j["kernels"].items() if "bf16x6" in k or "wino" in k})
PY
done
