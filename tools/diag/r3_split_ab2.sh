# repeated A/B of DM_SPLIT on DDIM-50 (B = 1024 and B = 8192), alternating, same box
mkdir -p gpurun_out/r3s
for rep in 1 2 3; do
for sp in 1 2 3; do
  DM_SPLIT=$sp timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-roofline > gpurun_out/r3s/a_$sp.json 2> gpurun_out/r3s/a_err.log || { tail -5 gpurun_out/r3s/a_err.log; exit 1; }
  DM_SPLIT=$sp timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-roofline --batch 8192 > gpurun_out/r3s/b_$sp.json 2> gpurun_out/r3s/b_err.log || { tail -5 gpurun_out/r3s/b_err.log; exit 1; }
  python - <<PY
import json
a=json.load(open("gpurun_out/r3s/a_$sp.json")); b=json.load(open("gpurun_out/r3s/b_$sp.json"))
print("rep $rep split $sp: B=1024", round(a["value"],1), "| B=8192", round(b["value"],1))
PY
done
done
