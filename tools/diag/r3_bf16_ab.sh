mkdir -p gpurun_out/r3i
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3i/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3i/pytest.log
for lr in 1 0; do
  DM_LOWRES=$lr timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --precision bf16 --cfg > gpurun_out/r3i/cfg4_lr$lr.json 2> gpurun_out/r3i/err.log
  python - <<PY
import json
j=json.load(open("gpurun_out/r3i/cfg4_lr$lr.json"))
print("lowres=$lr cfg4", round(j["value"],1), {k:(round(v["avg_us"],1), v["launches"]) for k,v in j["kernels"].items()})
PY
done
