# whole-sampler A/B: conv_x6_c3w_kernel (default) vs conv_x6_c3_kernel (DM_DEBUG_X6_NOC3W=1), DDIM-50 B=1024, alternating
for v in 0 1 0 1; do
    if [ $v = 1 ]; then export DM_DEBUG_X6_NOC3W=1; else unset DM_DEBUG_X6_NOC3W; fi
    timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-roofline > gpurun_out/c3w_bench_$v.json 2>gpurun_out/c3w_bench_$v.err || exit 1
    python - <<PY
import json
j = json.loads(open("gpurun_out/c3w_bench_$v.json").read().strip().splitlines()[-1])
print("noc3w=$v", j["value"], j["unit"], j["ms_per_step"])
PY
done
