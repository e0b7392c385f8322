# conv_x6_c3w_kernel (two waves of 64 pixels x 64 channels per tile, default) vs conv_x6_c3_kernel (DM_DEBUG_X6_NOC3W=1)
for v in 0 1 0 1; do
    if [ $v = 1 ]; then export DM_DEBUG_X6_NOC3W=1; else unset DM_DEBUG_X6_NOC3W; fi
    timeout -k 10 100 python - <<PY || exit 1
import torch, sys
sys.path.insert(0, '.')
import bench
import diffusion_model_amd as d
d.set_seed(1)
m = d.UNetMini(image_size=32).eval()
per, raw = bench.profile_model(m, d.DDPMScheduler(), 1024, 32, 32, reps=10)
print("noc3w=$v", {k: (v["launches"], round(v["avg_us"], 1)) for k, v in per.items() if "split16" in k or "wino" in k}, "total", round(sum(v["avg_us"] * v["launches"] for v in per.values()), 1))
PY
done
