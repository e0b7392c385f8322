# the fp16 two-term form of the stride-2 / transposed convolutions (default, DM_X6=39) against the three-way bf16 split (DM_X6=7)
mkdir -p gpurun_out/r3f16
true
true
for x in 487 231; do
  for B in 1024 64; do
    DM_X6=$x timeout -k 10 100 python - <<PY || exit 1
import torch, sys
sys.path.insert(0, '.')
import bench
import diffusion_model_amd as d
d.set_seed(1)
m = d.UNetMini(image_size=32).eval()
per, raw = bench.profile_model(m, d.DDPMScheduler(), $B, 32, 32, reps=10)
print("DM_X6=$x B=$B", {k: round(v["avg_us"], 1) for k, v in per.items() if "split16" in k or "wino" in k}, "total", round(sum(v["avg_us"] * v["launches"] for v in per.values()), 1))
PY
  done
done
