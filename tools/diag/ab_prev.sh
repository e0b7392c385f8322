for lib in tools/diag/bin/libdm_prev.so diffusion_model_amd/libdm_sampler.so tools/diag/bin/libdm_prev.so diffusion_model_amd/libdm_sampler.so; do
    DM_SAMPLER_LIB=$lib timeout -k 10 100 python - <<PY || exit 1
import torch, sys
sys.path.insert(0, '.')
import bench
import diffusion_model_amd as d
d.set_seed(1)
m = d.UNetMini(image_size=32).eval()
per, raw = bench.profile_model(m, d.DDPMScheduler(), 1024, 32, 32, reps=10)
print("$lib", {k: (v["launches"], round(v["avg_us"], 1)) for k, v in per.items() if "split16" in k}, "total", round(sum(v["avg_us"] * v["launches"] for v in per.values()), 1))
PY
done
