# timing experiment on conv_x6_c3_kernel's K loop (the four-wave kernel: DM_DEBUG_X6_NOC3W=1): which operand stream holds it back?  libdm_c3eN.so = -DDM_C3_EXP=N
# (bit 0: pixel operands read from LDS once per chunk instead of once per (tap, j); bit 1: weight fragments loaded once per launch).
# Results are wrong by construction; only the times are read.
for lib in diffusion_model_amd/libdm_sampler.so tools/diag/bin/libdm_c3e1.so tools/diag/bin/libdm_c3e2.so tools/diag/bin/libdm_c3e3.so; do
    DM_DEBUG_X6_NOC3W=1 DM_SAMPLER_LIB=$lib timeout -k 10 100 python - <<PY || exit 1
import torch, sys
sys.path.insert(0, '.')
import bench
import diffusion_model_amd as d
d.set_seed(1)
m = d.UNetMini(image_size=32).eval()
per, raw = bench.profile_model(m, d.DDPMScheduler(), 1024, 32, 32, reps=10)
print("$lib", {k: (v["launches"], round(v["avg_us"], 1)) for k, v in per.items() if "split16" in k})
PY
done
