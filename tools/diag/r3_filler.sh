# timing experiment: N extra vector instructions of the wave's own behind every bf16 MFMA of conv_x6_kernel's K loop
# (libdm_fillN.so: -DDM_X6_FILLER=N).  Does the matrix pipe's shadow take them for free?
mkdir -p gpurun_out/r3fill
for lib in diffusion_model_amd/libdm_sampler.so tools/diag/bin/libdm_fill3.so tools/diag/bin/libdm_fill6.so; do
    DM_X6=7 DM_DEBUG_X6_NOPIPE=1 DM_SAMPLER_LIB=$lib timeout -k 10 100 python - <<PY || exit 1
import torch, sys
sys.path.insert(0, '.')
import bench
import diffusion_model_amd as d
d.set_seed(1)
m = d.UNetMini(image_size=32).eval()
per, raw = bench.profile_model(m, d.DDPMScheduler(), 1024, 32, 32, reps=10)
print("$lib", {k: round(v["avg_us"], 1) for k, v in per.items() if "split16" in k})
PY
done
