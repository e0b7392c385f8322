# timing experiment: the fused low-resolution kernel with v_mfma_f32_16x16x32_bf16 in place of 32x32x16 (garbage results)
mkdir -p gpurun_out/r3f
for lib in diffusion_model_amd/libdm_sampler.so tools/diag/bin/libdm_diag16.so; do
  for B in 1024 64; do
    DM_SAMPLER_LIB=$lib timeout -k 10 100 python - <<PY
import torch, ctypes as C, sys
sys.path.insert(0, '.')
import bench
import diffusion_model_amd as d
d.set_seed(1)
m = d.UNetMini(image_size=32).eval()
per, raw = bench.profile_model(m, d.DDPMScheduler(), $B, 32, 32, reps=10)
print("$lib", $B, {k: round(v["avg_us"], 1) for k, v in per.items() if "lowres" in k or "bf16x6" in k})
PY
  done
done
