# A/B of the stride-2 split-bf16 conv: staging pipelined into the K loop (default) vs conv_x6_kernel (DM_DEBUG_X6_NOPIPE=1)
mkdir -p gpurun_out/r3x6p
timeout -k 10 300 python -u -m pytest tests/test_gpu_x6.py tests/test_gpu_lowres.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r3x6p/test.log 2>&1 || { tail -30 gpurun_out/r3x6p/test.log; exit 1; }
tail -2 gpurun_out/r3x6p/test.log
for v in 0 1; do
  for B in 1024 64; do
    if [ $v = 1 ]; then export DM_DEBUG_X6_NOPIPE=1; else unset DM_DEBUG_X6_NOPIPE; fi
    timeout -k 10 100 python - <<PY || exit 1
import torch, sys
sys.path.insert(0, '.')
import bench
import diffusion_model_amd as d
d.set_seed(1)
m = d.UNetMini(image_size=32).eval()
per, raw = bench.profile_model(m, d.DDPMScheduler(), $B, 32, 32, reps=10)
print("nopipe=$v B=$B", {k: round(v["avg_us"], 1) for k, v in per.items() if "bf16x6" in k})
PY
  done
done
