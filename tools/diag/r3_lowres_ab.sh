# A/B of the fused low-resolution kernel against the layer-by-layer kernels (DM_LOWRES=0), cfg2 and cfg1, one box
set -e
mkdir -p gpurun_out/r3d
timeout -k 10 120 python -u -m pytest tests/test_gpu_lowres.py -m gpu -x -q --timeout 100 --timeout-method thread > gpurun_out/r3d/pytest.log 2>&1 || { tail -30 gpurun_out/r3d/pytest.log; exit 1; }
tail -2 gpurun_out/r3d/pytest.log
timeout -k 10 120 python tools/diag/stamps_lowres.py 64 1024 | tee gpurun_out/r3d/stamps.log
for lr in ${LRS:-1 0}; do
  DM_LOWRES=$lr timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs > gpurun_out/r3d/cfg2_lr$lr.json 2> gpurun_out/r3d/cfg2_lr$lr.err
  DM_LOWRES=$lr timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --sampler ancestral --ddim-steps 1000 --batch 64 > gpurun_out/r3d/cfg1_lr$lr.json 2> gpurun_out/r3d/cfg1_lr$lr.err
done
python - <<'PY'
import json, os
for f in ("cfg2_lr1","cfg2_lr0","cfg1_lr1","cfg1_lr0"):
    if not os.path.exists(f"gpurun_out/r3d/{f}.json"): continue
    j=json.load(open(f"gpurun_out/r3d/{f}.json"))
    print(f, round(j["value"],1), {k:(round(v["avg_us"],1), v["launches"]) for k,v in j["kernels"].items()})
PY
