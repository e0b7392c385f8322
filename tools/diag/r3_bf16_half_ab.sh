# bf16 path, MFMA operands of the layers behind a GroupNorm as IEEE half (DM_BF16_F16 = 0 | 1 | 2): parity margin + cfg4 rate
mkdir -p gpurun_out/r3h
for md in 0 1 2; do
  DM_BF16_F16=$md timeout -k 10 300 python -u -m pytest tests/test_gpu_round2.py -m gpu -x -q -s -k "bf16_parity_margin" --timeout 180 --timeout-method thread > gpurun_out/r3h/margin_$md.log 2>&1 || { echo "margin mode $md failed"; tail -5 gpurun_out/r3h/margin_$md.log; }
  grep "bf16 seed" gpurun_out/r3h/margin_$md.log
  DM_BF16_F16=$md timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --precision bf16 --cfg > gpurun_out/r3h/cfg4_$md.json 2> gpurun_out/r3h/err_$md.log || exit 1
  python - <<PY
import json
j=json.load(open("gpurun_out/r3h/cfg4_$md.json"))
print("mode $md cfg4", round(j["value"],1), {k:(round(v["avg_us"],1), v["launches"]) for k,v in j["kernels"].items()})
PY
done
