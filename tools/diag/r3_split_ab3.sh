# DM_SPLIT 1 vs 2 (vs 4) over batch sizes, DDIM-50, 32x32 C=64; the widened model; bf16 without guidance
mkdir -p gpurun_out/r3s
run() { DM_SPLIT=$1 timeout -k 10 250 python bench.py --no-cpu-baseline --no-configs --no-roofline "${@:2}" 2> gpurun_out/r3s/c_err.log | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))" || { tail -3 gpurun_out/r3s/c_err.log; exit 1; }; }
for b in 128 256 512 2048 4096; do
  st=$((20480 / b)); [ $st -gt 20 ] && st=20; [ $st -lt 3 ] && st=3
  echo "B=$b: split1 $(run 1 --batch $b --steps $st --warmup 2) split2 $(run 2 --batch $b --steps $st --warmup 2) split4 $(run 4 --batch $b --steps $st --warmup 2) split1 $(run 1 --batch $b --steps $st --warmup 2)"
done
echo "wide B=256: split1 $(run 1 --model wide --batch 256 --steps 2 --warmup 1) split2 $(run 2 --model wide --batch 256 --steps 2 --warmup 1)"
echo "bf16 B=1024 (no guidance): split1 $(run 1 --precision bf16 --steps 5 --warmup 2) split2 $(run 2 --precision bf16 --steps 5 --warmup 2)"
echo "ancestral-1000 B=256: split1 $(run 1 --sampler ancestral --ddim-steps 1000 --batch 256 --steps 1 --warmup 1) split2 $(run 2 --sampler ancestral --ddim-steps 1000 --batch 256 --steps 1 --warmup 1)"
