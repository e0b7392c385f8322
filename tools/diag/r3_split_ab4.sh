mkdir -p gpurun_out/r3s
run() { DM_SPLIT=$1 timeout -k 10 250 python bench.py --no-cpu-baseline --no-configs --no-roofline "${@:2}" 2> gpurun_out/r3s/c_err.log | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))" || { tail -3 gpurun_out/r3s/c_err.log; exit 1; }; }
for rep in 1 2 3; do
  echo "B=1024: split2 $(run 2 --steps 10 --warmup 2) split4 $(run 4 --steps 10 --warmup 2) split1 $(run 1 --steps 10 --warmup 2)"
done
echo "B=8192: split1 $(run 1 --batch 8192 --steps 3 --warmup 1) split4 $(run 4 --batch 8192 --steps 3 --warmup 1) split2 $(run 2 --batch 8192 --steps 3 --warmup 1)"
echo "wide B=256: split1 $(run 1 --model wide --batch 256 --steps 2 --warmup 1) split4 $(run 4 --model wide --batch 256 --steps 2 --warmup 1)"
echo "wide-attn B=256: split1 $(run 1 --model wide-attn --batch 256 --steps 2 --warmup 1) split4 $(run 4 --model wide-attn --batch 256 --steps 2 --warmup 1) split2 $(run 2 --model wide-attn --batch 256 --steps 2 --warmup 1)"
echo "bf16 B=1024 (no guidance): split1 $(run 1 --precision bf16 --steps 5 --warmup 2) split4 $(run 4 --precision bf16 --steps 5 --warmup 2)"
echo "ancestral-1000 B=256: split1 $(run 1 --sampler ancestral --ddim-steps 1000 --batch 256 --steps 1 --warmup 1) split4 $(run 4 --sampler ancestral --ddim-steps 1000 --batch 256 --steps 1 --warmup 1)"
