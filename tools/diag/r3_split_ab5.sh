mkdir -p gpurun_out/r3s
run() { DM_SPLIT=$1 timeout -k 10 250 python bench.py --no-cpu-baseline --no-configs --no-roofline "${@:2}" 2> gpurun_out/r3s/c_err.log | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))" || { tail -3 gpurun_out/r3s/c_err.log; exit 1; }; }
for rep in 1 2; do
  echo "B=1024: split4 $(run 4 --steps 10 --warmup 2) split8 $(run 8 --steps 10 --warmup 2) split2 $(run 2 --steps 10 --warmup 2) split1 $(run 1 --steps 10 --warmup 2)"
done
echo "B=512: split4 $(run 4 --batch 512 --steps 20 --warmup 2) split8 $(run 8 --batch 512 --steps 20 --warmup 2)"
echo "B=2048: split4 $(run 4 --batch 2048 --steps 5 --warmup 2) split8 $(run 8 --batch 2048 --steps 5 --warmup 2)"
echo "B=8192: split4 $(run 4 --batch 8192 --steps 3 --warmup 1) split8 $(run 8 --batch 8192 --steps 3 --warmup 1)"
echo "bf16 cfg B=1024: split4 $(run 4 --precision bf16 --cfg --steps 5 --warmup 2) split8 $(run 8 --precision bf16 --cfg --steps 5 --warmup 2)"
