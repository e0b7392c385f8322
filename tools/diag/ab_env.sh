#!/bin/bash
# bench-only A/B of one environment variable on one box: [BENCH_ARGS="..."] tools/diag/ab_env.sh NAME v1 v2 ...   ("-" = unset)
n=$1; shift
for v in "$@" "$@"; do
  echo "=== $n=$v"
  if [ "$v" = "-" ]; then env -u $n timeout -k 10 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-configs $BENCH_ARGS 2>/dev/null > gpurun_out/ab_env.json
  else env $n=$v timeout -k 10 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-configs $BENCH_ARGS 2>/dev/null > gpurun_out/ab_env.json; fi
  python tools/diag/bsum.py gpurun_out/ab_env.json | head -10
done
