#!/bin/bash
# bench-only A/B of DM_X6 masks on one box (no tests)
for m in "$@"; do
  echo "=== DM_X6=$m"
  DM_X6=$m timeout -k 10 200 python bench.py --steps 5 --warmup 2 2> gpurun_out/x6_bench_$m.err > gpurun_out/x6_bench_$m.json || { tail -5 gpurun_out/x6_bench_$m.err; continue; }
  python tools/diag/bsum.py gpurun_out/x6_bench_$m.json | head -9
done
