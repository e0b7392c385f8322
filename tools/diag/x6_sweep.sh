#!/bin/bash
# GPU tests + bench for several DM_X6 masks
for m in "$@"; do
  echo "=== DM_X6=$m"
  DM_X6=$m timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
  DM_X6=$m timeout -k 10 200 python bench.py --steps 5 --warmup 2 2>/dev/null > gpurun_out/x6_bench_$m.json || exit 1
  python tools/diag/bsum.py gpurun_out/x6_bench_$m.json || exit 1
done
