# per-workgroup phase timelines of the stride-2 split-bf16 conv (kind 8, 32x32, B=1024): pipelined vs conv_x6_kernel, 1 and 2 workgroups per CU
export DM_SAMPLER_LIB=tools/diag/bin/libdm_timeline.so
for np in 0 1; do for sd in 1 2; do
  if [ $np = 1 ]; then export DM_DEBUG_X6_NOPIPE=1; else unset DM_DEBUG_X6_NOPIPE; fi
  echo "== nopipe=$np slot_div=$sd"
  DM_DEBUG_SLOT_DIV=$sd timeout -k 10 100 python tools/diag/timeline.py 8 32 1024 || exit 1
done; done
