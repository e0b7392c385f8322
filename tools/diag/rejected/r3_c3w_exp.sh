# (ran on the two-wave 3x3 variant before it was retired to tools/diag/rejected/dm_kernels_x6c3w.hip.txt; its -DDM_C3W_EXP=N libraries are no longer built)
# what holds conv_x6_c3w_kernel's K loop at 54 cycles per MFMA: timelines with the staging (e4), the pixel-fragment re-reads (e8),
# the weight-fragment loads (e16) or all three (e28) left out (-DDM_C3W_EXP=N builds; wrong results, only the times are read)
for lib in tools/diag/bin/libdm_timeline.so tools/diag/bin/libdm_tl_c3we4.so tools/diag/bin/libdm_tl_c3we8.so tools/diag/bin/libdm_tl_c3we16.so tools/diag/bin/libdm_tl_c3we28.so; do
  echo "== $lib"
  DM_SAMPLER_LIB=$lib timeout -k 10 100 python tools/diag/timeline.py 15 32 1024 2>/dev/null || exit 1
done
