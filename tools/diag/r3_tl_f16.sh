# per-workgroup phase timelines of the two-fp16-term kernels (B=1024): 3x3 at 32x32 and 16x16, stride-2 32->16, transposed 16->32
export DM_SAMPLER_LIB=tools/diag/bin/libdm_timeline.so
for a in "15 32" "15 16" "13 32" "14 16"; do
  echo "== kind H = $a"
  timeout -k 10 100 python tools/diag/timeline.py $a 1024 2>/dev/null || exit 1
done
