export DM_SAMPLER_LIB=tools/diag/bin/libdm_timeline.so
for z in 0 1; do for k in 8 9 5; do
  if [ $z = 1 ]; then export DM_TL_ZERO=1; else unset DM_TL_ZERO; fi
  echo "== zero=$z kind=$k"
  timeout -k 10 100 python tools/diag/timeline.py $k $([ $k = 9 ] && echo 16 || echo 32) 1024 2>/dev/null | head -3 || exit 1
done; done
