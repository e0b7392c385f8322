for v in NOB NOA; do cp tools/diag/bin/libdm_stamps_$v.so tools/diag/bin/libdm_stamps.so; echo "== $v"; timeout -k 10 120 python tools/diag/stamps_lowres.py 64 | head -4; done
