#!/bin/bash
# Register / scratch / LDS budget of every kernel of the library as hipcc reports it (the library's own flags):
#   bash tools/diag/r3_resource_usage.sh > profiles/r3_resource_usage.txt
cd "$(dirname "$0")/../.." || exit 1
KEY=$(python3 -c "import __graft_entry__ as g; print(g.build_key())")
echo "# hipcc -Rpass-analysis=kernel-resource-usage, flags of __graft_entry__.FLAGS, build key $KEY"
echo "# columns: kernel | SGPRs | VGPRs | AGPRs | scratch bytes/lane | occupancy waves/SIMD | SGPR spills | VGPR spills | LDS bytes (static)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -shared -fPIC -DDM_BUILD_KEY="\"$KEY\"" \
  -Rpass-analysis=kernel-resource-usage diffusion_model_amd/csrc/dm_api.hip -o /tmp/r3_ru.so 2>&1 | python3 -c '
import re, sys
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    for k, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                   ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("ss", r"SGPRs Spill: (\d+)"), ("vs", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None and k not in cur: cur[k] = m.group(1)
import subprocess
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    print(" | ".join([name] + [r.get(k, "?") for k in ("sgpr", "vgpr", "agpr", "scratch", "occ", "ss", "vs", "lds")]))
'
rm -f /tmp/r3_ru.so
