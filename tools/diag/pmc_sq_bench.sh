#!/bin/bash
# MFMA-busy fraction per kernel over a short bench run (one SQ counter pass): bash tools/diag/pmc_sq_bench.sh <tag> <bench args...>
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; tag=$1; shift; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sqb_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-configs "$@" > $O/pmc_sqb_$tag.json 2> $O/pmc_sqb_$tag.err || exit 1
python3 - <<PY
import csv, glob, collections, json
f = glob.glob("$O/pmc_sqb_$tag/*/*counter_collection.csv")[0]
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"]); per[r["Dispatch_Id"]]["_k"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0.0, 0])
for d in per.values():
    if "GRBM_GUI_ACTIVE" not in d: continue
    a = agg[d["_k"]]
    a[0] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0); a[1] += d["GRBM_GUI_ACTIVE"] / 8.0 * 1024; a[2] += d.get("SQ_INSTS_VALU", 0); a[3] += d.get("SQ_INSTS_MFMA", 0); a[4] += 1
out = {}
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if a[3] == 0: continue
    out[k] = {"launches": a[4], "mfma_busy_frac_of_simd_cycles": a[0] / a[1], "share_of_simd_cycles": a[1] / sum(v[1] for v in agg.values()),
              "valu_non_mfma_per_mfma": (a[2] - a[3]) / a[3]}
    print(f"{k:44s} n={a[4]:5d}  MFMA busy {a[0]/a[1]*100:5.1f} % of SIMD cycles   {out[k]['valu_non_mfma_per_mfma']:.1f} VALU per MFMA")
json.dump(out, open("$O/pmc_sqb_$tag" + "_summary.json", "w"), indent=1)
PY
