#!/bin/bash
# issue-activity breakdown of one conv layer (two SQ counter passes): bash tools/diag/sq_breakdown.sh <kind> <H>
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sqb_a_$1_$2 -- python3 $R/tools/diag/pmc_layer.py $1 $2 1024 > $O/sqb_a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sqb_b_$1_$2 -- python3 $R/tools/diag/pmc_layer.py $1 $2 1024 > $O/sqb_b.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for tag in ("a", "b"):
    f = glob.glob("$O/sqb_%s_$1_$2/*/*counter_collection.csv" % tag)[0]
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "conv_" in r["Kernel_Name"]: per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    i = sorted(per, key=int)[-1]; c = per[i]
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    print("kind $1 H $2 pass", tag, "kernel cycles %.0f" % cyc)
    for k, v in sorted(c.items()):
        if k != "GRBM_GUI_ACTIVE": print("   %-28s %14.0f  per SIMD-cycle %.3f" % (k, v, v / (cyc * 1024)))
PY
