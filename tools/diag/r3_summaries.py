# Summaries of tools/diag/r3_profiles.sh's passes (gpurun_out/r3prof/*) -> profiles/r3_* (run in the build container):
#   python3 tools/diag/r3_summaries.py
# Every JSON written here carries `build_key` = the key of the library the passes ran on (taken from the bench line each pass
# printed); bench.py reports a figure from them only while the loaded library has the same key.
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM).
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "gpurun_out", "r3prof")
DST = os.path.join(ROOT, "profiles")


def one(pat):
    fs = glob.glob(os.path.join(SRC, pat))
    if not fs:
        sys.exit("missing " + pat)
    return max(fs, key=os.path.getmtime)


def bench_line(name):
    txt = open(os.path.join(SRC, name)).read().strip().splitlines()
    return json.loads(txt[-1])


def short(k):
    return k.split("(")[0].replace("void ", "").strip()


def cls_of(k):
    """bench.py's kernel-class name (K_NAMES) of a kernel."""
    import re
    if k.startswith("conv_mfma_kernel<2, 5"): return "conv3x3_winograd_mfma"
    if re.match(r"conv_x6_kernel<\d, 0", k) or k.startswith("conv_x6_c3"): return "conv3x3_split16_mfma"
    if re.match(r"conv_x6_kernel<\d, 1", k) or k.startswith("conv_x6_down_kernel"): return "conv4x4s2_split16_mfma"
    if re.match(r"conv_x6_kernel<\d, [24]", k): return "convT4x4s2_split16_mfma"
    if k.startswith("lowres_fused_kernel"): return "lowres_fused_split16_mfma"
    if k.startswith("init5_kernel"): return "init_conv"
    if k.startswith("final_conv"): return "final_conv"
    return None


def by_class(per, wkey, vkeys):
    """launch-weighted class figures from per-kernel ones"""
    out = {}
    for k, e in per.items():
        c = cls_of(k)
        if c is None:
            continue
        o = out.setdefault(c, {"launches": 0, "kernels": []})
        for v in vkeys:
            o[v] = o.get(v, 0.0) + e[v] * e[wkey]
        o["launches"] += e[wkey]
        o["kernels"].append(k)
    for o in out.values():
        for v in vkeys:
            o[v] /= o["launches"]
    return out


keys = set()

# ---- kernel-trace --stats passes: the CSV as rocprofv3 wrote it + the bench line of the same command
for cfg in ("cfg1", "cfg2", "cfg4", "cfg5"):
    shutil.copy(one(f"prof_{cfg}/*/*kernel_stats.csv"), os.path.join(DST, f"r3_{cfg}_kernel_stats.csv"))
    j = bench_line(f"prof_{cfg}.json")
    keys.add(j["build_key"])
    json.dump(j, open(os.path.join(DST, f"r3_{cfg}_bench_profiled.json"), "w"), indent=1)

# ---- the DEFAULT scheduling of the cfg2 command (four sub-batch chains on parallel graph branches; r3_profiles.sh stats4), for the record:
#      launches cover a quarter of the batch and overlap each other, so their durations are not a kernel's own time
if glob.glob(os.path.join(SRC, "prof_cfg2_split4/*/*kernel_stats.csv")):
    shutil.copy(one("prof_cfg2_split4/*/*kernel_stats.csv"), os.path.join(DST, "r3_cfg2_kernel_stats_split4.csv"))
    j = bench_line("prof_cfg2_split4.json")
    keys.add(j["build_key"])
    json.dump(j, open(os.path.join(DST, "r3_cfg2_bench_profiled_split4.json"), "w"), indent=1)

# ---- class average of the dominant kernel by rocprof (cfg2 command), steady launches only
rows = list(csv.DictReader(open(one("prof_cfg2/*/*kernel_trace.csv"))))
per = collections.defaultdict(list)
for r in rows:
    per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
summary = {}
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) < 50.0:
        continue
    summary[k] = {"launches": len(v), "avg_us": sum(v) / len(v), "total_ms": sum(v) / 1e3, "min_us": min(v), "max_us": max(v)}
classes = by_class(summary, "launches", ["avg_us"])
cfg2 = bench_line("prof_cfg2.json")
dom = cfg2["roofline"]["class"]
out = {"source": "DM_SPLIT=1 (one sampling chain: a launch covers the whole batch, as in bench.py's per-class pass) rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs "
                 "(tools/diag/r3_profiles.sh stats; DDIM-50, B=1024, fp32): per-kernel durations from the kernel trace, every launch of "
                 "the run (warm-up, timed passes and bench.py's per-class profiling pass)",
       "build_key": cfg2["build_key"], "dominant_class": dom, "bench_line_avg_launch_us_same_run": cfg2["roofline"]["avg_launch_us"],
       "classes": classes, "per_kernel": summary}
json.dump(out, open(os.path.join(DST, "r3_rocprof_summary.json"), "w"), indent=1)
print("dominant class %s: %d launches, %.1f us by rocprof (bench line of the same run: %.1f us)" % (dom, classes[dom]["launches"], classes[dom]["avg_us"], cfg2["roofline"]["avg_launch_us"]))

# ---- HBM traffic (two PMC passes)
def load(path, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(one(path))):
        if r["Counter_Name"] == name:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


f = load("pmc_fetch/*/*counter_collection.csv", "FETCH_SIZE")
w = load("pmc_write/*/*counter_collection.csv", "WRITE_SIZE")
keys.add(bench_line("pmc_fetch.json")["build_key"])
keys.add(bench_line("pmc_write.json")["build_key"])
tper = {}
for k in sorted(f):
    if not k.startswith(("conv_", "final_", "sampler_", "attention", "init5", "lowres_", "gconv", "gnorm", "gstats")):
        continue
    rd = 2 * 1024 * sum(f[k]) / len(f[k])
    wr = 1024 * sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
    tper[k] = {"launches": len(f[k]), "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    print(f"{k:44s} {len(f[k]):5d}  read {rd/1e6:8.1f} MB  write {wr/1e6:8.1f} MB")
tcls = by_class(tper, "launches", ["hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "hbm_bytes_per_launch"])
json.dump({"source": "DM_SPLIT=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs: tools/diag/r3_profiles.sh traffic) -- python3 "
                     "bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-configs (DDIM-50, B=1024, fp32); FETCH_SIZE doubled "
                     "(gfx950 tallies 128-B requests as 64 B), KiB -> bytes; `classes`: launch-weighted per kernel class of bench.py",
           "build_key": bench_line("pmc_fetch.json")["build_key"], "classes": tcls, "per_kernel": tper},
          open(os.path.join(DST, "r3_traffic.json"), "w"), indent=1)
for c, o in tcls.items():
    print(f"class {c:28s} {o['launches']:5d} launches  {o['hbm_read_bytes_per_launch']/1e6:8.1f} MB read {o['hbm_write_bytes_per_launch']/1e6:8.1f} MB written per launch")

# ---- SQ counters per kernel (MFMA-busy share of SIMD cycles, vector instructions per MFMA)
for tag in ("model", "cfg1", "cfg4"):
    d = f"pmc_sq_{tag}"
    if tag == "cfg4" and not glob.glob(os.path.join(SRC, d + ".json")):
        continue  # (optional pass: r3_profiles.sh sq4)
    keys.add(bench_line(d + ".json")["build_key"])
    crow = list(csv.DictReader(open(one(d + "/*/*counter_collection.csv"))))
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(one(d + "/*/*kernel_trace.csv")))}
    disp = collections.defaultdict(dict)
    for r in crow:
        disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        disp[r["Dispatch_Id"]]["_k"] = short(r["Kernel_Name"])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for i, c in disp.items():
        a = agg[c["_k"]]
        a["launches"] += 1
        t = trace.get(i)
        if t:
            a["us"] += (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
        for k, v in c.items():
            if not k.startswith("_"):
                a[k] += v
    res = {}
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
        if a.get("SQ_INSTS_MFMA", 0) == 0 and a["us"] < 100:
            continue
        simd_cycles = a["GRBM_GUI_ACTIVE"] / 8.0 * 1024  # GRBM summed over the 8 XCDs -> kernel cycles x 256 CUs x 4 SIMDs
        res[k] = {"launches": int(a["launches"]), "avg_us_under_pmc": a["us"] / a["launches"],
                  "mfma_busy_frac_of_simd_cycles": a["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles if simd_cycles else None,
                  "mfma_coexec_cycles": a["SQ_VALU_MFMA_COEXEC_CYCLES"], "insts_valu": a["SQ_INSTS_VALU"], "insts_mfma": a["SQ_INSTS_MFMA"],
                  "valu_non_mfma_per_mfma": (a["SQ_INSTS_VALU"] - a["SQ_INSTS_MFMA"]) / a["SQ_INSTS_MFMA"] if a["SQ_INSTS_MFMA"] else None,
                  "kernel_clock_ghz_grbm": a["GRBM_GUI_ACTIVE"] / 8.0 / (a["us"] * 1e3) if a["us"] else None}
        r = res[k]
        print(f"{tag:6s} {k:44s} n={r['launches']:5d} {r['avg_us_under_pmc']:8.1f} us  busy {100*(r['mfma_busy_frac_of_simd_cycles'] or 0):5.1f}%  "
              f"valu/mfma {r['valu_non_mfma_per_mfma'] or 0:5.2f}  clk {r['kernel_clock_ghz_grbm'] or 0:4.2f} GHz")
    cmd = ("DDIM-50, B=1024, fp32" if tag == "model" else "DDIM-50, B=1024, bf16 + guidance (2,048-image denoiser batch)" if tag == "cfg4"
           else "ancestral, first 100 of 1000 steps, B=64, fp32")
    json.dump({"source": "DM_SPLIT=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES "
                         "SQ_WAVE_CYCLES GRBM_GUI_ACTIVE (tools/diag/r3_profiles.sh sq; one bench pass, " + cmd + "); all launches of a kernel summed; "
                         "busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)",
               "build_key": bench_line(d + ".json")["build_key"], "per_kernel": res},
              open(os.path.join(DST, f"r3_pmc_sq_{tag}.json"), "w"), indent=1)

if len(keys) != 1:
    sys.exit("the passes ran on different builds: %s" % sorted(keys))
print("build_key", keys.pop())
