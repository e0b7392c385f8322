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


keys = set()

# ---- kernel-trace --stats passes: the CSV as rocprofv3 wrote it + the bench line of the same command
for cfg in ("cfg1", "cfg2", "cfg4", "cfg5"):
    shutil.copy(one(f"prof_{cfg}/*/*kernel_stats.csv"), os.path.join(DST, f"r3_{cfg}_kernel_stats.csv"))
    j = bench_line(f"prof_{cfg}.json")
    keys.add(j["build_key"])
    json.dump(j, open(os.path.join(DST, f"r3_{cfg}_bench_profiled.json"), "w"), indent=1)

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
wino = [k for k in per if k.startswith("conv_mfma_kernel<2, 5, 64")]
wl = [x for k in wino for x in per[k]]
cfg2 = bench_line("prof_cfg2.json")
out = {"source": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs "
                 "(tools/diag/r3_profiles.sh stats; DDIM-50, B=1024, fp32): per-kernel durations from the kernel trace, every launch of "
                 "the run (warm-up, timed passes and bench.py's per-class profiling pass)",
       "build_key": cfg2["build_key"], "winograd_class": wino, "winograd_class_launches": len(wl),
       "winograd_class_avg_us": sum(wl) / len(wl), "bench_line_avg_launch_us_same_run": cfg2["roofline"]["avg_launch_us"],
       "per_kernel": summary}
json.dump(out, open(os.path.join(DST, "r3_rocprof_summary.json"), "w"), indent=1)
print("winograd class: %d launches, %.1f us (bench line of the same run: %.1f us)" % (len(wl), out["winograd_class_avg_us"], cfg2["roofline"]["avg_launch_us"]))

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
twino = [k for k in tper if k.startswith("conv_mfma_kernel<2, 5, 64")]
n = sum(tper[k]["launches"] for k in twino)
rd = sum(tper[k]["hbm_read_bytes_per_launch"] * tper[k]["launches"] for k in twino) / n
wr = sum(tper[k]["hbm_write_bytes_per_launch"] * tper[k]["launches"] for k in twino) / n
json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs: tools/diag/r3_profiles.sh traffic) -- python3 "
                     "bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-configs (DDIM-50, B=1024, fp32); FETCH_SIZE doubled "
                     "(gfx950 tallies 128-B requests as 64 B), KiB -> bytes; launch-weighted over the Winograd class (the 32x32 and 16x16 layers)",
           "build_key": bench_line("pmc_fetch.json")["build_key"], "kernel": "conv_mfma_kernel<2,5,64,*>", "launches": n,
           "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr, "per_kernel": tper},
          open(os.path.join(DST, "r3_traffic.json"), "w"), indent=1)
print("wino class:", n, "launches", rd / 1e6, "MB read", wr / 1e6, "MB written per launch")

# ---- SQ counters per kernel (MFMA-busy share of SIMD cycles, vector instructions per MFMA)
for tag in ("model", "cfg1"):
    d = f"pmc_sq_{tag}"
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
    cmd = ("DDIM-50, B=1024, fp32" if tag == "model" else "ancestral, first 100 of 1000 steps, B=64, fp32")
    json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES "
                         "SQ_WAVE_CYCLES GRBM_GUI_ACTIVE (tools/diag/r3_profiles.sh sq; one bench pass, " + cmd + "); all launches of a kernel summed; "
                         "busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)",
               "build_key": bench_line(d + ".json")["build_key"], "per_kernel": res},
              open(os.path.join(DST, f"r3_pmc_sq_{tag}.json"), "w"), indent=1)

if len(keys) != 1:
    sys.exit("the passes ran on different builds: %s" % sorted(keys))
print("build_key", keys.pop())
