import csv, collections, sys, json, glob
def load(path, name):
    rows = list(csv.DictReader(open(glob.glob(path)[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        if r['Counter_Name'] == name:
            agg[(r['Kernel_Name'].split('(')[0][-40:], r['Grid_Size'])].append(float(r['Counter_Value']))
    return agg
f = load('gpurun_out/pmc_fetch/*/*counter_collection.csv', 'FETCH_SIZE')
w = load('gpurun_out/pmc_write/*/*counter_collection.csv', 'WRITE_SIZE')
out = {}
for k in sorted(f):
    fk = sum(f[k]) / len(f[k]); wk = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
    # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM)
    out[f"{k[0]}|grid={k[1]}"] = {"launches": len(f[k]), "fetch_KiB_raw": fk, "write_KiB": wk,
                                  "hbm_read_MB_corrected": 2 * fk * 1024 / 1e6, "hbm_write_MB": wk * 1024 / 1e6}
    print(k, len(f[k]), f"read {2*fk*1024/1e6:8.1f} MB  write {wk*1024/1e6:8.1f} MB")
json.dump(out, open('gpurun_out/traffic_summary.json', 'w'), indent=1)
