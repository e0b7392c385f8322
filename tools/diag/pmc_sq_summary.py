# Summarise an SQ-counter pass (tools/diag/r2_profiles.sh sq): python3 tools/diag/pmc_sq_summary.py gpurun_out/pmc_sq_wino "conv_mfma"
import csv, glob, sys, collections, json
d, pat = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(glob.glob(d + '/*/*counter_collection.csv')[0])))
trace = {r['Dispatch_Id']: r for r in csv.DictReader(open(glob.glob(d + '/*/*kernel_trace.csv')[0]))}
per = collections.defaultdict(dict)
for r in rows:
    if pat in r['Kernel_Name']:
        per[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
        per[r['Dispatch_Id']]['_name'] = r['Kernel_Name'][:70]
ids = sorted(per, key=int)[2:]  # skip the first (cold) launches
out = []
for i in ids:
    c = per[i]
    t = trace.get(i)
    dur_ns = (int(t['End_Timestamp']) - int(t['Start_Timestamp'])) if t else float('nan')
    cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8.0            # sum over the 8 XCDs -> kernel cycles
    simd_cycles = cyc * 1024                            # 256 CUs x 4 SIMDs
    o = {'dispatch': i, 'kernel': c['_name'], 'duration_us_under_pmc': dur_ns / 1e3, 'kernel_cycles': cyc,
         'mfma_busy_frac_of_simd_cycles': c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd_cycles if simd_cycles else None,
         'mfma_coexec_cycles': c.get('SQ_VALU_MFMA_COEXEC_CYCLES'), 'insts_valu': c.get('SQ_INSTS_VALU'), 'insts_mfma': c.get('SQ_INSTS_MFMA'),
         'valu_non_mfma_per_mfma': (c.get('SQ_INSTS_VALU', 0) - c.get('SQ_INSTS_MFMA', 0)) / c['SQ_INSTS_MFMA'] if c.get('SQ_INSTS_MFMA') else None,
         'raw': {k: v for k, v in c.items() if not k.startswith('_')}}
    out.append(o)
    print(json.dumps({k: v for k, v in o.items() if k != 'raw'}))
json.dump(out, open(d + '_summary.json', 'w'), indent=1)
