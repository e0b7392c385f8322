# per-workgroup phase timelines of a persistent conv kernel (-DDM_STAMPS -DDM_TIMELINE build): how the phases of the
# two workgroups of a CU interleave.  usage: timeline.py kind H B [tf stats]
import os, sys, ctypes as C
os.environ.setdefault("DM_SAMPLER_LIB", "tools/variants/libdm_timeline.so")
sys.path.insert(0, '.')
import torch, numpy as np
from diffusion_model_amd import _lib
L = _lib.lib()
L.dm_debug_set_stamps.argtypes = [C.c_void_p]
dev = 'cuda'
kind, H, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tf = int(sys.argv[4]) if len(sys.argv) > 4 else 1
stats = int(sys.argv[5]) if len(sys.argv) > 5 else 1
x1 = torch.randn(B, H, H, 64, device=dev)
ks = 3 if kind in (0, 5, 7, 15) else 4
OH = H // 2 if kind in (1, 6, 8, 13) else 2 * H if kind in (2, 4, 9, 14) else H
w = torch.randn(64, 64, ks, ks, device=dev) * 0.05
if os.environ.get("DM_TL_ZERO"):  # all-zero operands: how much of the time is switching power (clock) rather than issue slots?
    x1.zero_(); w.zero_()
wp = torch.empty(L.dm_conv_packed_floats(kind, 64, 64), device=dev)
s = torch.cuda.current_stream().cuda_stream
_lib.check(L.dm_pack_conv_weight(kind, _lib.ptr(w), 64, 64, _lib.ptr(wp), s))
b = torch.randn(64, device=dev); g = torch.rand(64, device=dev) + .5; be = torch.randn(64, device=dev); te = torch.randn(64, device=dev)
pin = 4   # input partials (any producer): (mean 0, M2 = count) -> unit variance
si = torch.zeros(B, pin, 8, 2, device=dev); si[..., 1] = H * H * 8 / pin
y = torch.empty(B, OH, OH, 64, device=dev)
so = torch.empty(B, max(L.dm_conv_stats_partials(kind, H, H), 16), 8, 2, device=dev)
st = torch.zeros(1024 * 128, dtype=torch.int64, device=dev)
def call():
    _lib.check(L.dm_conv2d(kind, _lib.ptr(x1), 64, None, 0, _lib.ptr(wp), _lib.ptr(b), 64, _lib.ptr(si) if tf else None, pin,
                           H * H * 8 / pin, _lib.ptr(g), _lib.ptr(be), _lib.ptr(te), 0, _lib.ptr(y), _lib.ptr(so) if stats else None, B, H, H, s))
for _ in range(20): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(10): call()
e1.record(); torch.cuda.synchronize()
us_event = e0.elapsed_time(e1) * 100.0
L.dm_debug_set_stamps(st.data_ptr())
call(); torch.cuda.synchronize()
L.dm_debug_set_stamps(None)
a = st.cpu().numpy().astype(np.uint64).reshape(-1, 128)
wgs = []
for row in a:
    ev = row[2:]
    ev = ev[ev != 0]
    if len(ev) == 0: continue
    code = (ev >> np.uint64(56)).astype(int); tm = (ev & np.uint64((1 << 56) - 1)).astype(np.int64)
    hw = int(row[0]); cu = ((hw >> 32) & 0xff, (hw >> 8) & 0xff) if hw else (0, len(wgs) % 256)  # (xcc, se/sh/cu bits 8..15)
    wgs.append((cu, code, tm))
print(f"kind={kind} H={H} B={B} tf={tf} stats={stats} slot_div={os.environ.get('DM_DEBUG_SLOT_DIV','1')}: {len(wgs)} workgroups, {len(set(c for c,_,_ in wgs))} distinct CUs")
t0 = min(tm[0] for _, _, tm in wgs); t1 = max(tm[-1] for _, _, tm in wgs)
life = np.median([tm[-1] - tm[0] for _, _, tm in wgs])
print(f"launch {us_event:.1f} us by events; median workgroup life {life:.0f} cycles -> {life / us_event / 1e3:.2f} GHz if the launch is one workgroup life")
# per tile: events 1 (2 [5 2]*) 3 4 -> stage = time in 1->2 and 5->2, K = 2->5 / 2->3, epilogue = 3->4
def tiles_of(code, tm):
    out = []; cur = None
    for c, t in zip(code, tm):
        if c == 1: cur = {"t0": t, "stage": 0, "k": 0, "kiv": [], "last": t, "lastc": 1}
        elif cur is None: continue
        else:
            d = t - cur["last"]
            if cur["lastc"] in (1, 5): cur["stage"] += d
            elif cur["lastc"] == 2: cur["k"] += d; cur["kiv"].append((cur["last"], t))
            elif cur["lastc"] == 3: cur["epi"] = d
            cur["last"] = t; cur["lastc"] = c
            if c == 4:
                cur["t1"] = t; out.append(cur); cur = None
    return out
alltiles = []
bycu = {}
for cu, code, tm in wgs:
    tl = tiles_of(code, tm)
    alltiles += tl
    bycu.setdefault(cu, []).append(tl)
for nm in ("stage", "k", "epi"):
    v = np.array([t[nm] for t in alltiles]); print(f"  {nm:6s} n={len(v)} median {np.median(v):8.0f}  p10 {np.percentile(v,10):8.0f}  p90 {np.percentile(v,90):8.0f}")
v = np.array([t["t1"] - t["t0"] for t in alltiles]); print(f"  tile   n={len(v)} median {np.median(v):8.0f}  p10 {np.percentile(v,10):8.0f}  p90 {np.percentile(v,90):8.0f}")
# per CU: fraction of its span with 0 / 1 / 2 / 3+ workgroups inside a K loop
fr = []
for cu, lst in bycu.items():
    iv = []
    for tl in lst:
        for t in tl:
            for a0, a1 in t["kiv"]: iv.append((a0, 1)); iv.append((a1, -1))
    if not iv: continue
    iv.sort()
    lo = min(tl[0]["t0"] for tl in lst if tl); hi = max(tl[-1]["t1"] for tl in lst if tl)
    d = 0; last = lo; acc = [0, 0, 0, 0]
    for t, dd in iv:
        acc[min(d, 3)] += t - last; last = t; d += dd
    acc[min(d, 3)] += hi - last
    fr.append([a / (hi - lo) for a in acc])
fr = np.array(fr).mean(0)
print(f"per CU ({np.mean([len(l) for l in bycu.values()]):.1f} workgroups): fraction of time with 0 / 1 / 2 / >=3 workgroups inside a K loop: " + " / ".join(f"{x:.2f}" for x in fr))
