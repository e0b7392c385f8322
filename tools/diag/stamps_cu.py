import os, sys, ctypes as C
os.environ["DM_SAMPLER_LIB"] = "tools/diag/build/libdm_stamps.so"
sys.path.insert(0, '.')
import torch, numpy as np
from diffusion_model_amd import _lib
L = _lib.lib()
L.dm_debug_set_stamps.argtypes = [C.c_void_p]
dev = 'cuda'
B, H = 1024, 32
x1 = torch.randn(B, H, H, 64, device=dev)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
wp = torch.empty(L.dm_conv_packed_floats(0, 64, 64), device=dev)
s = torch.cuda.current_stream().cuda_stream
_lib.check(L.dm_pack_conv_weight(0, _lib.ptr(w), 64, 64, _lib.ptr(wp), s))
b = torch.randn(64, device=dev); g = torch.rand(64, device=dev) + .5; be = torch.randn(64, device=dev); te = torch.randn(64, device=dev)
pin = L.dm_conv_stats_partials(0, H, H)
si = torch.zeros(B, pin, 8, 2, device=dev); si[..., 1] = H * H * 8 / pin
y = torch.empty(B, H, H, 64, device=dev); so = torch.empty(B, pin, 8, 2, device=dev)
st = torch.zeros(1 << 21, dtype=torch.int64, device=dev)
def call():
    _lib.check(L.dm_conv2d(0, _lib.ptr(x1), 64, None, 0, _lib.ptr(wp), _lib.ptr(b), 64, _lib.ptr(si), pin, H * H * 8 / pin,
                           _lib.ptr(g), _lib.ptr(be), _lib.ptr(te), 0, _lib.ptr(y), _lib.ptr(so), B, H, H, s))
for _ in range(30): call()       # keep the GPU busy so the clock is up
L.dm_debug_set_stamps(st.data_ptr())
call(); torch.cuda.synchronize()
L.dm_debug_set_stamps(None)
a = st.cpu().numpy().reshape(-1, 8)[:8192]
clk = (a[:, 3] - a[:, 0]) / np.maximum(a[:, 5] - a[:, 4], 1) * 100.0
print("clock MHz median", np.median(clk))
hw = a[:, 6]
xcc = hw >> 32; hwid = hw & 0xffffffff
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
key = xcc * 4096 + se * 64 + sh * 16 + cu
keys = np.unique(key)
print("distinct CUs", len(keys), "WGs per CU", np.bincount(np.unique(key, return_inverse=True)[1]).mean())
cov, util, gaps = [], [], []
for k in keys[:256]:
    r = a[key == k]
    t0, t1 = r[:, 0].min(), r[:, 3].max()
    ev = []
    for row in r: ev.append((row[1], 1)); ev.append((row[2], -1))
    ev.sort()
    busy = 0; depth = 0; last = None
    for t, dlt in ev:
        if depth > 0: busy += t - last
        depth += dlt; last = t
    cov.append(busy / (t1 - t0)); util.append(len(r) * 36864.0 / (t1 - t0))
print("per CU: fraction of span with >=1 WG in its K loop: median %.3f p10 %.3f" % (np.median(cov), np.percentile(cov, 10)))
print("per CU: MFMA need / span (36864 cyc per WG-wave): median %.3f" % np.median(util))
k = keys[3]; r = a[key == k]; r = r[np.argsort(r[:, 0])]; t0 = r[:, 0].min()
print("timeline of one CU (start, stage_end, k_end, end) in kcycles:")
for row in r[:14]: print("   ", [round((row[j] - t0) / 1000, 1) for j in range(4)])
