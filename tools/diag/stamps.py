import os, sys, ctypes as C
os.environ["DM_SAMPLER_LIB"] = "tools/diag/build/libdm_stamps.so"
sys.path.insert(0, '.')
import torch, numpy as np
from diffusion_model_amd import _lib
L = _lib.lib()
L.dm_debug_set_stamps.argtypes = [C.c_void_p]
dev = 'cuda'
def run(kind, H, B, tf=True, stats=True):
    c1 = cout = 64
    OH = H if kind == 0 else H // 2 if kind == 1 else 2 * H
    x1 = torch.randn(B, H, H, c1, device=dev)
    k = 3 if kind == 0 else 4
    w = torch.randn(64, 64, k, k, device=dev) * 0.05
    wp = torch.empty(L.dm_conv_packed_floats(kind, 64, 64), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(L.dm_pack_conv_weight(kind, _lib.ptr(w), 64, 64, _lib.ptr(wp), s))
    b = torch.randn(64, device=dev); g = torch.rand(64, device=dev) + .5; be = torch.randn(64, device=dev); te = torch.randn(64, device=dev)
    pin = L.dm_conv_stats_partials(0, H, H)
    si = torch.zeros(B, pin, 8, 2, device=dev); si[..., 1] = H * H * 8 / pin
    y = torch.empty(B, OH, OH, cout, device=dev)
    pout = L.dm_conv_stats_partials(kind, H, H)
    so = torch.empty(B, pout, 8, 2, device=dev)
    st = torch.zeros(1 << 21, dtype=torch.int64, device=dev)
    def call():
        _lib.check(L.dm_conv2d(kind, _lib.ptr(x1), c1, None, 0, _lib.ptr(wp), _lib.ptr(b), cout,
                               _lib.ptr(si) if tf else None, pin, H * H * 8 / pin, _lib.ptr(g), _lib.ptr(be),
                               _lib.ptr(te), 0, _lib.ptr(y), _lib.ptr(so) if stats else None, B, H, H, s))
    for _ in range(3): call()
    torch.cuda.synchronize()
    L.dm_debug_set_stamps(st.data_ptr())
    call(); torch.cuda.synchronize()
    L.dm_debug_set_stamps(None)
    a = st.cpu().numpy().reshape(-1, 8)
    a = a[a[:, 0] != 0]
    t0 = a[:, 0].min()
    stage, kl, epi = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
    print(f"kind={kind} H={H} tf={int(tf)} stats={int(stats)} WGs={len(a)} span={a[:,3].max()-t0} cyc")
    for nm, v in (("stage", stage), ("kloop", kl), ("epi", epi), ("life", a[:, 3] - a[:, 0])):
        print(f"   {nm:6s} median {np.median(v):9.0f}  p10 {np.percentile(v,10):9.0f}  p90 {np.percentile(v,90):9.0f}")
    clk = (a[:, 3] - a[:, 0]) / np.maximum(a[:, 5] - a[:, 4], 1) * 100.0
    rt = (a[:, 5].max() - a[:, 4].min()) / 100.0
    print(f"   in-kernel clock MHz median {np.median(clk):.0f} p10 {np.percentile(clk,10):.0f} p90 {np.percentile(clk,90):.0f}; kernel span by realtime {rt:.1f} us")
    # start-time distribution of the first 1000 WGs
    starts = np.sort(a[:, 0] - t0)
    print("   starts[0,255,767,768,1000,2000]:", [int(starts[i]) for i in (0, 255, 767, 768, 1000, 2000) if i < len(starts)])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
run(0, 32, B); run(0, 32, B, tf=False, stats=False); run(0, 16, B); run(2, 16, B); run(1, 32, B)
