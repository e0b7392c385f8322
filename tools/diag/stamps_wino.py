# phase durations of the LAST tile of every persistent workgroup (stage / K loop / epilogue), -DDM_STAMPS build
import os, sys, ctypes as C
os.environ["DM_SAMPLER_LIB"] = "tools/diag/build/libdm_stamps.so"
sys.path.insert(0, '.')
import torch, numpy as np
from diffusion_model_amd import _lib
L = _lib.lib()
L.dm_debug_set_stamps.argtypes = [C.c_void_p]
dev = 'cuda'
def run(kind, H, B, tf=True, stats=True):
    x1 = torch.randn(B, H, H, 64, device=dev)
    w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    wp = torch.empty(L.dm_conv_packed_floats(kind, 64, 64), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(L.dm_pack_conv_weight(kind, _lib.ptr(w), 64, 64, _lib.ptr(wp), s))
    b = torch.randn(64, device=dev); g = torch.rand(64, device=dev) + .5; be = torch.randn(64, device=dev); te = torch.randn(64, device=dev)
    pin = L.dm_conv_stats_partials(kind, H, H)
    si = torch.zeros(B, pin, 8, 2, device=dev); si[..., 1] = H * H * 8 / pin
    y = torch.empty(B, H, H, 64, device=dev)
    so = torch.empty(B, pin, 8, 2, device=dev)
    st = torch.zeros(1 << 21, dtype=torch.int64, device=dev)
    def call():
        _lib.check(L.dm_conv2d(kind, _lib.ptr(x1), 64, None, 0, _lib.ptr(wp), _lib.ptr(b), 64,
                               _lib.ptr(si) if tf else None, pin, H * H * 8 / pin, _lib.ptr(g), _lib.ptr(be),
                               _lib.ptr(te), 0, _lib.ptr(y), _lib.ptr(so) if stats else None, B, H, H, s))
    for _ in range(20): call()
    torch.cuda.synchronize()
    L.dm_debug_set_stamps(st.data_ptr())
    call(); torch.cuda.synchronize()
    L.dm_debug_set_stamps(None)
    a = st.cpu().numpy().reshape(-1, 8)
    a = a[a[:, 0] != 0]
    stage, kl, epi = a[:, 1] - a[:, 7], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
    ntile_per_wg = (B * H * H // 128) / len(a)
    print(f"kind={kind} H={H} tf={int(tf)} stats={int(stats)} WGs={len(a)} tiles/WG={ntile_per_wg:.1f} life median {np.median(a[:,3]-a[:,0]):.0f} cyc -> {np.median(a[:,3]-a[:,0])/ntile_per_wg:.0f} cyc/tile")
    for nm, v in (("stage", stage), ("kloop", kl), ("epi", epi)):
        print(f"   {nm:6s} median {np.median(v):9.0f}  p10 {np.percentile(v,10):9.0f}  p90 {np.percentile(v,90):9.0f}")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for kind in (5, 0):
    run(kind, 32, B); run(kind, 32, B, tf=False, stats=False); run(kind, 16, B)
