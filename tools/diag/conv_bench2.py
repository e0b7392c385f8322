import os, sys
sys.path.insert(0, '.')
import torch
from diffusion_model_amd import _lib
L = _lib.lib()
dev = 'cuda'
def run(kind, H, B, tf=True, stats=True, reps=30):
    OH = H if kind == 0 else H // 2 if kind == 1 else 2 * H
    x1 = torch.randn(B, H, H, 64, device=dev)
    k = 3 if kind == 0 else 4
    w = torch.randn(64, 64, k, k, device=dev) * 0.05
    wp = torch.empty(L.dm_conv_packed_floats(kind, 64, 64), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(L.dm_pack_conv_weight(kind, _lib.ptr(w), 64, 64, _lib.ptr(wp), s))
    b = torch.randn(64, device=dev); g = torch.rand(64, device=dev) + .5; be = torch.randn(64, device=dev); te = torch.randn(64, device=dev)
    pin = L.dm_conv_stats_partials(0, H, H)
    si = torch.zeros(B, pin, 8, 2, device=dev); si[..., 1] = H * H * 8 / pin
    y = torch.empty(B, OH, OH, 64, device=dev)
    so = torch.empty(B, L.dm_conv_stats_partials(kind, H, H), 8, 2, device=dev)
    def call():
        _lib.check(L.dm_conv2d(kind, _lib.ptr(x1), 64, None, 0, _lib.ptr(wp), _lib.ptr(b), 64, _lib.ptr(si) if tf else None, pin,
                               H * H * 8 / pin, _lib.ptr(g), _lib.ptr(be), _lib.ptr(te), 0, _lib.ptr(y), _lib.ptr(so) if stats else None, B, H, H, s))
    for _ in range(30): call()   # warm the clock
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    taps = 9 if kind == 0 else 16 if kind == 1 else 16
    opix = B * (H * H if kind != 1 else H * H // 4)
    fl = 2.0 * opix * 64 * taps * 64
    print(f"[{os.environ.get('DM_SAMPLER_LIB','default')[-14:]}] kind={kind} H={H} tf={int(tf)} stats={int(stats)}: {us:8.1f} us {fl/us/1e6:6.1f} TF ({fl/us/1e6/157.3*100:.0f}%)")
run(0, 32, 1024); run(0, 32, 1024, tf=False, stats=False); run(0, 16, 1024); run(1, 32, 1024); run(2, 16, 1024)
