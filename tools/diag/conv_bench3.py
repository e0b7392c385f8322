# per-layer timing of the 3x3 conv forms (kind 0 direct, kind 5 Winograd) through the C ABI
import os, sys
sys.path.insert(0, '.')
import torch
from diffusion_model_amd import _lib
L = _lib.lib()
dev = 'cuda'
def run(kind, H, B, c2=0, tf=True, stats=True, reps=30):
    x1 = torch.randn(B, H, H, 64, device=dev)
    x2 = torch.randn(B, H, H, 64, device=dev) if c2 else None
    cin = 64 + c2
    w = torch.randn(64, cin, 3, 3, device=dev) * 0.05
    wp = torch.empty(L.dm_conv_packed_floats(kind, cin, 64), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(L.dm_pack_conv_weight(kind, _lib.ptr(w), cin, 64, _lib.ptr(wp), s))
    b = torch.randn(64, device=dev); g = torch.rand(64, device=dev) + .5; be = torch.randn(64, device=dev); te = torch.randn(64, device=dev)
    pin = L.dm_conv_stats_partials(kind, H, H)
    si = torch.zeros(B, pin, 8, 2, device=dev); si[..., 1] = H * H * 8 / pin
    y = torch.empty(B, H, H, 64, device=dev)
    so = torch.empty(B, pin, 8, 2, device=dev)
    tf = tf and not c2
    def call():
        _lib.check(L.dm_conv2d(kind, _lib.ptr(x1), 64, _lib.ptr(x2), c2, _lib.ptr(wp), _lib.ptr(b), 64, _lib.ptr(si) if tf else None, pin,
                               H * H * 8 / pin, _lib.ptr(g), _lib.ptr(be), _lib.ptr(te), 0, _lib.ptr(y), _lib.ptr(so) if stats else None, B, H, H, s))
    for _ in range(30): call()   # warm the clock
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * B * H * H * 64 * 9 * cin
    ex = fl * (4 / 9 if kind == 5 else 1)
    print(f"kind={kind} H={H} cin={cin} tf={int(tf)} stats={int(stats)}: {us:8.1f} us  alg {fl/us/1e6:6.1f} TF  executed-MFMA {ex/us/1e6/157.3*100:.0f}% of peak", flush=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for kind in (5, 0):
    run(kind, 32, B); run(kind, 32, B, tf=False, stats=False); run(kind, 16, B); run(kind, 16, B, c2=64); run(kind, 8, B); run(kind, 8, B, c2=64)
