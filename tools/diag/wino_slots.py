# one 3x3 layer (kind 5 Winograd / kind 0 direct) timed through the C ABI; run once per DM_DEBUG_SLOT_DIV value
# (1 = 2 persistent workgroups per CU, 2 = one per CU): does the second workgroup per CU overlap or time-share?
import os, sys
sys.path.insert(0, '.')
import torch
from diffusion_model_amd import _lib
L = _lib.lib()
dev = 'cuda'
def run(kind, H, B, tf=True, stats=True, reps=30):
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
    def call():
        _lib.check(L.dm_conv2d(kind, _lib.ptr(x1), 64, None, 0, _lib.ptr(wp), _lib.ptr(b), 64, _lib.ptr(si) if tf else None, pin,
                               H * H * 8 / pin, _lib.ptr(g), _lib.ptr(be), _lib.ptr(te), 0, _lib.ptr(y), _lib.ptr(so) if stats else None, B, H, H, s))
    for _ in range(30): call()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ex = 2.0 * B * H * H * 64 * 9 * 64 * (4 / 9 if kind == 5 else 1)
    print(f"slot_div={os.environ.get('DM_DEBUG_SLOT_DIV','1')} kind={kind} H={H} B={B} tf={int(tf)} stats={int(stats)}: {us:8.1f} us  executed-MFMA {ex/us/1e6/157.3*100:.0f}% of peak", flush=True)
for kind in (5, 0):
    run(kind, 32, 1024); run(kind, 32, 1024, tf=False, stats=False); run(kind, 16, 1024); run(kind, 8, 1024)
