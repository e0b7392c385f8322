"""Per-layer conv timing via the C ABI (dm_conv2d) with torch events on the launch stream."""
import sys, torch
sys.path.insert(0, '.')
from diffusion_model_amd import _lib
L = _lib.lib()
dev = 'cuda'
def run(kind, H, B, c2=0, tf=True, stats=True, reps=10):
    c1, cout = 64, 64
    OH = H if kind == 0 else H // 2 if kind == 1 else 2 * H
    x1 = torch.randn(B, H, H, c1, device=dev); x2 = torch.randn(B, H, H, 64, device=dev) if c2 else None
    k = 3 if kind == 0 else 4
    cin = c1 + c2
    w = torch.randn(64, cin, k, k, device=dev) * 0.05 if kind != 2 else torch.randn(cin, 64, 4, 4, device=dev) * 0.05
    wp = torch.empty(L.dm_conv_packed_floats(kind, cin, cout), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(L.dm_pack_conv_weight(kind, _lib.ptr(w), cin, cout, _lib.ptr(wp), s))
    b = torch.randn(64, device=dev); g = torch.rand(64, device=dev) + .5; be = torch.randn(64, device=dev)
    te = torch.randn(64, device=dev)
    pin = L.dm_conv_stats_partials(0, H, H)
    si = torch.zeros(B, pin, 8, 2, device=dev); si[..., 1] = H * H * 8 / pin
    y = torch.empty(B, OH, OH, cout, device=dev)
    pout = L.dm_conv_stats_partials(kind, H, H)
    so = torch.empty(B, pout, 8, 2, device=dev)
    def call():
        _lib.check(L.dm_conv2d(kind, _lib.ptr(x1), c1, _lib.ptr(x2), c2, _lib.ptr(wp), _lib.ptr(b), cout,
                               _lib.ptr(si) if (tf and not c2) else None, pin, H * H * 8 / pin, _lib.ptr(g), _lib.ptr(be),
                               _lib.ptr(te), 0, _lib.ptr(y), _lib.ptr(so) if stats else None, B, H, H, s))
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    taps = 9 if kind == 0 else 16 if kind == 1 else 4
    opix = B * (H * H if kind != 1 else H * H // 4) * (4 if kind == 2 else 1)
    fl = 2.0 * opix * cout * taps * cin
    print(f"kind={kind} H={H:2d} B={B} cin={cin:3d} tf={int(tf)} stats={int(stats)}: {us:8.1f} us  {fl/us/1e6:6.1f} TF  ({fl/us/1e6/157.3*100:.0f}%)")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for tf, st in ((True, True), (False, False)):
    run(0, 32, B, tf=tf, stats=st)
run(0, 16, B); run(0, 16, B, c2=64); run(0, 8, B); run(0, 8, B, c2=64); run(0, 4, B); run(0, 4, B, c2=64)
run(1, 32, B); run(1, 16, B); run(1, 8, B)
run(2, 16, B); run(2, 8, B); run(2, 4, B)
