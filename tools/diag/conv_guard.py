# stray global writes of one conv launch: y and the statistics buffer are slices of a large patterned buffer
import sys
sys.path.insert(0, '.')
import torch
from diffusion_model_amd import _lib
L = _lib.lib(); dev = "cuda"
def run(kind, H, B=32):
    base = {5: 0, 7: 0, 6: 1, 8: 1, 9: 2}.get(kind, kind); ks = 3 if base == 0 else 4
    OH = H // 2 if base == 1 else 2 * H if base == 2 else H
    x1 = torch.randn(B, H, H, 64, device=dev); w = torch.randn(64, 64, ks, ks, device=dev) * 0.05
    wp = torch.empty(L.dm_conv_packed_floats(kind, 64, 64), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(L.dm_pack_conv_weight(kind, _lib.ptr(w), 64, 64, _lib.ptr(wp), s))
    b = torch.randn(64, device=dev); gm = torch.rand(64, device=dev) + .5; be = torch.randn(64, device=dev); te = torch.randn(64, device=dev)
    pin = 4; si = torch.zeros(B, 4, 8, 2, device=dev); si[..., 1] = H * H * 8 / 4
    pout = L.dm_conv_stats_partials(kind, H, H)
    ny, ns = B * OH * OH * 64, B * pout * 16
    G = 4 << 20
    big = torch.full((G + ny + G + ns + G,), 12345.0, device=dev)
    y = big[G:G + ny]; so = big[G + ny + G:G + ny + G + ns]
    for _ in range(3):
        _lib.check(L.dm_conv2d(kind, _lib.ptr(x1), 64, None, 0, _lib.ptr(wp), _lib.ptr(b), 64, _lib.ptr(si), pin, H * H * 8 / pin,
                               _lib.ptr(gm), _lib.ptr(be), _lib.ptr(te), 0, _lib.ptr(y), _lib.ptr(so), B, H, H, s))
    torch.cuda.synchronize()
    g0 = big[:G]; g1 = big[G + ny:G + ny + G]; g2 = big[G + ny + G + ns:]
    msg = []
    for nm, gz in (("below y", g0), ("between y and stats", g1), ("above stats", g2)):
        badm = gz != 12345.0
        if badm.any():
            idx = badm.nonzero().flatten()
            msg.append(f"{nm}: {idx.numel()} floats overwritten, first at +{int(idx[0])}, last at +{int(idx[-1])}")
    print(f"kind {kind} H={H} B={B}: y written {bool((y != 12345.0).all())}, stats written {bool((so != 12345.0).all())}; guards: {'; '.join(msg) if msg else 'intact'}", flush=True)
for kind, H in ((8, 32), (9, 16), (7, 4), (8, 16), (9, 8), (5, 32), (7, 32), (9, 32), (8, 64)):
    run(kind, H)
run(8, 32, 33); run(9, 16, 33); run(9, 16, 1)
