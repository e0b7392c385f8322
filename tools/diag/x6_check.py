# split-bf16 (x6) conv kernels vs the fp32 kernels through the C ABI: max-abs difference and per-layer time
import os, sys
sys.path.insert(0, '.')
import torch
from diffusion_model_amd import _lib
L = _lib.lib()
dev = 'cuda'
torch.manual_seed(0)
def run(kind, H, B, c2=0, tf=True, stats=True, reps=20, ref_kind=None):
    base = {7: 0, 8: 1, 9: 2}.get(kind, kind)
    base = {5: 0, 6: 1}.get(base, base)
    ks = 3 if base == 0 else 4
    cin = 64 + c2
    OH = H // 2 if base == 1 else 2 * H if base == 2 else H
    x1 = torch.randn(B, H, H, 64, device=dev) * 2 + 0.3
    x2 = torch.randn(B, H, H, 64, device=dev) if c2 else None
    w = (torch.randn(64, cin, ks, ks, device=dev) if base != 2 else torch.randn(cin, 64, ks, ks, device=dev)) * 0.05
    b = torch.randn(64, device=dev); g = torch.rand(64, device=dev) + .5; be = torch.randn(64, device=dev); te = torch.randn(64, device=dev)
    pin = 4
    si = torch.zeros(B, pin, 8, 2, device=dev); si[..., 0] = 0.3 + 0.1 * torch.randn(B, pin, 8, device=dev); si[..., 1] = 4.0 * H * H * 8 / pin
    s = torch.cuda.current_stream().cuda_stream
    tf = tf and not c2
    outs = []
    for k in (kind, ref_kind):
        if k is None: continue
        wp = torch.zeros(L.dm_conv_packed_floats(k, cin, 64), device=dev)
        _lib.check(L.dm_pack_conv_weight(k, _lib.ptr(w), cin, 64, _lib.ptr(wp), s))
        y = torch.zeros(B, OH, OH, 64, device=dev)
        po = L.dm_conv_stats_partials(k, H, H)
        so = torch.zeros(B, max(po, 1), 8, 2, device=dev)
        def call():
            _lib.check(L.dm_conv2d(k, _lib.ptr(x1), 64, _lib.ptr(x2), c2, _lib.ptr(wp), _lib.ptr(b), 64, _lib.ptr(si) if tf else None, pin,
                                   H * H * 8 / pin, _lib.ptr(g), _lib.ptr(be), _lib.ptr(te), 0, _lib.ptr(y), _lib.ptr(so) if stats else None, B, H, H, s))
        for _ in range(5): call()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(reps): call()
        e1.record(); torch.cuda.synchronize()
        outs.append((k, e0.elapsed_time(e1) * 1e3 / reps, y, so, po))
    msg = f"H={H} B={B} cin={cin} tf={int(tf)}:"
    for k, us, y, so, po in outs: msg += f"  kind {k}: {us:7.1f} us"
    if len(outs) == 2:
        d = (outs[0][2] - outs[1][2]).abs().max().item()
        msg += f"   max-abs diff {d:.2e} (|y|max {outs[1][2].abs().max().item():.1f})"
        if stats and outs[0][4] == outs[1][4]:
            msg += f"  stats diff {(outs[0][3] - outs[1][3]).abs().max().item():.2e}"
    print(msg, flush=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
run(7, 32, B, ref_kind=5); run(7, 32, B, ref_kind=0, tf=False); run(7, 16, B, ref_kind=5); run(7, 16, B, c2=64, ref_kind=5)
run(7, 8, B, ref_kind=5); run(7, 8, B, c2=64, ref_kind=5); run(7, 4, B, ref_kind=0); run(7, 4, B, c2=64, ref_kind=0)
run(8, 32, B, ref_kind=6); run(8, 16, B, ref_kind=1); run(8, 8, B, ref_kind=1)
run(9, 16, B, ref_kind=2); run(9, 8, B, ref_kind=2); run(9, 4, B, ref_kind=2)
