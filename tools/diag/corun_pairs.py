import sys
sys.path.insert(0, '.')
import torch
from diffusion_model_amd import _lib
L = _lib.lib(); dev = "cuda"
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
class Conv:
    def __init__(self, kind, H, B=32, seed=0):
        g = torch.Generator(device=dev).manual_seed(seed)
        base = {5: 0, 7: 0, 6: 1, 8: 1, 9: 2}.get(kind, kind); ks = 3 if base == 0 else 4
        OH = H // 2 if base == 1 else 2 * H if base == 2 else H
        self.kind, self.H, self.B = kind, H, B
        self.x1 = torch.randn(B, H, H, 64, device=dev, generator=g); w = torch.randn(64, 64, ks, ks, device=dev, generator=g) * 0.05
        self.wp = torch.empty(L.dm_conv_packed_floats(kind, 64, 64), device=dev)
        _lib.check(L.dm_pack_conv_weight(kind, _lib.ptr(w), 64, 64, _lib.ptr(self.wp), torch.cuda.current_stream().cuda_stream))
        self.b = torch.randn(64, device=dev, generator=g); self.gm = torch.rand(64, device=dev, generator=g) + .5
        self.be = torch.randn(64, device=dev, generator=g); self.te = torch.randn(64, device=dev, generator=g)
        self.pin = 4; self.si = torch.zeros(B, 4, 8, 2, device=dev); self.si[..., 1] = H * H * 8 / 4
        self.y = torch.empty(B, OH, OH, 64, device=dev); self.so = torch.empty(B, 16, 8, 2, device=dev)
        torch.cuda.synchronize()
    def go(self, reps):
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(reps):
            _lib.check(L.dm_conv2d(self.kind, _lib.ptr(self.x1), 64, None, 0, _lib.ptr(self.wp), _lib.ptr(self.b), 64, _lib.ptr(self.si), self.pin,
                                   self.H * self.H * 8 / self.pin, _lib.ptr(self.gm), _lib.ptr(self.be), _lib.ptr(self.te), 0,
                                   _lib.ptr(self.y), _lib.ptr(self.so), self.B, self.H, self.H, s))
victims = [(5, 32), (5, 16), (5, 8), (0, 4), (7, 4), (8, 32), (8, 16), (8, 8), (9, 16), (9, 8), (9, 4)]
for ak, aH in ((8, 32), (9, 16), (5, 32)):
    agg = Conv(ak, aH, seed=99)
    for vk, vH in victims:
        v = Conv(vk, vH, seed=vk * 10 + vH)
        with torch.cuda.stream(s1): v.go(1)
        torch.cuda.synchronize()
        y0, so0 = v.y.clone(), v.so[:, :L.dm_conv_stats_partials(vk, vH, vH)].clone()
        bad = 0
        for trial in range(30):
            v.y.zero_()
            with torch.cuda.stream(s2): agg.go(30)
            with torch.cuda.stream(s1): v.go(1)
            torch.cuda.synchronize()
            if not (torch.equal(v.y, y0) and torch.equal(v.so[:, :so0.shape[1]], so0)): bad += 1
        print(f"aggressor kind {ak} H={aH}  victim kind {vk} H={vH}: {bad}/30 runs differ from solo", flush=True)
