# one conv layer launched a few times, for rocprofv3 --pmc passes: python3 tools/diag/pmc_layer.py [kind] [H] [B]
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusion_model_amd import _lib
L = _lib.lib()
dev = 'cuda'
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 5
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
x1 = torch.randn(B, H, H, 64, device=dev)
base = {5: 0, 7: 0, 6: 1, 8: 1, 9: 2}.get(kind, kind)   # 0: 3x3, 1: 4x4 stride 2, 2: transposed 4x4
ks = 3 if base == 0 else 4
OH = H // 2 if base == 1 else 2 * H if base == 2 else H
w = torch.randn(64, 64, ks, ks, device=dev) * 0.05
wp = torch.empty(L.dm_conv_packed_floats(kind, 64, 64), device=dev)
s = torch.cuda.current_stream().cuda_stream
_lib.check(L.dm_pack_conv_weight(kind, _lib.ptr(w), 64, 64, _lib.ptr(wp), s))
b = torch.randn(64, device=dev); g = torch.rand(64, device=dev) + .5; be = torch.randn(64, device=dev); te = torch.randn(64, device=dev)
pin = 4
si = torch.zeros(B, pin, 8, 2, device=dev); si[..., 1] = H * H * 8 / pin
y = torch.empty(B, OH, OH, 64, device=dev)
so = torch.empty(B, max(L.dm_conv_stats_partials(kind, H, H), 1), 8, 2, device=dev)
for _ in range(6):
    _lib.check(L.dm_conv2d(kind, _lib.ptr(x1), 64, None, 0, _lib.ptr(wp), _lib.ptr(b), 64, _lib.ptr(si), pin, H * H * 8 / pin,
                           _lib.ptr(g), _lib.ptr(be), _lib.ptr(te), 0, _lib.ptr(y), _lib.ptr(so), B, H, H, s))
torch.cuda.synchronize()
