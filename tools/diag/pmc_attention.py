# the fused attention kernel launched a few times, for rocprofv3 --pmc passes: python3 tools/diag/pmc_attention.py [B] [T] [C]
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from diffusion_model_amd import _lib
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
C = int(sys.argv[3]) if len(sys.argv) > 3 else 256
qkv = torch.randn(B, T, 3 * C, device="cuda")
o = torch.empty(B, T, C, device="cuda")
for _ in range(6):
    _lib.check(L.dm_attention(_lib.ptr(qkv), _lib.ptr(o), B, T, C, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
