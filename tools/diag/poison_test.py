# which kernels use LDS words they never wrote?  LDS on every CU is poisoned (NaN pattern) before, and beside, each launch.
import sys, os, ctypes as C
sys.path.insert(0, '.')
import torch
import diffusion_model_amd as d
from diffusion_model_amd import _lib
P = C.CDLL(os.path.abspath("tools/variants/liblds_poison.so"))
P.lds_poison.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_int, C.c_void_p]
L = _lib.lib(); dev = "cuda"
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def poison_all(stream, pattern=0x7FC12345):
    # 160 KB per workgroup: one workgroup per CU, 2048 of them sweep every CU several times
    assert P.lds_poison(2048, 160 * 1024, pattern, 0, stream.cuda_stream) == 0
def poison_beside(stream, pattern=0x7FC12345):
    assert P.lds_poison(4096, 48 * 1024, pattern, 2, stream.cuda_stream) == 0
ns = {}
exec(open('tools/diag/corun_pairs.py').read().split("victims =")[0].replace("s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()", ""), ns)
Conv = ns["Conv"]
def check(name, fn):
    with torch.cuda.stream(s1): ref = fn()
    torch.cuda.synchronize()
    ref = [r.clone() for r in ref]
    res = []
    for mode in ("before", "beside"):
        bad = 0; nan = 0
        for trial in range(10):
            if mode == "before":
                poison_all(s1)
            else:
                poison_beside(s2)
            with torch.cuda.stream(s1): out = fn()
            torch.cuda.synchronize()
            if not all(torch.equal(o, r) for o, r in zip(out, ref)): bad += 1
            if any(torch.isnan(o).any() for o in out): nan += 1
        res.append(f"{mode}: {bad}/10 differ ({nan} with NaN)")
    print(f"{name:34s} " + "; ".join(res), flush=True)
for kind, H in ((5, 32), (5, 16), (5, 8), (0, 4), (0, 32), (1, 32), (1, 8), (6, 32), (2, 16), (2, 4), (7, 4), (7, 32), (8, 32), (8, 16), (8, 8), (9, 16), (9, 8), (9, 4)):
    v = Conv(kind, H, seed=kind * 10 + H)
    P_ = L.dm_conv_stats_partials(kind, H, H)
    def fn(v=v, P_=P_):
        v.go(1); return [v.y, v.so[:, :P_]]
    check(f"conv kind {kind} H={H}", fn)
d.set_seed(42)
for prec in ("fp32", "bf16"):
    m = d.UNetMini(image_size=32, precision=prec).eval()
    x = torch.randn(32, 1, 32, 32, device=dev); t = torch.tensor(321)
    check(f"model forward {prec} B=32", lambda: [m(x, t)])
    x5 = torch.randn(5, 1, 32, 32, device=dev)
    check(f"model forward {prec} B=5", lambda: [m(x5, t)])
