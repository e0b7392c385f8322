# every conv kernel (one launch) beside a pure bf16 MFMA burner on another stream: output and statistics against the solo launch
import sys, os, ctypes as C
sys.path.insert(0, '.')
import torch
from diffusion_model_amd import _lib
A = C.CDLL(os.path.abspath("tools/variants/libaggressors.so"))
A.aggressor.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
L = _lib.lib()
ns = {}
exec(open('tools/diag/corun_pairs.py').read().split("victims =")[0], ns)
Conv, s1, s2 = ns["Conv"], ns["s1"], ns["s2"]
buf = torch.zeros(1 << 20, device="cuda")
mode = int(os.environ.get("AGG", 1))
for vk, vH in [(5, 32), (5, 16), (5, 8), (0, 32), (0, 4), (1, 32), (1, 8), (6, 32), (2, 16), (2, 4), (7, 32), (7, 4), (8, 32), (8, 16), (8, 8), (9, 16), (9, 8), (9, 4)]:
    v = Conv(vk, vH, seed=vk * 10 + vH)
    with torch.cuda.stream(s1): v.go(1)
    torch.cuda.synchronize()
    P_ = L.dm_conv_stats_partials(vk, vH, vH)
    y0, so0 = v.y.clone(), v.so[:, :P_].clone()
    bad = 0; mx = 0.0
    for trial in range(25):
        v.y.zero_()
        assert A.aggressor(mode, 2048, 600, buf.data_ptr(), buf.numel(), s2.cuda_stream) == 0
        with torch.cuda.stream(s1): v.go(3)
        torch.cuda.synchronize()
        if not (torch.equal(v.y, y0) and torch.equal(v.so[:, :P_], so0)):
            bad += 1; mx = max(mx, (v.y - y0).abs().max().item())
    print(f"aggressor mode {mode}  victim kind {vk} H={vH}: {bad}/25 runs differ from solo (max {mx:.3g})", flush=True)
