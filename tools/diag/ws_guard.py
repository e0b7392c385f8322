# does dm_sample write outside, or use data from outside, the workspace it was given?  The workspace is a slice of a larger
# buffer whose guard zones are filled with a pattern (0xFF.. = NaN / 0x00): guards must come back untouched, results must not
# depend on the pattern.
import sys
sys.path.insert(0, '.')
import torch
import diffusion_model_amd as d
from diffusion_model_amd import _lib
d.set_seed(42)
L = _lib.lib()
G = 64 << 20
for (B, n, eta, prec) in ((32, 100, 1.0, "fp32"), (64, 60, 1.0, "fp32"), (5, 50, 0.0, "fp32"), (32, 50, 0.5, "bf16")):
    m = d.UNetMini(image_size=32, precision=prec).eval()
    smp = d.AncestralSampler(d.DDPMScheduler(), num_inference_steps=n, eta=eta)
    g = torch.Generator().manual_seed(1)
    x_T = torch.randn((B, 1, 32, 32), generator=g); nz = torch.randn((smp.scheduler.timesteps.numel(), B, 1, 32, 32), generator=g).cuda()
    need = L.dm_sample_workspace_bytes(m._native(), B, 32, 32, smp.scheduler.timesteps.numel(), 0, 1 if eta > 0 else 0)
    outs = []
    for pat in (0xFF, 0x00, 0x7F):
        big = torch.full((need + 2 * G,), pat, dtype=torch.uint8, device="cuda")
        m._ws = big[G:G + need]
        o = smp.sample_from(m, x_T, nz if eta > 0 else None); torch.cuda.synchronize()
        lo_ok = bool((big[:G] == pat).all()); hi_ok = bool((big[G + need:] == pat).all())
        outs.append(o)
        print(f"B={B} n={n} {prec} pattern {pat:#x}: guards intact below/above: {lo_ok}/{hi_ok}; finite: {bool(torch.isfinite(o).all())}")
        if not hi_ok:
            bad = (big[G + need:] != pat).nonzero().flatten()
            print("   first/last overwritten byte above:", int(bad[0]), int(bad[-1]), "count", bad.numel())
        if not lo_ok:
            bad = (big[:G] != pat).nonzero().flatten()
            print("   first/last overwritten byte below:", int(bad[0]) - G, int(bad[-1]) - G, "count", bad.numel())
    print("   results independent of the pattern:", torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]))
