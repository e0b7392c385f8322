# Accuracy of an fp32 dot product emulated with TWO-way fp16 splits (a = a1 + a2, b = 2^-k (b1 + b2) with the weights scaled by a
# power of two into fp16's normal range; products a1 b1 + a1 b2 + a2 b1, fp32 accumulation) against the three-way bf16 split
# (six products), the two-way bf16 split (three products) and fp32-MFMA-like accumulation; truth = fp64.
import sys
import torch
RTZ = len(sys.argv) > 1 and sys.argv[1] == 'rtz'   # first term rounded toward zero (as the kernels do), second to nearest
torch.manual_seed(0)
def split_bf(x, n):
    out = []; r = x.clone()
    for _ in range(n):
        t = r.bfloat16().float(); out.append(t); r = r - t
    return out
def rtz_half(x):  # v_cvt_pkrtz_f16_f32: round toward zero (saturating)
    h = x.half()
    over = h.float().abs() > x.abs()
    hb = h.view(torch.int16)
    hb = torch.where(over, hb - 1, hb)   # one ulp toward zero (sign-magnitude encoding)
    return hb.view(torch.half).float()
def split_h(x, rtz=RTZ):
    a1 = rtz_half(x) if rtz else x.half().float(); r = x - a1
    return a1, r.half().float()
def mm32(a, b, chunk):
    acc = torch.zeros(a.shape[0], b.shape[1])
    for k in range(0, a.shape[1], chunk):
        acc = acc + (a[:, k:k + chunk].double() @ b[k:k + chunk].double()).float()
    return acc
def emu(terms, K):
    acc = torch.zeros(terms[0][0].shape[0], terms[0][1].shape[1])
    for k in range(0, K, 16):
        s = slice(k, k + 16)
        for (p, q) in terms:
            acc = acc + (p[:, s].double() @ q[s].double()).float()
    return acc
def report(name, a, b):
    K = a.shape[1]
    ref = a.double() @ b.double()
    f32 = mm32(a, b, 2)
    a1, a2, a3 = split_bf(a, 3); b1, b2, b3 = split_bf(b, 3)
    x6 = emu(((a1, b3), (a2, b2), (a1, b2), (a3, b1), (a2, b1), (a1, b1)), K)
    x3 = emu(((a1, b2), (a2, b1), (a1, b1)), K)
    kk = 10 - int(torch.floor(torch.log2(b.abs().max())).item())   # max |w| 2^kk in [2^10, 2^11)
    bs = b * 2.0 ** kk
    h1, h2 = split_h(a); g1, g2 = split_h(bs)
    hx = emu(((h1, g2), (h2, g1), (h1, g1)), K) * 2.0 ** -kk
    e = lambda y: ((y - ref).abs().max().item(), (y - ref).pow(2).mean().sqrt().item())
    sa = (a.abs().double() @ b.abs().double()).max().item()
    print(f"{name:34s} K={K:5d} |ref|max {ref.abs().max():8.2f}  max / rms error:  fp32-MFMA-like {e(f32)[0]:.2e} / {e(f32)[1]:.2e}   bf16x6 {e(x6)[0]:.2e} / {e(x6)[1]:.2e}"
          f"   f16x3 {e(hx)[0]:.2e} / {e(hx)[1]:.2e}   bf16x3 {e(x3)[0]:.2e} / {e(x3)[1]:.2e}   (max sum|a||b| {sa:.1f})")
for K in (576, 1024, 1152, 4096):
    a = torch.randn(512, K) * torch.rand(512, 1) * 3
    a = torch.nn.functional.silu(a) + 0.3 * torch.randn(1, K)
    report("GN+SiLU-like activations", a, torch.randn(K, 64) * 0.05)
# badly scaled operands (ten binades each)
K = 1024
a = torch.randn(512, K) * 2.0 ** torch.randint(-8, 3, (512, K)).float()
b = torch.randn(K, 64) * 0.05 * 2.0 ** torch.randint(-8, 3, (K, 64)).float()
report("operands over ten binades", a, b)
# tiny activations (normalisation absent): the fp16 split loses relative accuracy here (its residual term goes subnormal)
report("activations ~1e-3 (no GroupNorm)", torch.randn(512, K) * 1e-3, torch.randn(K, 64) * 0.05)
report("activations ~1e+3 (no GroupNorm)", torch.randn(512, K) * 1e3, torch.randn(K, 64) * 0.05)
