# Accuracy of an fp32 dot product emulated with three-way bf16 splits (a = a1+a2+a3, b likewise; products
# a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1, fp32 accumulation) against plain fp32 accumulation; truth = fp64.
import torch
torch.manual_seed(0)
def split3(x):
    a1 = x.bfloat16().float(); r = x - a1
    a2 = r.bfloat16().float(); r = r - a2
    a3 = r.bfloat16().float()
    return a1, a2, a3
def mm32(a, b, chunk):  # fp32 accumulation in chunks of `chunk` along K (each chunk's partial rounded to fp32)
    acc = torch.zeros(a.shape[0], b.shape[1])
    for k in range(0, a.shape[1], chunk):
        acc = acc + (a[:, k:k + chunk].double() @ b[k:k + chunk].double()).float()
    return acc
for K in (576, 1024, 1152):
    a = torch.randn(512, K) * torch.rand(512, 1) * 3      # activations after GN+SiLU-ish scale
    a = torch.nn.functional.silu(a) + 0.3 * torch.randn(1, K)
    b = torch.randn(K, 64) * 0.05
    ref = a.double() @ b.double()
    f32 = mm32(a, b, 2)                                    # v_mfma_f32_32x32x2_f32: two products per accumulate
    a1, a2, a3 = split3(a); b1, b2, b3 = split3(b)
    x6 = torch.zeros(512, 64)
    for k in range(0, K, 16):
        s = slice(k, k + 16)
        for (p, q) in ((a1, b1), (a1, b2), (a2, b1), (a2, b2), (a1, b3), (a3, b1)):
            x6 = x6 + (p[:, s].double() @ q[s].double()).float()
    x3 = torch.zeros(512, 64)
    for k in range(0, K, 16):
        s = slice(k, k + 16)
        for (p, q) in ((a1, b1), (a1, b2), (a2, b1)):
            x3 = x3 + (p[:, s].double() @ q[s].double()).float()
    sc = ref.abs().max().item()
    print(f"K={K}: |ref|max {sc:.2f}; max-abs error fp32-MFMA-like {(f32 - ref).abs().max():.2e}, bf16x6 {(x6 - ref).abs().max():.2e}, bf16x3 {(x3 - ref).abs().max():.2e}")
