# what ARE the wrong values init5 writes beside a bf16 MFMA burner?  Compare with the correct output at other pixels / images / channels.
import sys, os, ctypes as C
sys.path.insert(0, '.')
import torch, torch.nn.functional as F
import diffusion_model_amd as d
A = C.CDLL(os.path.abspath("tools/variants/libaggressors.so"))
A.aggressor.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
dev = "cuda"
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
d.set_seed(42)
m = d.UNetMini(image_size=32).eval()
B = 32
x = torch.randn(B, 1, 32, 32, device=dev); t = torch.tensor(321)
buf = torch.zeros(1 << 20, device=dev)
def al(n): return (n * 4 + 255) // 256 * 256 // 4
full = B * 1024 * 64
off = al(64) + 3 * al(B * 1024) + al(B * 128)
T1 = off + al(full)
with torch.cuda.stream(s1): ref = m(x, t)
torch.cuda.synchronize()
good = m._ws.view(torch.float32)[T1:T1 + full].clone().view(B, 32, 32, 64)
assert A.aggressor(1, 2048, 3000, buf.data_ptr(), buf.numel(), s2.cuda_stream) == 0
with torch.cuda.stream(s1): o = m(x, t)
torch.cuda.synchronize()
bad = m._ws.view(torch.float32)[T1:T1 + full].view(B, 32, 32, 64)
mask = (bad != good); mask[:8] = False   # (images 0..7 of T1 are overwritten by the 16x16 layers)
idx = mask.nonzero()
print("wrong elements:", idx.shape[0], " images:", sorted(set(idx[:, 0].tolist()))[:20])
pix = torch.unique(idx[:, :3], dim=0)
print("wrong pixels:", pix.shape[0], " per-pixel wrong channel counts:", torch.unique(torch.tensor([int(((idx[:, :3] == p).all(1)).sum()) for p in pix[:200]]), return_counts=True))
ys, xs_ = pix[:, 1], pix[:, 2]
print("y % 8 histogram:", torch.bincount(ys % 8, minlength=8).tolist(), " x % 16 histogram:", torch.bincount(xs_ % 16, minlength=16).tolist())
for n, y, xx in pix[:6].tolist():
    g, b_ = good[n, y, xx], bad[n, y, xx]
    ch = (g != b_).nonzero().flatten().tolist()
    c = ch[0]
    print(f"pixel (img {n}, y {y}, x {xx}): wrong channels {ch[:6]}..; c={c}: good {g[c].item():.6f} bad {b_[c].item():.6f}")
    cands = {}
    for dy in range(-2, 3):
        for dx in range(-3, 4):
            yy, x2 = y + dy, xx + dx
            if 0 <= yy < 32 and 0 <= x2 < 32: cands[f"same image (dy {dy}, dx {dx})"] = good[n, yy, x2, c].item()
    for dn in (-8, -4, -2, -1, 1, 2, 4, 8):
        if 8 <= n + dn < B: cands[f"image {n + dn} same pixel"] = good[n + dn, y, xx, c].item()
    best = sorted(cands.items(), key=lambda kv: abs(kv[1] - b_[c].item()))[:3]
    print("     nearest correct values elsewhere:", [(k, round(v, 6)) for k, v in best])
    # is the error explained by ONE tap using a neighbouring input pixel?  error / (x[n, y+dy, x+dx+1] - x[n, y+dy, x+dx]) constant over channels -> weights
    err = (b_ - g)[ch]
    print("     error over the wrong channels (first 6):", [round(v, 5) for v in err[:6].tolist()])
