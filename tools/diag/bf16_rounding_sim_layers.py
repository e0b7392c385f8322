"""Per-layer attribution of bf16 operand rounding (stage + weights), fp16 storage everywhere."""
import sys
import torch, torch.nn.functional as F
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from oracle import unet_oracle as O
torch.set_grad_enabled(False)
G = torch.load(ROOT + '/tests/golden/golden_unet32.pt', weights_only=True)
sd = torch.load(ROOT + '/tests/golden/unet32_seed42.pth', weights_only=True)['model_state_dict']
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
h16 = lambda t_: t_.clamp(-65504, 65504).to(torch.float16).to(torch.float32)
LAYERS = ["down1", "down2", "down3", "bott", "up1", "up2", "up3"]

def fwd(x, t, lay_w, lay_a):
    """lay_w / lay_a: sets of layer-group names whose weights / staged activations are rounded to bf16."""
    E = 128
    te = O.time_mlp(sd, t, E)
    def W(name, k): return bf(sd[k]) if name in lay_w else sd[k]
    def A(name, v): return bf(v) if name in lay_a else v
    def gn_apply(raw, stored, g, b):
        B = raw.shape[0]
        r = raw.reshape(B, 8, -1); mu = r.mean(2); var = r.var(2, unbiased=False)
        s = stored.reshape(B, 8, -1)
        y = ((s - mu[..., None]) * torch.rsqrt(var[..., None] + 1e-5)).reshape(stored.shape)
        return y * g[None, :, None, None] + b[None, :, None, None]
    h = F.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=1)
    def block(name, xin, up):
        p = name + "."
        xin = A(name, xin)
        raw1 = F.conv2d(xin, W(name, p + "conv1.weight"), sd[p + "conv1.bias"], padding=1)
        a = F.silu(gn_apply(raw1, h16(raw1), sd[p + "bnorm1.weight"], sd[p + "bnorm1.bias"]))
        tt = F.silu(F.linear(te, sd[p + "time_mlp.weight"], sd[p + "time_mlp.bias"]))
        a = A(name, a + tt[(...,) + (None,) * 2])
        raw2 = F.conv2d(a, W(name, p + "conv2.weight"), sd[p + "conv2.bias"], padding=1)
        a = A(name, F.silu(gn_apply(raw2, h16(raw2), sd[p + "bnorm2.weight"], sd[p + "bnorm2.bias"])))
        if up: return F.conv_transpose2d(a, W(name, p + "transform.weight"), sd[p + "transform.bias"], stride=2, padding=1)
        return F.conv2d(a, W(name, p + "transform.weight"), sd[p + "transform.bias"], stride=2, padding=1)
    d1 = block("down1", h, False); d2 = block("down2", d1, False); d3 = block("down3", d2, False)
    rawb = F.conv2d(A("bott", d3), W("bott", "bottleneck.0.weight"), sd["bottleneck.0.bias"], padding=1)
    b = A("bott", F.silu(gn_apply(rawb, h16(rawb), sd["bottleneck.1.weight"], sd["bottleneck.1.bias"])))
    bt = F.conv2d(b, W("bott", "bottleneck.3.weight"), sd["bottleneck.3.bias"], padding=1)
    u1 = block("up1", torch.cat([bt, d3], 1), True)
    u2 = block("up2", torch.cat([u1, d2], 1), True)
    u3 = block("up3", torch.cat([u2, d1], 1), True)
    o = F.silu(gn_apply(u3, h16(u3), sd["final_conv.0.weight"], sd["final_conv.0.bias"]))
    return F.conv2d(o, sd["final_conv.2.weight"], sd["final_conv.2.bias"])

ts = G["timesteps_50"]
def worst(lw, la):
    w = 0.0
    for i in range(0, 50, 4):
        e = fwd(G["ddim50_xs"][i], ts[i], set(lw), set(la))
        w = max(w, (e - G["ddim50_es"][i]).abs().max().item())
    return w
print("none:", worst([], []))
print("all :", worst(LAYERS, LAYERS))
for l in LAYERS:
    print(f"{l:6s} only: weights {worst([l], []):.3e}  acts {worst([], [l]):.3e}  both {worst([l],[l]):.3e}")
for keep in (["up3"], ["up3", "down1"], ["up3", "up2"], ["up3", "up2", "down1"]):
    rest = [l for l in LAYERS if l not in keep]
    print("all but", keep, worst(rest, rest), " w split only there:", worst(rest, LAYERS), " a split only there:", worst(LAYERS, rest))
print("weights split everywhere:", worst([], LAYERS), " acts split everywhere:", worst(LAYERS, []))
