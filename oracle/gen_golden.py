"""Generate golden fixtures by running the REAL reference (CPU) in the build container.

TEST INFRASTRUCTURE ONLY.  Run once, here (the reference never travels to the GPU box):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

The reference's ``diffusion/utils.py`` imports torchvision at module top for functions that
are off the hot path; empty stand-in modules are pre-seeded so the import succeeds
(SURVEY 8(c)).  Nothing from the reference is copied: only its outputs (weights it
initialised, tensors it computed) are written to ``tests/golden/``.
"""
import os
import sys
import types

for _m in ("torchvision", "torchvision.datasets", "torchvision.transforms", "torchvision.utils"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
import matplotlib

matplotlib.use("Agg")
import torch

from diffusion.model import UNetMini
from diffusion.sampler import AncestralSampler, DDIMSampler
from diffusion.schedulers import DDPMScheduler
from diffusion.utils import set_seed
from diffusion.conditioning import ClassConditioner

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_grad_enabled(False)

set_seed(42)
model = UNetMini(image_size=32).eval()
sd = {k: v.clone() for k, v in model.state_dict().items()}
# checkpoint dict in the reference's format (utils.py:119-160 keys that sample.py:61-85 reads)
torch.save({"epoch": 0, "model_state_dict": sd, "ema_shadow": None, "loss": 0.0},
           os.path.join(OUT, "unet32_seed42.pth"))

G = {}
# ---- schedule tables (linear + cosine)
for name in ("linear", "cosine"):
    s = DDPMScheduler(beta_schedule=name)
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "posterior_variance"):
        G[f"sched_{name}_{k}"] = getattr(s, k).clone()
s = DDPMScheduler()
s.set_timesteps(50)
G["timesteps_50"] = s.timesteps.clone()
s.set_timesteps(30)  # n does not divide T -> more than n steps
G["timesteps_30"] = s.timesteps.clone()

# ---- single forwards: 0-dim t (sampling) and [B] t (training style), with block taps
B = 4
torch.manual_seed(1234)
x = torch.randn(B, 1, 32, 32)
G["fwd_x"] = x
for tval in (0, 500, 999):
    G[f"fwd_eps_t{tval}"] = model(x, torch.tensor(tval))
tb = torch.tensor([3, 250, 777, 999])
G["fwd_tb"] = tb
G["fwd_eps_tb"] = model(x, tb)
G["fwd_temb_t500"] = model.time_mlp(torch.tensor(500))
h = model.init_conv(x)
G["fwd_init"] = h
d1 = model.down1(h, G["fwd_temb_t500"])
G["fwd_down1_t500"] = d1


# ---- samplers: record (x_t, eps_t) at every step via a model wrapper
class Rec:
    def __init__(self, m):
        self.m, self.xs, self.es = m, [], []

    def __call__(self, x, t):
        e = self.m(x, t)
        self.xs.append(x.clone())
        self.es.append(e.clone())
        return e


def run(sampler_cls, n, eta, Bs, seed, keep):
    sch = DDPMScheduler()
    smp = sampler_cls(sch, num_inference_steps=n, eta=eta)
    rec = Rec(model)
    # replay the global-generator draw order to capture the noise
    torch.manual_seed(seed)
    x_T = torch.randn((Bs, 1, 32, 32))
    noises = [torch.randn((Bs, 1, 32, 32)) for _ in range(len(sch.timesteps))] if eta > 0 else []
    torch.manual_seed(seed)
    out = smp(rec, (Bs, 1, 32, 32), progress_bar=False)
    assert torch.equal(rec.xs[0], x_T)
    idx = list(range(len(rec.xs))) if keep is None else keep
    d = dict(x_T=x_T, out=out, steps=torch.tensor(idx),
             xs=torch.stack([rec.xs[i] for i in idx]), es=torch.stack([rec.es[i] for i in idx]))
    if eta > 0:
        d["noises_kept"] = torch.stack([noises[i] for i in idx])
    return d, noises


d, _ = run(DDIMSampler, 50, 0.0, 4, 42, None)
for k, v in d.items():
    G[f"ddim50_{k}"] = v
d, nz = run(DDIMSampler, 50, 0.5, 2, 42, None)
for k, v in d.items():
    G[f"ddim50eta_{k}"] = v
# ancestral T=1000 eta=1: B=2; keep first 5 / a middle few / last 5 steps for teacher forcing.
keep = [0, 1, 2, 3, 4, 498, 499, 500, 995, 996, 997, 998, 999]
d, nz = run(AncestralSampler, 1000, 1.0, 2, 42, keep)
for k, v in d.items():
    G[f"anc1000_{k}"] = v
# ancestral short free-running chain with full noise kept: n=20 eta=1
d, nz = run(AncestralSampler, 20, 1.0, 2, 7, None)
for k, v in d.items():
    G[f"anc20_{k}"] = v

# ---- ClassConditioner (conditioning.py:8-55) weights + output, for the build-defined cfg4 path
set_seed(43)
cc = ClassConditioner(11, 128).eval()
torch.save({k: v.clone() for k, v in cc.state_dict().items()}, os.path.join(OUT, "classcond11_seed43.pth"))
lab = torch.tensor([0, 3, 9, 10])
G["cc_labels"] = lab
G["cc_out"] = cc(lab)

torch.save(G, os.path.join(OUT, "golden_unet32.pt"))
print({k: tuple(v.shape) for k, v in G.items()})
print("bytes", os.path.getsize(os.path.join(OUT, "golden_unet32.pt")))
