"""Golden vectors for geometries outside the tuned kernels' set (VERDICT r2 "missing 1"): the REAL reference UNetMini on
CPU for image sides that are not powers of two and model_channels outside {64,128,256,512} - the reference accepts any
H, W divisible by 8 and any C divisible by 8 (model.py:72-73,120-151).  Weights are not stored: they are the constructor's
seeded init, which the drop-in constructor reproduces; a checksum per model guards that.  Also one short DDIM chain per
model (sampler.py:136-172) from a stored x_T.  TEST INFRASTRUCTURE ONLY.  Run once, in the build container:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_geometries.py
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
from diffusion.sampler import DDIMSampler
from diffusion.schedulers import DDPMScheduler
from diffusion.utils import set_seed

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
torch.set_grad_enabled(False)

# (image_size, in_channels, out_channels, model_channels, (H, W) of the input, batch)
CASES = [
    (24, 1, 1, 32, (24, 24), 3),
    (48, 1, 1, 64, (48, 48), 2),     # tuned channel count, image side not a power of two
    (40, 3, 3, 96, (40, 40), 2),
    (56, 2, 5, 72, (56, 24), 2),     # non-square, in != out channels, 9 channels per group
    (32, 1, 1, 40, (32, 32), 3),     # power-of-two image, C outside the tuned set (5 channels per group)
    (96, 3, 3, 192, (96, 96), 1),
    (32, 6, 6, 64, (32, 32), 2),     # tuned C, more than 4 image channels
]
G = {}
for size, cin, cout, C, (H, W), B in CASES:
    key = f"s{size}_i{cin}_o{cout}_c{C}_{H}x{W}"
    set_seed(100 + size + C)
    m = UNetMini(image_size=size, in_channels=cin, out_channels=cout, model_channels=C).eval()
    g = torch.Generator().manual_seed(size * 7 + C)
    x = torch.randn((B, cin, H, W), generator=g)
    G[key + "/x"] = x
    G[key + "/eps_t321"] = m(x, torch.tensor(321))
    tb = torch.randint(0, 1000, (B,), generator=g)
    G[key + "/tb"] = tb
    G[key + "/eps_tb"] = m(x, tb)
    G[key + "/chk"] = torch.stack([m.up3.conv1.weight.double().sum(), m.down2.transform.weight.double().sum(),
                                   m.final_conv[2].weight.double().sum()])
    if cin == cout:  # a short free-running DDIM chain (sampler.py:136-172), replayed from the stored x_T
        smp = DDIMSampler(DDPMScheduler(), num_inference_steps=4, eta=0.0)
        xs = x.clone()
        for i, t in enumerate(smp.scheduler.timesteps):
            xs = smp.step(m(xs, t), t, xs, None)
        G[key + "/ddim4"] = xs
torch.save(G, os.path.join(OUT, "golden_geometries.pt"))
print({k: (tuple(v.shape)) for k, v in G.items()})
print("bytes", os.path.getsize(os.path.join(OUT, "golden_geometries.pt")))
