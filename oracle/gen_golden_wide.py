"""Golden vectors for the widened config (BASELINE configs[4] without the build-defined attention
block): the REAL reference UNetMini(image_size=64, in_channels=3, out_channels=3,
model_channels=256) on CPU.  The 17.6 M weights are not stored: they are the constructor's
seeded init (set_seed(7)), which the drop-in constructor reproduces; two weight checksums in the
fixture guard that assumption.  Run as oracle/gen_golden.py."""
import os
import sys
import types

for _m in ("torchvision", "torchvision.datasets", "torchvision.transforms", "torchvision.utils"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
import matplotlib

matplotlib.use("Agg")
import torch

from diffusion.model import UNetMini
from diffusion.utils import set_seed

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
torch.set_grad_enabled(False)
set_seed(7)
m = UNetMini(image_size=64, in_channels=3, out_channels=3, model_channels=256).eval()
g = torch.Generator().manual_seed(8)
x = torch.randn((2, 3, 64, 64), generator=g)
G = {"x": x, "eps_t250": m(x, torch.tensor(250)), "eps_tb": m(x, torch.tensor([3, 900])),
     "tb": torch.tensor([3, 900]),
     "chk_up3_conv1": m.up3.conv1.weight.double().sum(), "chk_final": m.final_conv[2].weight.double().sum(),
     "nparams": torch.tensor(sum(p.numel() for p in m.parameters()))}
torch.save(G, os.path.join(OUT, "golden_unet64_wide.pt"))
print({k: (tuple(v.shape) if v.dim() else v.item()) for k, v in G.items()})
