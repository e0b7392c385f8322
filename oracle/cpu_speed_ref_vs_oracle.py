"""Record, once, in the build container: the CPU oracle runs at the speed of the REAL reference and
gives the same samples (bench.py's cpu_baseline is kind "port" because the reference's
files never travel to the GPU box).  TEST INFRASTRUCTURE ONLY.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python oracle/cpu_speed_ref_vs_oracle.py

Writes profiles/r2_cpu_ref_vs_oracle.json.  torchvision is stubbed as in oracle/gen_golden.py.
"""
import json
import os
import sys
import time
import types

for _m in ("torchvision", "torchvision.datasets", "torchvision.transforms", "torchvision.utils"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
import matplotlib

matplotlib.use("Agg")
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusion.model import UNetMini
from diffusion.sampler import DDIMSampler
from diffusion.schedulers import DDPMScheduler
from diffusion.utils import set_seed
from oracle import unet_oracle as O

torch.set_grad_enabled(False)
B, n = 64, 50
set_seed(42)
model = UNetMini(image_size=32).eval()
sd = {k: v.clone() for k, v in model.state_dict().items()}
smp = DDIMSampler(DDPMScheduler(), num_inference_steps=n, eta=0.0)
osch = O.make_schedule()


def ref_run():
    torch.manual_seed(7)
    return smp(model, (B, 1, 32, 32), progress_bar=False)


def ora_run():
    torch.manual_seed(7)
    return O.sample(sd, osch, (B, 1, 32, 32), n, 0.0)


ref_run(); ora_run()  # warm-up
tr, to = [], []
for _ in range(3):  # interleaved, so both see the same machine state
    t0 = time.perf_counter(); a = ref_run(); tr.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); b = ora_run(); to.append(time.perf_counter() - t0)
res = {"workload": f"DDIM-{n} eta=0, batch {B}, 32x32, fp32, seed-42 random-init weights", "threads": torch.get_num_threads(),
       "reference_s": min(tr), "oracle_s": min(to), "reference_images_per_s": B / min(tr), "oracle_images_per_s": B / min(to),
       "oracle_over_reference_time": min(to) / min(tr), "max_abs_diff_samples": (a - b).abs().max().item(),
       "where": "build container (CPU only), torch " + torch.__version__}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(ROOT, "profiles", "r2_cpu_ref_vs_oracle.json"), "w"), indent=1)
