"""Sampling driver (reference: diffusion/sample.py:15-321): checkpoint ingestion with the EMA
override, sample generation with the [0,1] post-processing, image-grid PNG output.

Plotting (``plt.show``) and torchvision are not used: the grid is laid out as
``torchvision.utils.make_grid(nrow, padding=2)`` does and written by a small zlib PNG writer.
The reference's broken entry points (``generate_comparison`` passing kwargs that
``generate_samples`` does not take; CLI default ``--sampler deterministic``) are made to work
rather than reproduced."""
import argparse
import os
import struct
import zlib

import torch

from . import _lib
from .model import UNetMini
from .sampler import create_sampler
from .schedulers import DDPMScheduler
from .utils import get_device, set_seed


def postprocess(samples: torch.Tensor, uint8: bool = False) -> torch.Tensor:
    """(x + 1) / 2 clamped to [0, 1] (sample.py:117-119), one fused kernel; ``uint8`` also
    quantises as ``torchvision.utils.save_image`` would."""
    x = samples.contiguous()
    if x.device.type != "cuda" or x.dtype != torch.float32:
        raise RuntimeError("postprocess expects an fp32 tensor on the GPU")
    out = torch.empty_like(x, dtype=torch.uint8 if uint8 else torch.float32)
    _lib.check(_lib.lib().dm_postprocess(_lib.ptr(x), None if uint8 else _lib.ptr(out), _lib.ptr(out) if uint8 else None,
                                         x.numel(), torch.cuda.current_stream(x.device).cuda_stream))
    return out


def make_grid(images: torch.Tensor, nrow: int = 8, padding: int = 2, pad_value: float = 0.0) -> torch.Tensor:
    """Layout of ``torchvision.utils.make_grid(images, nrow, normalize=False, padding)``:
    [N,C,H,W] -> [3|C, rows*(H+p)+p, cols*(W+p)+p]; single-channel images become 3 channels."""
    t = images.detach().cpu()
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.shape[1] == 1:
        t = t.expand(-1, 3, -1, -1)
    n, c, h, w = t.shape
    if n == 1:
        return t[0]
    xmaps = min(nrow, n)
    ymaps = (n + xmaps - 1) // xmaps
    hh, ww = h + padding, w + padding
    grid = torch.full((c, hh * ymaps + padding, ww * xmaps + padding), pad_value, dtype=t.dtype)
    for k in range(n):
        y, x = divmod(k, xmaps)
        grid[:, y * hh + padding:y * hh + padding + h, x * ww + padding:x * ww + padding + w] = t[k]
    return grid


def save_png(grid: torch.Tensor, filepath: str):
    """[C,H,W] float in [0,1] (or uint8) -> 8-bit PNG (grey or RGB)."""
    g = grid.detach().cpu()
    if g.dtype != torch.uint8:
        g = g.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8)
    c, h, w = g.shape
    if c not in (1, 3):
        raise ValueError("PNG writer takes 1 or 3 channels")
    rows = g.permute(1, 2, 0).contiguous().numpy()
    raw = b"".join(b"\x00" + rows[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0 if c == 1 else 2, 0, 0, 0)) \
        + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
    with open(filepath, "wb") as f:
        f.write(png)


def save_image_grid(images, filepath, nrow=4):
    """diffusion/utils.py:406-419."""
    save_png(make_grid(images, nrow=nrow, padding=2), filepath)


class DDPMSampler:
    """diffusion/sample.py:15-265."""

    def __init__(self, config, sampler_type="ancestral", num_inference_steps=1000, eta=0.1):
        self.config = config
        self.device = get_device()
        self.sampler_type = sampler_type
        self.num_inference_steps = num_inference_steps
        self.eta = eta
        set_seed(config['seed'])
        self.model = UNetMini(image_size=config['image_size'], in_channels=config['in_channels'],
                              out_channels=config['out_channels'], model_channels=config['model_channels'],
                              time_emb_dim=config['time_emb_dim'])
        self.scheduler = DDPMScheduler(num_train_timesteps=config['num_train_timesteps'],
                                       beta_start=config['beta_start'], beta_end=config['beta_end'],
                                       beta_schedule=config['beta_schedule'])
        if config.get('checkpoint_path'):
            self.load_checkpoint(config['checkpoint_path'])

    def load_checkpoint(self, checkpoint_path):
        """sample.py:61-85: ``model_state_dict`` then per-parameter ``ema_shadow`` override.
        Loaded with ``weights_only=True`` (nothing from the file is executed)."""
        ckpt = torch.load(checkpoint_path, map_location=self.device, weights_only=True)
        self.model.load_state_dict(ckpt['model_state_dict'])
        if ckpt.get('ema_shadow'):
            for name, param in self.model.named_parameters():
                if name in ckpt['ema_shadow']:
                    param.data = ckpt['ema_shadow'][name].to(self.device).contiguous()
        self.model.eval()
        return ckpt.get('epoch', 'Unknown')

    def generate_samples(self, num_samples=4, progress_bar=True, sampler_type=None, num_inference_steps=None, eta=None):
        """sample.py:87-121 (the extra keywords are what generate_comparison passes, sample.py:199-205)."""
        sampler = create_sampler(sampler_type=sampler_type or self.sampler_type, scheduler=self.scheduler,
                                 num_inference_steps=num_inference_steps or self.num_inference_steps,
                                 eta=self.eta if eta is None else eta)
        with torch.no_grad():
            samples = sampler(self.model, (num_samples, self.config['in_channels'], self.config['image_size'],
                                           self.config['image_size']), progress_bar=progress_bar)
        return postprocess(samples)

    def save_samples(self, samples, filepath, nrow=4):
        """sample.py:123-150 without the matplotlib window."""
        save_png(make_grid(samples, nrow=nrow, padding=2), filepath)

    def generate_and_save(self, num_samples=4, output_dir="samples"):
        os.makedirs(output_dir, exist_ok=True)
        samples = self.generate_samples(num_samples=num_samples, progress_bar=False)
        name = f"samples_{self.sampler_type}_steps{self.num_inference_steps}_eta{self.eta}.png"
        self.save_samples(samples, os.path.join(output_dir, name))
        return samples

    def generate_comparison(self, num_samples=16, output_dir="samples"):
        """sample.py:172-221; "deterministic" is not a sampler type (sampler.py:187-192) -> DDIM eta 0."""
        os.makedirs(output_dir, exist_ok=True)
        settings = [("ddim", 1000, 0.0), ("ddim", 50, 0.0), ("ddim", 50, 0.5), ("ancestral", 1000, 1.0)]
        alls = []
        for st, steps, eta in settings:
            s = self.generate_samples(num_samples=num_samples, sampler_type=st, num_inference_steps=steps, eta=eta,
                                      progress_bar=False)
            alls.append(s)
            self.save_samples(s, os.path.join(output_dir, f"comparison_{st}_steps{steps}_eta{eta}.png"), nrow=4)
        self.save_samples(torch.cat(alls, dim=0), os.path.join(output_dir, "sampler_comparison.png"), nrow=num_samples)
        return alls

    def generate_progressive(self, num_samples=4, output_dir="samples", eta=0.0, num_inference_steps=1000, save_every=10):
        """sample.py:223-265: per-step model / scheduler.step calls with snapshots."""
        os.makedirs(output_dir, exist_ok=True)
        shape = (num_samples, self.config['in_channels'], self.config['image_size'], self.config['image_size'])
        samples = torch.randn(shape).to(self.device)
        self.save_samples(samples, os.path.join(output_dir, "progressive_noise.png"), nrow=num_samples)
        self.scheduler.set_timesteps(num_inference_steps=num_inference_steps)
        for i, timestep in enumerate(self.scheduler.timesteps):
            with torch.no_grad():
                eps = self.model(samples, timestep)
                samples = self.scheduler.step(eps, timestep, samples, eta=eta)
            if (i + 1) % save_every == 0:
                self.save_samples(samples, os.path.join(output_dir, f"progressive_step_{i + 1:03d}.png"), nrow=num_samples)
        self.save_samples(samples, os.path.join(output_dir, "progressive_final.png"), nrow=num_samples)
        return samples


def get_default_config():
    """sample.py:268-296."""
    return {'image_size': 32, 'in_channels': 1, 'out_channels': 1, 'model_channels': 64, 'time_emb_dim': 128,
            'num_train_timesteps': 1000, 'beta_start': 0.0001, 'beta_end': 0.02, 'beta_schedule': 'linear',
            'checkpoint_path': None, 'batch_size': 32, 'num_workers': 4, 'data_dir': 'data',
            'results_dir': 'evaluation_results', 'seed': 42}


def load_config(config_path):
    """sample.py:298-321: ``key: value`` text lines over the defaults."""
    config = get_default_config()
    with open(config_path) as f:
        for line in f:
            if ':' not in line:
                continue
            key, value = (s.strip() for s in line.split(':', 1))
            if key not in config:
                continue
            cur = config[key]
            try:
                if isinstance(cur, bool):
                    config[key] = value.lower() == 'true'
                elif isinstance(cur, int):
                    config[key] = int(value)
                elif isinstance(cur, float):
                    config[key] = float(value)
                else:
                    config[key] = None if value == 'None' else value
            except ValueError:
                config[key] = value
    return config


def main(argv=None):
    """sample.py:323-363 with the same flags.  ``--sampler deterministic`` (the reference's default, which its
    own ``create_sampler`` rejects: SURVEY F10) is taken as DDIM with eta 0."""
    ap = argparse.ArgumentParser(description="Generate samples from a trained DDPM (MI355X-native sampler)")
    ap.add_argument("--checkpoint", type=str, required=True)
    ap.add_argument("--config", type=str)
    ap.add_argument("--num-samples", type=int, default=16)
    ap.add_argument("--sampler", type=str, default="deterministic", choices=["ancestral", "deterministic", "ddim"])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--eta", type=float, default=1.0)
    ap.add_argument("--output-dir", type=str, default="samples")
    ap.add_argument("--comparison", action="store_true")
    ap.add_argument("--progressive", action="store_true")
    ap.add_argument("--seed", type=int, default=42)
    args = ap.parse_args(argv)
    config = load_config(args.config) if args.config else get_default_config()
    config["checkpoint_path"] = args.checkpoint
    config["seed"] = args.seed
    stype, eta = ("ddim", 0.0) if args.sampler == "deterministic" else (args.sampler, args.eta)
    drv = DDPMSampler(config, sampler_type=stype, num_inference_steps=args.steps, eta=eta)
    if args.comparison:
        drv.generate_comparison(num_samples=args.num_samples, output_dir=args.output_dir)
    elif args.progressive:
        drv.generate_progressive(num_samples=args.num_samples, output_dir=args.output_dir, eta=eta,
                                 num_inference_steps=args.steps)
    else:
        drv.generate_and_save(num_samples=args.num_samples, output_dir=args.output_dir)


if __name__ == "__main__":
    main()
