"""CPU-side pieces of the sampling driver (sample.py's grid layout, PNG writer, config parser)."""
import os
import zlib

import torch

from diffusion_model_amd import sample as S


def test_make_grid_layout():
    imgs = torch.arange(6 * 1 * 4 * 4, dtype=torch.float32).reshape(6, 1, 4, 4) / 100
    g = S.make_grid(imgs, nrow=4, padding=2)
    assert g.shape == (3, 2 * 6 + 2, 4 * 6 + 2)           # torchvision: rows*(H+p)+p, cols*(W+p)+p
    assert torch.equal(g[0, 2:6, 2:6], imgs[0, 0]) and torch.equal(g[2, 8:12, 2:6], imgs[4, 0])
    assert g[:, :2].abs().sum() == 0 and g[:, 8:12, 14:18].abs().sum() == 0  # padding and empty cell


def test_png_roundtrip(tmp_path):
    g = torch.rand(3, 10, 12)
    p = os.path.join(tmp_path, "a.png")
    S.save_png(g, p)
    b = open(p, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n" and b[12:16] == b"IHDR"
    w, h = int.from_bytes(b[16:20], "big"), int.from_bytes(b[20:24], "big")
    assert (w, h) == (12, 10)
    i = b.index(b"IDAT")
    n = int.from_bytes(b[i - 4:i], "big")
    raw = zlib.decompress(b[i + 4:i + 4 + n])
    assert len(raw) == h * (1 + 3 * w)
    px = torch.frombuffer(bytearray(raw), dtype=torch.uint8).reshape(h, 1 + 3 * w)[:, 1:].reshape(h, w, 3)
    assert torch.equal(px.permute(2, 0, 1), g.mul(255).add(0.5).clamp(0, 255).to(torch.uint8))


def test_text_config_parser(tmp_path):
    p = os.path.join(tmp_path, "cfg.txt")
    open(p, "w").write("image_size: 64\nbeta_end: 0.03\nbeta_schedule: cosine\ncheckpoint_path: None\nunknown: 1\n")
    c = S.load_config(p)
    assert c["image_size"] == 64 and c["beta_end"] == 0.03 and c["beta_schedule"] == "cosine"
    assert c["checkpoint_path"] is None and "unknown" not in c and c["seed"] == 42


def test_chunked_noise_draws_equal_one_draw():
    """AncestralSampler.__call__ draws the eta > 0 noise one 50-step chunk at a time (into pinned staging buffers, while
    the GPU works on the previous chunk).  torch's CPU normal_ fills contiguous tensors in blocks of 16 values, so draws
    whose sizes are multiples of 16 concatenate to exactly the one big draw - and to the reference's per-step
    `randn_like` draws (schedulers.py:125-126).  Pinned here, because parity of __call__ rests on it."""
    shape = (3, 1, 8, 8)   # 192 values per step
    n, S0 = 130, 50
    torch.manual_seed(9)
    x_a = torch.randn(shape)
    one = torch.randn((n,) + shape)
    torch.manual_seed(9)
    x_b = torch.randn(shape)
    ring = [torch.empty((S0,) + shape) for _ in range(3)]
    parts = []
    for k, first in enumerate(range(0, n, S0)):
        cnt = min(S0, n - first)
        buf = ring[k % 3][:cnt]
        torch.randn(buf.shape, out=buf)
        parts.append(buf.clone())
    torch.manual_seed(9)
    x_c = torch.randn(shape)
    per_step = torch.stack([torch.randn(shape) for _ in range(n)])
    assert torch.equal(x_a, x_b) and torch.equal(x_a, x_c)
    assert torch.equal(one, torch.cat(parts)) and torch.equal(one, per_step)
