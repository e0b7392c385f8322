"""ClassConditioner drop-in (reference: diffusion/conditioning.py:8-55) on a HIP kernel.

The reference defines the module but wires it to nothing; how it enters the UNet
and classifier-free guidance are BUILD-DEFINED (parity unpinned, oracle = oracle/unet_oracle.py
``cfg_eps``): the class embedding is added to the time embedding and guidance combines
eps = eps_u + w (eps_c - eps_u) on the doubled batch.  The module itself is the reference's:
``nn.Embedding(num_classes, E)``, labels in [0, num_classes).  A caller that wants a null class
for guidance constructs it with ONE MORE row (``ClassConditioner(10 + 1)`` for ten digits) and
uses label ``10`` as the null label; nothing here reserves that row.
"""
import torch
import torch.nn as nn

from . import _lib
from .utils import get_device


class ClassConditioner(nn.Module):
    """Same parameters / state_dict keys as the reference: ``class_embedding.weight``,
    ``projection.{0,2}.{weight,bias}``."""

    def __init__(self, num_classes: int, embedding_dim: int = 128):
        super().__init__()
        self.device = get_device()
        self.num_classes = num_classes
        self.embedding_dim = embedding_dim
        self.class_embedding = nn.Embedding(num_classes, embedding_dim)
        self.projection = nn.Sequential(
            nn.Linear(embedding_dim, embedding_dim),
            nn.SiLU(),
            nn.Linear(embedding_dim, embedding_dim),
        )
        self.to(self.device)

    def forward(self, class_labels: torch.LongTensor) -> torch.FloatTensor:
        tbl = self.class_embedding.weight
        if tbl.device.type != "cuda":
            raise RuntimeError("ClassConditioner runs on the GPU only (no CPU path)")
        lab = class_labels.to(tbl.device, torch.int64).reshape(-1).contiguous()
        if lab.numel() and (int(lab.min()) < 0 or int(lab.max()) >= self.num_classes):
            raise IndexError("index out of range in self")  # what nn.Embedding raises
        out = torch.empty((lab.numel(), self.embedding_dim), dtype=torch.float32, device=tbl.device)
        if lab.numel() == 0:  # the reference returns an empty [0, E] tensor
            return out
        p = self.projection
        _lib.check(_lib.lib().dm_class_embed(
            _lib.ptr(tbl.contiguous()), self.num_classes, self.embedding_dim,
            _lib.ptr(p[0].weight.contiguous()), _lib.ptr(p[0].bias), _lib.ptr(p[2].weight.contiguous()),
            _lib.ptr(p[2].bias), _lib.ptr(lab), lab.numel(), _lib.ptr(out),
            torch.cuda.current_stream(tbl.device).cuda_stream))
        return out


def create_conditioner(conditioner_type: str, **kwargs):
    """diffusion/conditioning.py factory; only the class conditioner is on the sampling path."""
    if conditioner_type == "class":
        return ClassConditioner(**kwargs)
    raise ValueError(f"Unknown conditioner type: {conditioner_type}")
