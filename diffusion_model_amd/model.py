"""UNetMini drop-in (reference: diffusion/model.py:67-151) running on hand-written gfx950 kernels.

Same constructor signature, attributes, sub-module names and ``state_dict`` keys as the
reference, so its checkpoints load unchanged and ``set_seed(s); UNetMini(...)`` produces the
same initial weights.  The ``nn`` sub-modules are parameter containers only: ``forward``
never calls them — it hands device pointers to libdm_sampler.so (include/dm_sampler.h).
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from . import _lib
from .utils import get_device


class SinusoidalPositionEmbeddings(nn.Module):
    """Parameter-less placeholder keeping ``time_mlp.{1,3}`` key numbering
    (diffusion/model.py:8-30).  The embedding itself is computed by the time_embed kernel."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.device = get_device()

    def frequencies(self):
        # diffusion/model.py:19-21, evaluated with the same torch-CPU ops (bit-identical table)
        half = self.dim // 2
        return torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))

    def forward(self, time):
        raise RuntimeError("SinusoidalPositionEmbeddings is evaluated inside UNetMini.forward "
                           "(time_embed kernel); call the model, not the sub-module")


class Block(nn.Module):
    """Parameter container with the reference's layout (diffusion/model.py:33-50)."""

    def __init__(self, in_ch, out_ch, time_emb_dim, up=False):
        super().__init__()
        self.time_mlp = nn.Linear(time_emb_dim, out_ch)
        if up:
            self.conv1 = nn.Conv2d(2 * in_ch, out_ch, 3, padding=1)
            self.transform = nn.ConvTranspose2d(out_ch, out_ch, 4, 2, 1)
        else:
            self.conv1 = nn.Conv2d(in_ch, out_ch, 3, padding=1)
            self.transform = nn.Conv2d(out_ch, out_ch, 4, 2, 1)
        self.conv2 = nn.Conv2d(out_ch, out_ch, 3, padding=1)
        self.bnorm1 = nn.GroupNorm(8, out_ch)
        self.bnorm2 = nn.GroupNorm(8, out_ch)
        self.relu = nn.SiLU()

    def forward(self, x, t):
        raise RuntimeError("Block is fused into UNetMini.forward (conv_mfma kernels); "
                           "call the model, not the sub-module")


class AttnBlock(nn.Module):
    """Parameter container of the BUILD-DEFINED self-attention block (no reference counterpart,
    model.py:70): GroupNorm(8) -> 1x1 qkv -> softmax(QK^T/sqrt(C))V -> 1x1 proj -> residual."""

    def __init__(self, ch):
        super().__init__()
        self.norm = nn.GroupNorm(8, ch)
        self.qkv = nn.Conv2d(ch, 3 * ch, 1)
        self.proj = nn.Conv2d(ch, ch, 1)

    def forward(self, x):
        raise RuntimeError("AttnBlock is evaluated inside UNetMini.forward")


class UNetMini(nn.Module):
    """diffusion/model.py:67-151.  ``forward(x[B,Cin,H,W], timesteps)`` -> eps[B,Cout,H,W];
    ``timesteps`` is a 0-dim int64 tensor (sampling) or a [B] tensor (training style)."""

    def __init__(self, image_size=28, in_channels=1, out_channels=1, model_channels=64,
                 time_emb_dim=128, attention=False, precision="fp32"):
        super().__init__()
        self.attention = bool(attention)  # build-defined extension; False = the reference model
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"Unknown precision: {precision}")
        # "bf16": bf16 activations / conv weights, fp32 accumulate (BASELINE configs[3]); the
        # parameters themselves stay fp32 (state_dict unchanged)
        self.precision = precision
        self.device = get_device()
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.time_emb_dim = time_emb_dim
        self.out_channels = out_channels
        self.time_mlp = nn.Sequential(
            SinusoidalPositionEmbeddings(time_emb_dim),
            nn.Linear(time_emb_dim, time_emb_dim),
            nn.SiLU(),
            nn.Linear(time_emb_dim, time_emb_dim),
        )
        self.init_conv = nn.Conv2d(in_channels, model_channels, kernel_size=3, padding=1)
        self.down1 = Block(model_channels, model_channels, time_emb_dim, up=False)
        self.down2 = Block(model_channels, model_channels, time_emb_dim, up=False)
        self.down3 = Block(model_channels, model_channels, time_emb_dim, up=False)
        self.bottleneck = nn.Sequential(
            nn.Conv2d(model_channels, model_channels, 3, padding=1),
            nn.GroupNorm(8, model_channels),
            nn.SiLU(),
            nn.Conv2d(model_channels, model_channels, 3, padding=1),
        )
        self.up1 = Block(model_channels, model_channels, time_emb_dim, up=True)
        self.up2 = Block(model_channels, model_channels, time_emb_dim, up=True)
        self.up3 = Block(model_channels, model_channels, time_emb_dim, up=True)
        self.final_conv = nn.Sequential(
            nn.GroupNorm(8, model_channels),
            nn.SiLU(),
            nn.Conv2d(model_channels, out_channels, 1),
        )
        if self.attention:  # registered last: the reference's 88 tensors keep their order
            self.attn_down = AttnBlock(model_channels)
            self.attn_up = AttnBlock(model_channels)
        self.to(self.device)
        self._handle = None
        self._packed = None
        self._pack_key = None
        self._ws = None

    # ------------------------------------------------------------------ native plumbing
    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None:
            try:
                _lib.lib().dm_unet_destroy(h)
            except Exception:
                pass

    def _native(self):
        if self._handle is None:
            h = C.c_void_p()
            _lib.check(_lib.lib().dm_unet_create_ex(C.byref(h), self.in_channels, self.out_channels,
                                                    self.model_channels, self.time_emb_dim, int(self.attention)))
            _lib.check(_lib.lib().dm_unet_set_precision(h, 1 if self.precision == "bf16" else 0))
            self._handle = h
        return self._handle

    def _params_in_order(self):
        ps = list(self.parameters())
        want = 100 if self.attention else 88
        if len(ps) != want:
            raise RuntimeError(f"expected {want} parameter tensors, found {len(ps)}")
        return ps

    def packed_weights(self):
        """Packed weight blob; re-packed lazily whenever a parameter's storage or version
        changes (EMA ``param.data = ...`` swaps, load_state_dict, optimizer steps)."""
        ps = self._params_in_order()
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if key != self._pack_key:
            dev = ps[0].device
            if dev.type != "cuda":
                raise RuntimeError("UNetMini needs its parameters on the GPU (cuda/ROCm device); "
                                   f"they are on {dev}. There is no CPU path.")
            L, h = _lib.lib(), self._native()
            for p in ps:
                if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                    raise RuntimeError("parameters must be contiguous fp32 tensors on one device")
            if self._packed is None or self._packed.device != dev:
                self._packed = torch.empty(L.dm_unet_packed_floats(h), dtype=torch.float32, device=dev)
            arr = (C.c_void_p * len(ps))(*[p.data_ptr() for p in ps])
            freqs = self.time_mlp[0].frequencies().to(dev)
            with torch.cuda.device(dev):
                _lib.check(L.dm_unet_pack_weights(h, arr, len(ps), _lib.ptr(freqs), _lib.ptr(self._packed),
                                                  torch.cuda.current_stream(dev).cuda_stream))
            self._pack_key = key
            self._freqs_keepalive = freqs
        return self._packed

    def workspace(self, B, H, W, nsteps=1):
        return self.workspace_bytes(_lib.lib().dm_unet_workspace_bytes(self._native(), B, H, W, nsteps))

    def workspace_bytes(self, need):
        """Grow-only device workspace of at least ``need`` bytes (the library never allocates)."""
        dev = next(self.parameters()).device
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._ws

    # ------------------------------------------------------------------ forward
    def forward(self, x, timesteps, cond_emb=None):
        """diffusion/model.py:120-151.  ``cond_emb`` ([1|B, time_emb_dim], added to the time
        embedding) is a build-defined conditioning hook with no reference counterpart."""
        if x.dim() != 4:
            raise RuntimeError(f"expected a 4-D input [B,C,H,W], got {tuple(x.shape)}")
        B, Cin, H, W = x.shape
        if Cin != self.in_channels:
            raise RuntimeError(f"expected {self.in_channels} input channels, got {Cin}")
        packed = self.packed_weights()
        dev = packed.device
        if x.device != dev or x.dtype != torch.float32:
            raise RuntimeError(f"input must be an fp32 tensor on {dev} (got {x.dtype} on {x.device})")
        x = x.contiguous()
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor(timesteps)
        t = timesteps.to(device=dev, dtype=torch.int64).reshape(-1).contiguous()
        nt = t.numel()
        cond = None
        if cond_emb is not None:
            cond = cond_emb.to(device=dev, dtype=torch.float32).contiguous()
            if cond.shape != (nt, self.time_emb_dim):
                if cond.shape == (B, self.time_emb_dim) and nt == 1:
                    t, nt = t.expand(B).contiguous(), B
                else:
                    raise RuntimeError(f"cond_emb must be [{nt},{self.time_emb_dim}]")
        eps = torch.empty((B, self.out_channels, H, W), dtype=torch.float32, device=dev)
        if B == 0:  # the reference's stock ops return an empty [0, Cout, H, W] tensor
            return eps
        ws = self.workspace(B, H, W)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().dm_unet_forward(
                self._native(), _lib.ptr(packed), _lib.ptr(x), _lib.ptr(t), nt, _lib.ptr(cond), _lib.ptr(eps),
                B, H, W, _lib.ptr(ws), ws.numel(), torch.cuda.current_stream(dev).cuda_stream))
        return eps


if __name__ == "__main__":
    m = UNetMini(image_size=32)
    print(sum(p.numel() for p in m.parameters()))
