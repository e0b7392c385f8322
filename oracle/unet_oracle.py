"""CPU oracle for the reverse-diffusion sampling hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``diffusion_model_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg do, and only as the checker.

It is a build-owned restatement (stock torch-CPU ops, functional style over a plain
``state_dict``) of the reference algorithm, function by function:

* ``sinusoidal_embedding``  <- diffusion/model.py:18-30
* ``block_forward``         <- diffusion/model.py:52-64
* ``unet_forward``          <- diffusion/model.py:120-151
* ``make_schedule``         <- diffusion/schedulers.py:30-63 (+ cosine :65-72)
* ``make_timesteps``        <- diffusion/schedulers.py:140-147
* ``sampler_step``          <- diffusion/schedulers.py:113-131 == diffusion/sampler.py:117-134
  (``sampler_step_np``: the same in strict IEEE fp32 numpy)
* ``sample``                <- diffusion/sampler.py:45-76 and :136-172

Parity pinning: the reference holds no golden vectors of its own, so this
oracle is pinned by fixtures generated in the build container by importing the real
reference (``oracle/gen_golden.py`` -> ``tests/golden/``); ``tests/test_oracle_golden.py``
checks the restatement against them.  Build-defined extensions (class conditioning with
classifier-free guidance, attention) have no reference counterpart: "parity unpinned"
for those functions, stated at each.
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- model
def sinusoidal_embedding(time: Tensor, dim: int) -> Tensor:
    """model.py:18-30.  ``time`` is 0-dim or [B] int64; returns [1|B, dim] fp32,
    sin block first then cos; note the (half-1) denominator."""
    half = dim // 2
    scale = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half) * -scale)
    if time.dim() == 0:
        time = time.unsqueeze(0)
    arg = time[:, None] * freqs[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def time_mlp(sd: Dict[str, Tensor], timesteps: Tensor, dim: int) -> Tensor:
    """model.py:82-87,132: sinusoid -> Linear -> SiLU -> Linear."""
    e = sinusoidal_embedding(timesteps, dim)
    e = F.linear(e, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    e = F.silu(e)
    return F.linear(e, sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])


def block_forward(sd: Dict[str, Tensor], p: str, x: Tensor, t: Tensor, up: bool) -> Tensor:
    """model.py:52-64 for the block whose parameters are prefixed ``p``."""
    h = F.conv2d(x, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, 8, sd[p + "bnorm1.weight"], sd[p + "bnorm1.bias"], eps=1e-5))
    te = F.silu(F.linear(t, sd[p + "time_mlp.weight"], sd[p + "time_mlp.bias"]))
    h = h + te[(...,) + (None,) * 2]
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    h = F.silu(F.group_norm(h, 8, sd[p + "bnorm2.weight"], sd[p + "bnorm2.bias"], eps=1e-5))
    if up:
        return F.conv_transpose2d(h, sd[p + "transform.weight"], sd[p + "transform.bias"],
                                  stride=2, padding=1)
    return F.conv2d(h, sd[p + "transform.weight"], sd[p + "transform.bias"], stride=2, padding=1)


def attention_block(sd: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    """BUILD-DEFINED (no reference counterpart; parity unpinned): x + proj(softmax(QK^T/sqrt(C))V),
    [Q,K,V] = qkv_1x1(GroupNorm8(x)), single head over the H*W tokens."""
    B, C, H, W = x.shape
    h = F.group_norm(x, 8, sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-5)
    q, k, v = F.conv2d(h, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(B, 3, C, H * W).unbind(1)
    s = torch.einsum("bct,bcs->bts", q, k) / math.sqrt(C)
    o = torch.einsum("bts,bcs->bct", s.softmax(-1), v).reshape(B, C, H, W)
    return x + F.conv2d(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def unet_forward(sd: Dict[str, Tensor], x: Tensor, timesteps: Tensor,
                 cond_emb: Optional[Tensor] = None, taps: Optional[dict] = None) -> Tensor:
    """model.py:120-151.  ``cond_emb`` ([B,E], added to the time embedding) is the
    build-defined class-conditioning hook; parity unpinned when used.
    ``taps`` (optional dict) receives intermediate activations for per-kernel tests."""
    E = sd["time_mlp.1.weight"].shape[0]
    t = time_mlp(sd, timesteps, E)
    if cond_emb is not None:
        t = t + cond_emb
    h = F.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=1)
    if taps is not None:
        taps["temb"] = t
        taps["init"] = h
    d1 = block_forward(sd, "down1.", h, t, False)
    d2 = block_forward(sd, "down2.", d1, t, False)
    attn = "attn_down.qkv.weight" in sd  # build-defined extension
    if attn:
        d2 = attention_block(sd, "attn_down.", d2)
    d3 = block_forward(sd, "down3.", d2, t, False)
    b = F.conv2d(d3, sd["bottleneck.0.weight"], sd["bottleneck.0.bias"], padding=1)
    b = F.silu(F.group_norm(b, 8, sd["bottleneck.1.weight"], sd["bottleneck.1.bias"], eps=1e-5))
    b = F.conv2d(b, sd["bottleneck.3.weight"], sd["bottleneck.3.bias"], padding=1)
    u1 = block_forward(sd, "up1.", torch.cat([b, d3], dim=1), t, True)
    if attn:
        u1 = attention_block(sd, "attn_up.", u1)
    u2 = block_forward(sd, "up2.", torch.cat([u1, d2], dim=1), t, True)
    u3 = block_forward(sd, "up3.", torch.cat([u2, d1], dim=1), t, True)
    if taps is not None:
        taps.update(down1=d1, down2=d2, down3=d3, bott=b, up1=u1, up2=u2, up3=u3)
    o = F.silu(F.group_norm(u3, 8, sd["final_conv.0.weight"], sd["final_conv.0.bias"], eps=1e-5))
    return F.conv2d(o, sd["final_conv.2.weight"], sd["final_conv.2.bias"])


# ----------------------------------------------------------------------- schedules
def make_schedule(num_train_timesteps: int = 1000, beta_start: float = 1e-4,
                  beta_end: float = 0.02, beta_schedule: str = "linear") -> Dict[str, Tensor]:
    """schedulers.py:30-63.  Only the tables the sampling path reads."""
    if beta_schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps)
    elif beta_schedule == "cosine":  # schedulers.py:65-72
        steps = num_train_timesteps + 1
        x = torch.linspace(0, num_train_timesteps, steps)
        acp = torch.cos(((x / num_train_timesteps) + 0.008) / 1.008 * torch.pi * 0.5) ** 2
        acp = acp / acp[0]
        betas = torch.clip(1 - (acp[1:] / acp[:-1]), 0.0001, 0.9999)
    else:
        raise ValueError(f"Unknown beta schedule: {beta_schedule}")
    alphas = 1.0 - betas
    acp = torch.cumprod(alphas, dim=0)
    acp_prev = F.pad(acp[:-1], (1, 0), value=1.0)
    pv = betas * (1.0 - acp_prev) / (1.0 - acp)
    return dict(betas=betas, alphas=alphas, alphas_cumprod=acp,
                alphas_cumprod_prev=acp_prev, posterior_variance=pv)


def make_timesteps(num_train_timesteps: int, num_inference_steps: int) -> Tensor:
    """schedulers.py:146-147: ascending, integer stride."""
    return torch.arange(0, num_train_timesteps, num_train_timesteps // num_inference_steps)


def sampler_step(sched: Dict[str, Tensor], eps: Tensor, t: Tensor, x: Tensor,
                 eta: float = 0.0, noise: Optional[Tensor] = None) -> Tensor:
    """schedulers.py:113-131 (== sampler.py:117-134).  ``noise`` replaces the
    ``randn_like`` draw when given (same values the global generator would produce)."""
    a = sched["alphas_cumprod"][t].view(-1, 1, 1, 1)
    ap = sched["alphas_cumprod_prev"][t].view(-1, 1, 1, 1)
    x0 = (x - torch.sqrt(1 - a) * eps) / torch.sqrt(a)
    mean = torch.sqrt(ap) * x0 + torch.sqrt(1 - ap) * eps
    if eta > 0:
        if noise is None:
            noise = torch.randn_like(x)
        std = torch.sqrt(sched["posterior_variance"][t].view(-1, 1, 1, 1)) * eta
        mean = mean + std * noise
    return mean


def sampler_step_np(sched: Dict[str, Tensor], eps: Tensor, t: int, x: Tensor,
                    eta: float = 0.0, noise: Optional[Tensor] = None) -> Tensor:
    """The same update in strict IEEE-754 fp32 with numpy (one correctly rounded operation per
    step, reference operation order).  ``torch.sqrt`` on CPU goes through MKL VML, whose last
    bit differs between host CPU vendors (seen: Intel build host vs the AMD EPYC GPU-box host,
    sqrt(1-a_1) one ulp apart), so bit-exactness of the HIP kernel is asserted against this
    restatement and closeness (<= 1e-6 relative) against ``sampler_step``."""
    import numpy as np
    f = np.float32
    a = sched["alphas_cumprod"][t].numpy().astype(f)
    ap = sched["alphas_cumprod_prev"][t].numpy().astype(f)
    c0, c1, c2, c3 = np.sqrt(f(1) - a), np.sqrt(a), np.sqrt(ap), np.sqrt(f(1) - ap)
    xn, en = x.numpy(), eps.numpy()
    out = c2 * ((xn - c0 * en) / c1) + c3 * en
    if eta > 0:
        c4 = np.sqrt(sched["posterior_variance"][t].numpy().astype(f)) * f(eta)
        out = out + c4 * noise.numpy()
    return torch.from_numpy(out.astype(f))


@torch.no_grad()
def sample(sd: Dict[str, Tensor], sched: Dict[str, Tensor], shape, num_inference_steps: int,
           eta: float, x_T: Optional[Tensor] = None, noises: Optional[Tensor] = None,
           record: Optional[list] = None) -> Tensor:
    """sampler.py:45-76 / :136-172.  Draw order as the reference: x_T first, then one
    randn_like per step AFTER the model call when eta>0."""
    T = sched["betas"].numel()
    x = torch.randn(shape) if x_T is None else x_T.clone()
    for i, t in enumerate(make_timesteps(T, num_inference_steps)):
        eps = unet_forward(sd, x, t)
        nz = None if noises is None else noises[i]
        xn = sampler_step(sched, eps, t, x, eta, nz)
        if record is not None:
            record.append((int(t), x, eps, xn))
        x = xn
    return x


# ------------------------------------------------- build-defined extensions (unpinned)
def class_conditioner(cd: Dict[str, Tensor], labels: Tensor) -> Tensor:
    """diffusion/conditioning.py:39-55 (Embedding -> Linear -> SiLU -> Linear)."""
    e = cd["class_embedding.weight"][labels]
    e = F.silu(F.linear(e, cd["projection.0.weight"], cd["projection.0.bias"]))
    return F.linear(e, cd["projection.2.weight"], cd["projection.2.bias"])


def cfg_eps(sd, cd, x: Tensor, t: Tensor, labels: Tensor, null_label: int, w: float) -> Tensor:
    """Build-defined classifier-free guidance (no reference counterpart; parity unpinned):
    eps = eps_u + w (eps_c - eps_u) on the doubled batch [x;x], labels [y;null]."""
    B = x.shape[0]
    lab = torch.cat([labels, torch.full_like(labels, null_label)])
    e = unet_forward(sd, torch.cat([x, x]), t, cond_emb=class_conditioner(cd, lab))
    ec, eu = e[:B], e[B:]
    return eu + w * (ec - eu)


@torch.no_grad()
def cfg_sample(sd, cd, sched, x_T: Tensor, labels: Tensor, null_label: int, w: float,
               num_inference_steps: int) -> Tensor:
    """Build-defined guided DDIM loop (eta = 0) used as the oracle of dm_sample's CFG mode."""
    T = sched["betas"].numel()
    x = x_T.clone()
    for t in make_timesteps(T, num_inference_steps):
        e = cfg_eps(sd, cd, x, t, labels, null_label, w)
        x = sampler_step(sched, e, t, x, 0.0)
    return x


def psnr(a: Tensor, b: Tensor, max_val: float) -> Tensor:
    """diffusion/eval.py:282-283, per image: 20 log10(max_val / sqrt(mse))."""
    mse = ((a - b) ** 2).flatten(1).mean(1)
    return 20 * torch.log10(max_val / torch.sqrt(mse))
