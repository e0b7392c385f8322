"""Batch-sharded sampling over the GPUs of one node; no reference counterpart —
the reference is single-device.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Images are fully
independent along the whole reverse chain (GroupNorm is per sample, the time embedding is
identical on every rank), so rank r samples the contiguous rows [r*N/W, (r+1)*N/W) of the
host-drawn x_T with a full weight replica and there is NO communication inside the loop; the
finished shards are exchanged by ONE all-gather.  The full [N,...] noise tensor is drawn once
(same generator state on every rank) and sliced, because ``torch.randn`` of a smaller shape
does not reproduce a slice of the larger draw.
"""
from typing import Callable, Optional

import torch
import torch.distributed as dist


def shard_bounds(n_total: int, rank: int, world: int):
    """Contiguous, equal row shards; N must divide evenly (pad the batch otherwise)."""
    if n_total % world:
        raise ValueError(f"batch {n_total} is not divisible by world size {world}")
    per = n_total // world
    return rank * per, (rank + 1) * per


def sample_sharded(sample_fn: Callable[[torch.Tensor, Optional[torch.Tensor]], torch.Tensor],
                   x_T_full: torch.Tensor, noises_full: Optional[torch.Tensor] = None,
                   group=None, gather: bool = True) -> torch.Tensor:
    """Run ``sample_fn(x_T_shard, noises_shard)`` on this rank's rows and all-gather the results.

    ``x_T_full`` [N,C,H,W] (and ``noises_full`` [n,N,C,H,W] when eta > 0) are the SAME tensors on
    every rank.  Returns the gathered [N,C,H,W] samples (on the device ``sample_fn`` returned)
    or just the local shard when ``gather`` is False.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(x_T_full.shape[0], rank, world)
    nz = None if noises_full is None else noises_full[:, lo:hi].contiguous()
    out = sample_fn(x_T_full[lo:hi].contiguous(), nz).contiguous()
    if world == 1 or not gather:
        return out
    full = torch.empty((x_T_full.shape[0],) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
    if dist.get_backend(group) == "gloo":  # CPU tests / one-GPU rehearsal: gloo gathers host tensors
        parts = [torch.empty(out.shape, dtype=out.dtype) for _ in range(world)]
        dist.all_gather(parts, out.cpu(), group=group)
        full.copy_(torch.cat(parts))
    else:
        dist.all_gather_into_tensor(full, out, group=group)
    return full


def sampler_fn(sampler, model):
    """Adapter: a sampler/model pair of this package as a ``sample_fn``."""
    return lambda x_T, noises: sampler.sample_from(model, x_T, noises)
