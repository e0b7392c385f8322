"""Batch-sharded sampling over the GPUs of one node; no reference counterpart —
the reference is single-device.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Images are fully
independent along the whole reverse chain (GroupNorm is per sample, the time embedding is
identical on every rank), so rank r samples the contiguous rows [r*N/W, (r+1)*N/W) of the
host-drawn x_T with a full weight replica and there is NO communication inside the loop; the
finished shards are exchanged by ONE all-gather.  The full [N,...] noise tensor is drawn once
(same generator state on every rank) and sliced ON THE HOST, because ``torch.randn`` of a smaller
shape does not reproduce a slice of the larger draw; only the rank's own rows are uploaded
(4 KiB per image, not the whole job's x_T on every GPU).
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


def _world_rank(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def shard_rows(full: torch.Tensor, dim: int = 0, group=None) -> torch.Tensor:
    """This rank's contiguous row slice of a tensor every rank holds (x_T: dim 0; per-step noise [n,N,...]: dim 1).
    Slicing happens where `full` lives - hand it a HOST tensor and upload the result."""
    world, rank = _world_rank(group)
    lo, hi = shard_bounds(full.shape[dim], rank, world)
    return full.narrow(dim, lo, hi - lo).contiguous()


_GATHER_BUF = {}


def gather_shards(shard: torch.Tensor, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ONE all-gather of the finished shards ([N/W, ...] on every rank -> [N, ...] on every rank).  The result buffer is
    kept and reused for the same shape (no allocation per pass); pass `out` to own it.  On the fully connected xGMI mesh
    this is a direct push of each shard over its own link; with uint8 shards it moves a quarter of the bytes."""
    world, _ = _world_rank(group)
    shard = shard.contiguous()
    if world == 1:
        return shard
    shape = (shard.shape[0] * world,) + tuple(shard.shape[1:])
    if out is None:
        key = (shape, shard.dtype, shard.device)
        out = _GATHER_BUF.get(key)
        if out is None:
            _GATHER_BUF.clear()  # one live geometry at a time
            out = _GATHER_BUF[key] = torch.empty(shape, dtype=shard.dtype, device=shard.device)
    elif tuple(out.shape) != shape or out.dtype != shard.dtype or out.device != shard.device:
        raise ValueError(f"gather buffer must be {shape} {shard.dtype} on {shard.device}")
    if dist.get_backend(group) == "gloo" and shard.device.type != "cpu":  # one-GPU rehearsal: gloo gathers host tensors
        parts = [torch.empty(shard.shape, dtype=shard.dtype) for _ in range(world)]
        dist.all_gather(parts, shard.cpu(), group=group)
        out.copy_(torch.cat(parts))
    else:
        dist.all_gather_into_tensor(out, shard, group=group)
    return out


def sample_sharded(sample_fn: Callable[[torch.Tensor, Optional[torch.Tensor]], torch.Tensor],
                   x_T_full: torch.Tensor, noises_full: Optional[torch.Tensor] = None,
                   group=None, gather: bool = True, as_uint8: bool = False) -> torch.Tensor:
    """Run ``sample_fn(x_T_shard, noises_shard)`` on this rank's rows and all-gather the results.

    ``x_T_full`` [N,C,H,W] (and ``noises_full`` [n,N,C,H,W] when eta > 0) are the SAME tensors on
    every rank, normally on the host.  Returns the gathered [N,C,H,W] samples (on the device
    ``sample_fn`` returned) or just the local shard when ``gather`` is False.  ``as_uint8``: the
    shards are post-processed to [0,255] uint8 images (sample.py:117-119 + quantisation) BEFORE the
    gather, which then moves a quarter of the bytes."""
    nz = None if noises_full is None else shard_rows(noises_full, 1, group)
    out = sample_fn(shard_rows(x_T_full, 0, group), nz).contiguous()
    if as_uint8:
        from .sample import postprocess
        out = postprocess(out, uint8=True)
    if not gather:
        return out
    return gather_shards(out, group)


def sampler_fn(sampler, model):
    """Adapter: a sampler/model pair of this package as a ``sample_fn``."""
    return lambda x_T, noises: sampler.sample_from(model, x_T, noises)
