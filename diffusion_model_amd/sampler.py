"""AncestralSampler / DDIMSampler / create_sampler drop-ins (reference: diffusion/sampler.py).

``__call__`` keeps the reference's behaviour (ascending timesteps, a_{t-1} target, unused
``prev_timestep`` — SURVEY F5) and its noise draw order (x_T, then one draw per step after the
model call when eta > 0 — SURVEY F9), with the draws taken from the seeded CPU generator so a
run reproduces the reference CPU sampler.  When the model is this package's UNetMini the whole
loop runs inside libdm_sampler.so (dm_sample: time-embedding table, per-step UNet + fused
update, captured in a hipGraph); any other callable model is driven step by step.
"""
from typing import Tuple

import torch
import tqdm

from . import _lib
from .model import UNetMini


class _SamplerBase:
    _desc = "Sampling"

    def _init(self, scheduler, num_inference_steps, eta):
        self.scheduler = scheduler
        self.num_inference_steps = num_inference_steps
        self.eta = eta
        self.scheduler.set_timesteps(num_inference_steps)  # mutates the caller's scheduler, as the reference
        self.device = self.scheduler.device
        self.use_graph = True

    def _draw(self, shape, n):
        """Host draws in the reference's order."""
        x_T = torch.randn(shape)
        noises = torch.randn((n,) + tuple(shape)) if self.eta > 0 else None
        return x_T, noises

    def sample_from(self, model, x_T, noises=None, progress_bar=False, cond_emb=None, null_emb=None,
                    guidance_scale=1.0):
        """Run the loop from given x_T (and per-step noises [n,*shape] when eta > 0).

        Build-defined classifier-free guidance (UNetMini only): ``cond_emb`` [B,E] and
        ``null_emb`` [B,E] or [1,E] are added to the time embedding of the conditional /
        unconditional halves of a doubled batch; eps = eps_u + guidance_scale*(eps_c - eps_u)."""
        timesteps = self.scheduler.timesteps
        n = timesteps.numel()
        if self.eta > 0 and noises is None:
            raise RuntimeError("eta > 0 needs per-step noise")
        if isinstance(model, UNetMini):
            cond = None
            if cond_emb is not None:
                B = x_T.shape[0]
                if null_emb is None:
                    raise RuntimeError("guidance needs null_emb (embedding of the null class)")
                cond = torch.cat([cond_emb.reshape(B, -1), null_emb.reshape(-1, cond_emb.shape[-1]).expand(B, -1)])
            return self._fused(model, x_T, noises if self.eta > 0 else None, timesteps, cond, guidance_scale,
                               progress_bar=progress_bar)
        if cond_emb is not None:
            raise RuntimeError("guidance is implemented for this package's UNetMini only")
        x = x_T.to(self.device)
        it = tqdm.tqdm(timesteps, desc=self._desc) if progress_bar else timesteps
        for i, t in enumerate(it):
            with torch.no_grad():
                eps = model(x, t)
            nz = noises[i] if self.eta > 0 else None
            x = self.scheduler.step(eps, t, x, eta=self.eta, noise=nz)
        return x

    def _fused(self, model, x_T, noises, timesteps, cond=None, guidance=1.0, draw_chunks=False, progress_bar=False):
        """The whole loop inside libdm_sampler.so (dm_sample_chunked).  ``draw_chunks``: the per-step noise is drawn HERE,
        from torch's global CPU generator, one graph chunk (<= 50 steps) at a time inside the library's chunk callback -
        the draw of chunk k+1 overlaps the GPU's work on chunk k, and the values are those of one ``torch.randn((n,) + shape)``
        (tests/test_sample_driver_cpu.py pins that).  ``progress_bar``: a tqdm bar advanced by the chunk length as chunks
        COMPLETE on the GPU (the reference ticks per step, sampler.py:65-66; a captured graph cannot)."""
        dev = model.packed_weights().device
        B, Cin, H, W = x_T.shape
        n = timesteps.numel()
        x = x_T.to(dev, torch.float32).contiguous().clone()
        if B == 0:  # the reference loops over an empty batch and returns it
            return x
        # device copies of the timestep list and the per-step coefficient table, kept across calls (they depend on
        # the scheduler tables, the timesteps and eta only): a pageable H2D copy is a host-side sync point per pass
        src = getattr(self, "_dev_tables_src", None)  # (identity + version of the source tensors: no D2H read-back)
        if (src is None or src[0] is not timesteps or src[1] != timesteps._version or src[2] != (dev, float(self.eta))
                or src[3] is not self.scheduler.alphas_cumprod):
            self._dev_tables = (timesteps.to(dev, torch.int64).contiguous(),
                                self.scheduler.step_coefs(timesteps, self.eta).to(dev))
            self._dev_tables_src = (timesteps, timesteps._version, (dev, float(self.eta)), self.scheduler.alphas_cumprod)
        ts_dev, coefs = self._dev_tables
        # per-step noise [n,B,C,H,W]: handed over where it lives.  A host tensor is staged by the library one
        # 50-step graph chunk at a time (ancestral-1000 at B=1024 is 4 GB of noise: never resident on the device)
        nz = None
        if noises is not None:
            nz = noises.to(torch.float32).contiguous()
            if nz.device.type == "cuda" and nz.device != dev:
                nz = nz.to(dev)
            if nz.shape != (n,) + tuple(x.shape):
                raise RuntimeError(f"noises must be [{n},{B},{Cin},{H},{W}], got {tuple(nz.shape)}")
        if cond is not None:
            cond = cond.to(dev, torch.float32).contiguous()
            if cond.shape != (2 * B, model.time_emb_dim):
                raise RuntimeError(f"cond must be [{2 * B},{model.time_emb_dim}]")
        with_noise = nz is not None or draw_chunks
        L = _lib.lib()
        need = L.dm_sample_workspace_bytes(model._native(), B, H, W, n, 0 if cond is None else 1, 1 if with_noise else 0)
        ws = model.workspace_bytes(need)
        packed = model.packed_weights()
        cur = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        side.wait_stream(cur)

        # ---- chunk callback: progress ticks and (draw_chunks) the noise of one chunk, drawn into a ring of pinned buffers
        S0 = min(n, CHUNK_STEPS)
        ring = self._noise_ring(S0, x.shape) if draw_chunks and nz is None else None
        bar = tqdm.tqdm(total=n, desc=self._desc) if progress_bar else None
        st = {"k": 0, "done": 0, "err": None, "ev": {}}

        def on_chunk(_user, first, count):
            try:
                k = st["k"]
                st["k"] = k + 1
                if ring is not None or bar is not None:
                    # everything before this chunk is enqueued: E_k completes when chunk k-1 has.  Buffer k % 3 was last
                    # read by the copy of chunk k-3, which E_(k-2) covers; two chunks stay queued behind the wait.
                    ev = torch.cuda.Event()
                    ev.record(side)
                    st["ev"][k] = ev
                    old = st["ev"].pop(k - 2, None)
                    if old is not None:
                        old.synchronize()
                        if bar is not None:
                            bar.update((k - 2) * S0 - st["done"])
                            st["done"] = (k - 2) * S0
                if ring is None:
                    return None
                buf = ring[k % len(ring)][:count]
                torch.randn(buf.shape, out=buf)  # global CPU generator, loop order (schedulers.py:125-126)
                return buf.data_ptr()
            except BaseException as e:  # never unwind through the C frames
                st["err"] = e
                return None

        cb = _lib.CHUNK_FN(on_chunk) if (ring is not None or bar is not None) else None
        with torch.cuda.device(dev), torch.cuda.stream(side):
            rc = L.dm_sample_chunked(
                model._native(), _lib.ptr(packed), _lib.ptr(x), B, H, W, _lib.ptr(ts_dev), _lib.ptr(coefs), n,
                _lib.ptr(nz), 1 if with_noise else 0, cb, None, _lib.ptr(cond), float(guidance), _lib.ptr(ws), ws.numel(),
                1 if self.use_graph else 0, side.cuda_stream)
            if st["err"] is not None:
                side.synchronize()
                raise st["err"]
            _lib.check(rc)
            if (nz is not None and nz.device.type != "cuda") or ring is not None or bar is not None:
                side.synchronize()  # the library reads host noise chunk by chunk on this stream; the bar ends at n
        if bar is not None:
            bar.update(n - st["done"])
            bar.close()
        cur.wait_stream(side)
        for t_ in (x, ts_dev, coefs, nz, ws, packed, cond):
            if t_ is not None and t_.device.type == "cuda":
                t_.record_stream(side)
        return x

    def _noise_ring(self, S0, shape):
        """Three pinned [S0, *shape] staging buffers for the chunk-wise noise draw, kept across calls."""
        key = (S0, tuple(shape))
        if getattr(self, "_ring_key", None) != key:
            self._ring = [torch.empty((S0,) + tuple(shape), dtype=torch.float32).pin_memory() for _ in range(3)]
            self._ring_key = key
        return self._ring

    def __call__(self, model, shape: Tuple[int, ...], progress_bar: bool = True):
        """diffusion/sampler.py:45-76 / 136-172.  With this package's UNetMini the loop runs inside the library, one
        graph launch per <= 50 steps: the bar advances per chunk and the eta > 0 noise is drawn chunk by chunk while
        the GPU works on the previous chunk (same values, same generator order as the reference's per-step draws)."""
        n = self.scheduler.timesteps.numel()
        numel = 1
        for v in shape[1:]:
            numel *= int(v)
        if isinstance(model, UNetMini) and len(shape) == 4 and (int(shape[0]) * numel) % 16 == 0 and shape[0] > 0:
            # (torch's CPU normal_ fills in blocks of 16 values: draws of a multiple of 16 values concatenate exactly)
            x_T = torch.randn(shape)
            return self._fused(model, x_T, None, self.scheduler.timesteps, draw_chunks=self.eta > 0, progress_bar=progress_bar)
        x_T, noises = self._draw(shape, n)
        return self.sample_from(model, x_T, noises, progress_bar)


CHUNK_STEPS = 50  # steps per captured graph / noise staging chunk (dm_api.hip: S0)
_SIDE = {}


def _side_stream(dev):
    """hipGraph capture is illegal on the legacy default stream: the loop runs on a side stream - one of the library's own
    per device (_lib.own_stream), not one of torch's 32 pooled streams, which another component's torch.cuda.Stream() may
    alias: a launch into it from another thread during the capture would invalidate the graph."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _SIDE:
        _SIDE[key] = _lib.own_stream(torch.device("cuda", key[1]))
    return _SIDE[key]


class AncestralSampler(_SamplerBase):
    """diffusion/sampler.py:8-76."""
    _desc = "Ancestral Sampling"

    def __init__(self, scheduler, num_inference_steps: int = 1000, eta: float = 1.0):
        self._init(scheduler, num_inference_steps, eta)

    def step(self, model_output, timestep, sample, noise=None):
        return self.scheduler.step(model_output, timestep, sample, eta=self.eta, noise=noise)


class DDIMSampler(_SamplerBase):
    """diffusion/sampler.py:79-172."""
    _desc = "DDIM Sampling"

    def __init__(self, scheduler, num_inference_steps: int = 50, eta: float = 0.0):
        self._init(scheduler, num_inference_steps, eta)

    def step(self, model_output, timestep, sample, prev_timestep=None, noise=None):
        # prev_timestep is accepted and ignored, exactly like the reference (sampler.py:117-118)
        return self.scheduler.step(model_output, timestep, sample, eta=self.eta, noise=noise)


def create_sampler(sampler_type: str, scheduler, **kwargs):
    """diffusion/sampler.py:175-192."""
    if sampler_type == "ancestral":
        return AncestralSampler(scheduler, **kwargs)
    elif sampler_type == "ddim":
        return DDIMSampler(scheduler, **kwargs)
    else:
        raise ValueError(f"Unknown sampler type: {sampler_type}")
