"""Non-square images (the reference's forward takes any H, W divisible by 8; model.py:120-151): every kernel's
tile geometry with H != W, against the CPU oracle, plus batch independence and a short sampling chain."""
import pytest
import torch

import diffusion_model_amd as d
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
EPS_TOL = 1e-4


@pytest.fixture(scope="module")
def sd():
    d.set_seed(42)
    m = d.UNetMini(image_size=32).eval()
    return {k: v.detach().cpu() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("H,W,B", [(32, 64, 3), (64, 32, 3), (16, 32, 5), (32, 16, 5), (8, 16, 4), (16, 8, 4), (8, 32, 2),
                                   (64, 16, 2), (32, 128, 1), (128, 32, 1)])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_forward_non_square(sd, H, W, B, precision):
    m = d.UNetMini(image_size=max(H, W), precision=precision).eval()
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(H * 1000 + W)
    x = torch.randn((B, 1, H, W), generator=g)
    t = torch.tensor(417)
    if precision == "bf16" and min(H, W) < 16:
        with pytest.raises(RuntimeError, match=">= 16"):
            m(x.to(DEV), t)
        return
    with torch.no_grad():
        ref = O.unet_forward(sd, x, t)
    full = m(x.to(DEV), t)
    err = (full.cpu() - ref).abs().max().item()
    assert err <= (EPS_TOL if precision == "fp32" else 2e-2), err
    assert torch.equal(full, m(x.to(DEV), t))
    if B > 1:
        assert torch.equal(m(x[B - 1:].contiguous().to(DEV), t), full[B - 1:])
    smp = d.DDIMSampler(d.DDPMScheduler(), num_inference_steps=5, eta=0.0)
    out = smp.sample_from(m, x).cpu()
    refs = O.sample(sd, O.make_schedule(), x.shape, 5, 0.0, x_T=x)
    assert (out - refs).abs().max().item() <= (1e-3 if precision == "fp32" else 1e-1)


def test_empty_batch(sd):
    """B = 0: the reference's stock ops return empty tensors (forward, both sampler loops, scheduler.step)."""
    m = d.UNetMini(image_size=32).eval()
    m.load_state_dict(sd)
    x = torch.zeros((0, 1, 32, 32), device=DEV)
    assert m(x, torch.tensor(5)).shape == (0, 1, 32, 32)
    sched = d.DDPMScheduler()
    assert d.DDIMSampler(sched, num_inference_steps=5).sample_from(m, x.cpu()).shape == (0, 1, 32, 32)
    assert d.AncestralSampler(sched, num_inference_steps=10)(m, (0, 1, 32, 32), progress_bar=False).shape == (0, 1, 32, 32)
    assert sched.step(x, 3, x).shape == (0, 1, 32, 32)
    assert sched.step(x, torch.zeros(0, dtype=torch.int64), x).shape == (0, 1, 32, 32)


def test_noise_staging_three_chunks_host_and_device_noise(sd):
    """dm_sample stages the per-step noise 50 steps at a time: n = 130 is three chunks (50, 50, 30).  Pageable, pinned and
    device noise, graph and eager, must equal the step-by-step loop (model(x, t) + scheduler.step: no staging at all) bit
    for bit; a second call with other noise right behind the first must not see the first call's chunks."""
    m = d.UNetMini(image_size=32).eval()
    m.load_state_dict(sd)
    sched = d.DDPMScheduler()
    smp = d.AncestralSampler(sched, num_inference_steps=130, eta=1.0)
    n = sched.timesteps.numel()
    assert n > 100
    g = torch.Generator().manual_seed(5)
    x_T = torch.randn((3, 1, 32, 32), generator=g)
    nz_a = torch.randn((n, 3, 1, 32, 32), generator=g)
    nz_b = torch.randn((n, 3, 1, 32, 32), generator=g)

    def loop(nz):  # any callable that is not a UNetMini is driven step by step
        return smp.sample_from(lambda x, t: m(x, t), x_T, nz)

    ref_a, ref_b = loop(nz_a), loop(nz_b)
    assert not torch.equal(ref_a, ref_b)
    for use_graph in (True, False):
        smp.use_graph = use_graph
        for src in (nz_a, nz_a.pin_memory(), nz_a.to(DEV)):
            assert torch.equal(smp.sample_from(m, x_T, src), ref_a), (use_graph, src.device, src.is_pinned())
        a = smp.sample_from(m, x_T, nz_a.pin_memory())
        b = smp.sample_from(m, x_T, nz_b.pin_memory())  # enqueued behind a: the staging area is rewritten only after a's steps
        assert torch.equal(a, ref_a) and torch.equal(b, ref_b), use_graph


def test_full_size_shard_ddim50_rows_bit_identical_and_match_oracle(sd):
    """BASELINE configs[2] at the size of one GPU's shard (8,192 images, DDIM-50): what each of 8 ranks of a 1,024-image
    sharding would compute alone equals the same rows of the 8,192-image run bit for bit (size-independent property of
    the sharded path), and a few rows agree with the CPU oracle after 50 free-running steps."""
    m = d.UNetMini(image_size=32).eval()
    m.load_state_dict(sd)
    smp = d.DDIMSampler(d.DDPMScheduler(), num_inference_steps=50, eta=0.0)
    g = torch.Generator().manual_seed(8192)
    x_T = torch.randn((8192, 1, 32, 32), generator=g)
    full = smp.sample_from(m, x_T)
    assert torch.isfinite(full).all()
    for lo in (0, 3072, 7168):
        part = smp.sample_from(m, x_T[lo:lo + 1024].contiguous())
        assert torch.equal(part, full[lo:lo + 1024]), lo
    ref = O.sample(sd, O.make_schedule(), (4, 1, 32, 32), 50, 0.0, x_T=x_T[8188:])
    assert (full[8188:].cpu() - ref).abs().max().item() <= 1e-3
