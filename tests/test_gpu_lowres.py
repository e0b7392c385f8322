"""The fused low-resolution kernel (dm_kernels_lowres.hip: down2.transform ... up2.transform of the 32x32 model in one
launch, activations in LDS) against the CPU oracle's block taps (model.py:140,143,146: down3, bottleneck, up1), against
the layer-by-layer kernels it replaces, and for batch independence."""
import os

import pytest
import torch

import diffusion_model_amd as d
from diffusion_model_amd import _lib
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def make(sd, lowres, precision="fp32"):
    old = os.environ.get("DM_LOWRES")
    os.environ["DM_LOWRES"] = "1" if lowres else "0"
    try:
        m = d.UNetMini(image_size=32, precision=precision).eval()
        m._native()  # the handle reads DM_LOWRES when it is created
    finally:
        if old is None:
            del os.environ["DM_LOWRES"]
        else:
            os.environ["DM_LOWRES"] = old
    m.load_state_dict(sd)
    return m


@pytest.mark.parametrize("B,tshape", [(5, "scalar"), (3, "batch"), (1, "scalar")])
def test_fused_lowres_taps_match_the_oracle(sd, B, tshape):
    L = _lib.lib()
    m = make(sd, True)
    g = torch.Generator().manual_seed(40 + B)
    x = torch.randn((B, 1, 32, 32), generator=g)
    t = torch.tensor(417) if tshape == "scalar" else torch.randint(0, 1000, (B,), generator=g)
    taps = {}
    with torch.no_grad():
        ref = O.unet_forward(sd, x, t, taps=taps)
    buf = torch.full((B, 6144), float("nan"), device=DEV)
    L.dm_debug_lowres_taps(_lib.ptr(buf))
    try:
        eps = m(x.to(DEV), t.to(DEV))
        torch.cuda.synchronize()
    finally:
        L.dm_debug_lowres_taps(None)
    assert torch.isfinite(buf).all(), "the fused kernel did not run (or skipped a tap)"
    got = {"down3": buf[:, :1024].reshape(B, 4, 4, 64), "bott": buf[:, 1024:2048].reshape(B, 4, 4, 64),
           "up1": buf[:, 2048:].reshape(B, 8, 8, 64)}
    for k, v in got.items():
        r = taps[k]
        err = (v.permute(0, 3, 1, 2).cpu() - r).abs().max().item()
        assert err <= 2e-5 * max(1.0, r.abs().max().item()), (k, err)
    assert (eps.cpu() - ref).abs().max().item() <= 1e-4


def test_fused_lowres_vs_layer_kernels_and_batch_independence(sd):
    mf, ml = make(sd, True), make(sd, False)
    g = torch.Generator().manual_seed(8)
    x = torch.randn((37, 1, 32, 32), generator=g).to(DEV)
    t = torch.tensor(600)
    ef, el = mf(x, t), ml(x, t)
    assert (ef - el).abs().max().item() <= 2e-5  # same arithmetic class (fp32-equivalent), different summation orders
    # per-image results do not depend on the batch (persistent workgroups walk images n, n + grid, ...)
    for lo, hi in ((0, 1), (36, 37), (5, 21)):
        assert torch.equal(mf(x[lo:hi].contiguous(), t), ef[lo:hi])
    big = torch.cat([x] * 16)  # 592 images: more than the 512 resident workgroups
    eb = mf(big, t)
    assert torch.equal(eb[:37], ef) and torch.equal(eb[-37:], ef)


def test_fused_lowres_sampling_chain_graph_and_eager(sd):
    m = make(sd, True)
    smp = d.DDIMSampler(d.DDPMScheduler(), num_inference_steps=10, eta=0.0)
    x_T = torch.randn((4, 1, 32, 32), generator=torch.Generator().manual_seed(2))
    outs = []
    for use_graph in (True, False):
        smp.use_graph = use_graph
        outs.append(smp.sample_from(m, x_T))
    assert torch.equal(outs[0], outs[1])
    ref = O.sample(sd, O.make_schedule(), x_T.shape, 10, 0.0, x_T=x_T)
    assert (outs[0].cpu() - ref).abs().max().item() <= 1e-3


def test_fused_lowres_bf16_form(sd):
    """The bf16 path's form of the fused kernel (one bf16 activation plane, (hi, lo) weight pairs, half in / bf16 out):
    inner taps and eps against the fp32 oracle within the bf16 path's bound (north_star: |eps - eps_fp32| <= 2e-2), against
    the bf16 layer kernels it replaces, and batch independence."""
    L = _lib.lib()
    mf, ml = make(sd, True, "bf16"), make(sd, False, "bf16")
    g = torch.Generator().manual_seed(21)
    B = 9
    x = torch.randn((B, 1, 32, 32), generator=g)
    t = torch.tensor(333)
    taps = {}
    with torch.no_grad():
        ref = O.unet_forward(sd, x, t, taps=taps)
    buf = torch.full((B, 6144), float("nan"), device=DEV)
    L.dm_debug_lowres_taps(_lib.ptr(buf))
    try:
        ef = mf(x.to(DEV), t)
        torch.cuda.synchronize()
    finally:
        L.dm_debug_lowres_taps(None)
    assert torch.isfinite(buf).all(), "the fused kernel did not run"
    got = {"down3": buf[:, :1024].reshape(B, 4, 4, 64), "bott": buf[:, 1024:2048].reshape(B, 4, 4, 64), "up1": buf[:, 2048:].reshape(B, 8, 8, 64)}
    for k, v in got.items():
        r = taps[k]
        assert (v.permute(0, 3, 1, 2).cpu() - r).abs().max().item() <= 3e-2 * max(1.0, r.abs().max().item()), k
    el = ml(x.to(DEV), t)
    # the maximum over 9 x 1,024 values moves by +-0.2e-2 with any change of summation order (conv_bf16_kernel); the
    # RMS error is the stable comparison: the fused form must not be worse than the layer kernels it replaces
    ef_err, el_err = (ef.cpu() - ref).abs().max().item(), (el.cpu() - ref).abs().max().item()
    ef_rms, el_rms = (ef.cpu() - ref).pow(2).mean().sqrt().item(), (el.cpu() - ref).pow(2).mean().sqrt().item()
    assert ef_err <= 2e-2 and ef_rms <= 1.05 * el_rms, (ef_err, el_err, ef_rms, el_rms)
    assert torch.equal(mf(x[4:6].contiguous().to(DEV), t), ef[4:6])
