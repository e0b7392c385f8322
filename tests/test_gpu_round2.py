"""Round-2 GPU parity tests (through the C ABI): the gaps VERDICT r1 / ADVICE r1 named.

* full model below 32x32 (GroupNorm-partials buffers: 8x8 writes 4 partials per image)
* hipGraph cache: different step counts / samplers / kept outputs on one model and workspace
* DDIMScheduler.step (schedulers.py:153-197), [B]-timestep scheduler.step (schedulers.py:113-115),
  generate_progressive (sample.py:223-265), final_conv per kernel (model.py:111-115)
"""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

import diffusion_model_amd as d
from diffusion_model_amd import _lib
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
EPS_TOL = 1e-4


@pytest.fixture(scope="module")
def model(sd):
    m = d.UNetMini(image_size=32).eval()
    m.load_state_dict(sd)
    return m


def cpu_sd(m):
    return {k: v.detach().cpu() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("size,B", [(8, 5), (16, 5), (8, 1), (16, 33)])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_full_model_below_32x32(sd, size, B, precision):
    """at 8x8 the transposed conv of up3 writes 4 statistics partials per image; the
    workspace must hold them (no overflow into the neighbouring buffer / the time-embedding table)."""
    m = d.UNetMini(image_size=size, precision=precision).eval()
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(size + B)
    x = torch.randn((B, 1, size, size), generator=g)
    t = torch.tensor(321)
    if precision == "bf16" and size < 16:  # documented limit of the bf16 kernels (64 halo tiles of 4x4 do not fit LDS)
        with pytest.raises(RuntimeError, match=">= 16"):
            m(x.to(DEV), t)
        return
    with torch.no_grad():
        ref = O.unet_forward(sd, x, t)
    full = m(x.to(DEV), t)
    tol = EPS_TOL if precision == "fp32" else 2e-2
    assert (full.cpu() - ref).abs().max().item() <= tol
    again = m(x.to(DEV), t)  # deterministic (an overflow onto a buffer the same launch reads is a race)
    assert torch.equal(full, again)
    for lo, hi in ((0, 1), (B // 2, B)):
        assert torch.equal(m(x[lo:hi].contiguous().to(DEV), t), full[lo:hi])
    # a short sampling chain: heads table + statistics buffers stay intact over the steps
    smp = d.DDIMSampler(d.DDPMScheduler(), num_inference_steps=6, eta=0.0)
    out = smp.sample_from(m, x).cpu()
    refs = O.sample(sd, O.make_schedule(), x.shape, 6, 0.0, x_T=x)
    assert (out - refs).abs().max().item() <= (1e-3 if precision == "fp32" else 1e-1)


def test_graph_cache_different_step_counts_and_kept_outputs(model):
    """n=100 then n=50 with B < n on ONE model/workspace (generate_comparison's pattern) must not
    replay a graph captured for another step count; outputs the caller keeps must not force re-captures."""
    g = torch.Generator().manual_seed(77)
    x_T = torch.randn((3, 1, 32, 32), generator=g)
    sched = d.DDPMScheduler()
    res = {}
    for n in (100, 50, 100, 20, 50):
        smp = d.DDIMSampler(sched, num_inference_steps=n, eta=0.0)
        smp.use_graph = True
        a = smp.sample_from(model, x_T)
        smp.use_graph = False
        b = smp.sample_from(model, x_T)
        assert torch.equal(a, b), n
        if n in res:
            assert torch.equal(res[n], a), n
        res[n] = a
    native = model._native()
    # kept outputs: 6 calls with live result tensors reuse ONE captured graph (pointer-stable staging)
    smp = d.DDIMSampler(sched, num_inference_steps=20, eta=0.0)
    keep = [smp.sample_from(model, x_T) for _ in range(6)]
    assert all(torch.equal(k, keep[0]) for k in keep)
    assert torch.equal(keep[0], res[20])


def test_graph_cache_eta_noise_kept_and_host_noise(model):
    """eta > 0: per-step noise handed over as a HOST tensor (staged chunk by chunk) == as a device tensor ==
    eager; a second call with other noise values through the cached graph uses the new values."""
    g = torch.Generator().manual_seed(5)
    x_T = torch.randn((2, 1, 32, 32), generator=g)
    sched = d.DDPMScheduler()
    smp = d.AncestralSampler(sched, num_inference_steps=125, eta=1.0)  # stride 8: 125 steps = 2 full chunks + a 25-step tail
    assert sched.timesteps.numel() == 125
    outs = []
    for k in range(2):
        nz = torch.randn((125, 2, 1, 32, 32), generator=g)
        smp.use_graph = True
        a = smp.sample_from(model, x_T, nz)            # host noise
        b = smp.sample_from(model, x_T, nz.to(DEV))    # device noise
        smp.use_graph = False
        c = smp.sample_from(model, x_T, nz)
        assert torch.equal(a, b) and torch.equal(a, c), k
        outs.append(a)
    assert not torch.equal(outs[0], outs[1])


def test_cfg_sampler_swap_reuses_graph_safely(model, sd):
    """in CFG mode the time embedding runs INSIDE the graph; a second sampler with another
    timestep/coefficient table and otherwise identical arguments must not see the first one's tables."""
    cd = torch.load(os.path.join(os.path.dirname(__file__), "golden", "classcond11_seed43.pth"), weights_only=True)
    cc = d.ClassConditioner(11, 128)
    cc.load_state_dict(cd)
    B, n, w = 3, 10, 3.0
    torch.manual_seed(21)
    x_T = torch.randn(B, 1, 32, 32)
    lab = torch.tensor([2, 7, 9])
    null = cc(torch.tensor([10]))
    for T in (1000, 500, 1000):
        sched = d.DDPMScheduler(num_train_timesteps=T)
        smp = d.DDIMSampler(sched, num_inference_steps=n, eta=0.0)
        out = smp.sample_from(model, x_T, cond_emb=cc(lab), null_emb=null, guidance_scale=w).cpu()
        ref = O.cfg_sample(sd, cd, O.make_schedule(T), x_T, lab, 10, w, n)
        assert (out - ref).abs().max().item() <= 1e-3, T


def test_ddim_scheduler_step_vs_golden(golden):
    """DDIMScheduler (schedulers.py:153-197; never instantiated by the reference's callers): same update as
    DDPMScheduler.step, eta stored on the instance.  Teacher-forced on the reference's own DDIM chains."""
    s0 = d.DDIMScheduler()  # eta = 0
    s0.set_timesteps(50)
    for i in (0, 1, 25, 48):
        xn = s0.step(golden["ddim50_es"][i].to(DEV), s0.timesteps[i], golden["ddim50_xs"][i].to(DEV)).cpu()
        ref = golden["ddim50_xs"][i + 1]
        assert (xn - ref).abs().max().item() <= 1e-6 * max(1.0, ref.abs().max().item())
    xn = s0.step(golden["ddim50_es"][49].to(DEV), s0.timesteps[49], golden["ddim50_xs"][49].to(DEV)).cpu()
    assert (xn - golden["ddim50_out"]).abs().max().item() <= 1e-6 * max(1.0, golden["ddim50_out"].abs().max().item())
    s5 = d.DDIMScheduler(eta=0.5)
    s5.set_timesteps(50)
    for i in (0, 7, 48):
        xn = s5.step(golden["ddim50eta_es"][i].to(DEV), s5.timesteps[i], golden["ddim50eta_xs"][i].to(DEV),
                     noise=golden["ddim50eta_noises_kept"][i]).cpu()
        ref = golden["ddim50eta_xs"][i + 1]
        assert (xn - ref).abs().max().item() <= 1e-6 * max(1.0, ref.abs().max().item())
    # explicit eta overrides the stored one
    a = s5.step(golden["ddim50_es"][3].to(DEV), 60, golden["ddim50_xs"][3].to(DEV), eta=0.0).cpu()
    assert torch.equal(a, s0.step(golden["ddim50_es"][3].to(DEV), 60, golden["ddim50_xs"][3].to(DEV)).cpu())
    assert len(s5) == 1000 and s5.eta == 0.5


@pytest.mark.parametrize("eta", [0.0, 0.7])
def test_scheduler_step_with_batched_timesteps(eta):
    """schedulers.py:113-115: a [B] timestep tensor gives per-image (`.view(-1,1,1,1)`) coefficients; one launch."""
    torch.manual_seed(8)
    sch = d.DDPMScheduler()
    osch = O.make_schedule()
    B = 6
    x, e, nz = torch.randn(B, 1, 32, 32), torch.randn(B, 1, 32, 32), torch.randn(B, 1, 32, 32)
    t = torch.tensor([0, 1, 20, 500, 999, 20])
    out = sch.step(e.to(DEV), t.to(DEV), x.to(DEV), eta=eta, noise=nz).cpu()
    for b in range(B):  # bit-exact against the strict-IEEE restatement, image by image
        ref = O.sampler_step_np(osch, e[b:b + 1], int(t[b]), x[b:b + 1], eta, nz[b:b + 1])
        assert torch.equal(out[b:b + 1], ref), b
    ref_t = O.sampler_step(osch, e, t, x, eta, nz)
    assert (out - ref_t).abs().max().item() <= 1e-6 * max(1.0, ref_t.abs().max().item())
    with pytest.raises(RuntimeError):
        sch.step(e.to(DEV), torch.tensor([1, 2]), x.to(DEV))


def test_generate_progressive_vs_oracle(tmp_path, ckpt):
    """sample.py:223-265: per-step model / scheduler.step calls outside any graph, snapshots every k steps."""
    from diffusion_model_amd import sample as S
    path = os.path.join(tmp_path, "ck.pth")
    torch.save({"epoch": 1, "model_state_dict": ckpt["model_state_dict"], "ema_shadow": None, "loss": 0.0}, path)
    cfg = S.get_default_config()
    cfg["checkpoint_path"] = path
    drv = S.DDPMSampler(cfg, sampler_type="ddim", num_inference_steps=20, eta=0.0)
    out_dir = os.path.join(tmp_path, "prog")
    torch.manual_seed(3)
    out = drv.generate_progressive(num_samples=2, output_dir=out_dir, eta=0.0, num_inference_steps=20, save_every=10)
    torch.manual_seed(3)
    rec = []
    ref = O.sample(ckpt["model_state_dict"], O.make_schedule(), (2, 1, 32, 32), 20, 0.0, record=rec)
    assert (out.cpu() - ref).abs().max().item() <= 1e-3
    names = sorted(os.listdir(out_dir))
    assert names == ["progressive_final.png", "progressive_noise.png", "progressive_step_010.png", "progressive_step_020.png"]
    for nm in names:
        assert open(os.path.join(out_dir, nm), "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    # eta > 0: the per-step draws come from the global CPU generator in loop order
    torch.manual_seed(4)
    out2 = drv.generate_progressive(num_samples=2, output_dir=out_dir, eta=0.5, num_inference_steps=10, save_every=5)
    torch.manual_seed(4)
    ref2 = O.sample(ckpt["model_state_dict"], O.make_schedule(), (2, 1, 32, 32), 10, 0.5)
    assert (out2.cpu() - ref2).abs().max().item() <= 1e-3


@pytest.mark.parametrize("B,H,C,Cout", [(3, 32, 64, 1), (2, 16, 64, 1), (2, 8, 64, 3), (2, 64, 256, 3), (70, 32, 64, 1)])
def test_final_conv_kernel(B, H, C, Cout):
    """final_conv (model.py:111-115,151): GroupNorm(8) -> SiLU -> Conv2d 1x1, from (mean, M2) partials."""
    torch.manual_seed(B + H + C)
    L = _lib.lib()
    u = torch.randn(B, C, H, H) * 1.5 + 0.7
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C) * 0.1
    w, b = torch.randn(Cout, C, 1, 1) * 0.1, torch.randn(Cout)
    pin = 8 if H >= 16 else 4
    rows = H // pin
    # partials over `rows` image rows x W x (C/8) channels each, layout [B][pin][8][2]
    ug = u.reshape(B, 8, C // 8, pin, rows, H).permute(0, 3, 1, 2, 4, 5).reshape(B, pin, 8, -1)
    mean = ug.mean(-1)
    m2 = ((ug - mean[..., None]) ** 2).sum(-1)
    st = torch.stack([mean, m2], -1).contiguous().to(DEV)
    cnt = float(ug.shape[-1])
    und = u.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = torch.full((B, Cout, H, H), float("nan"), device=DEV)
    gd, bd, wd, bbd = gamma.to(DEV), beta.to(DEV), w.reshape(Cout, C).contiguous().to(DEV), b.to(DEV)
    _lib.check(L.dm_final_conv(_lib.ptr(und), _lib.ptr(st), pin, cnt, _lib.ptr(gd), _lib.ptr(bd), _lib.ptr(wd), _lib.ptr(bbd),
                               _lib.ptr(out), B, H, H, C, Cout, torch.cuda.current_stream().cuda_stream))
    ref = F.conv2d(F.silu(F.group_norm(u, 8, gamma, beta, eps=1e-5)), w, b)
    assert (out.cpu() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_class_conditioner_empty_labels():
    cc = d.ClassConditioner(11, 128)
    out = cc(torch.empty(0, dtype=torch.int64))
    assert out.shape == (0, 128)


# ---------------------------------------------------------------- bf16 margin
BF16_EPS_TOL = 2e-2      # north_star: per-step eps max-abs <= 2e-2 in bf16
BF16_EPS_MARGIN = 0.85e-2   # measured worst case over three seeds x 64 images x 50 steps: 0.66e-2 (north_star bound: 2e-2)
#   measured (round 3: the layers behind a GroupNorm in down1 / down2.conv2 / up3 and every layer of the fused low-resolution
#   kernel take their MFMA operands as IEEE half, one half weight per product): 0.63e-2 / 0.66e-2 / 0.65e-2 on the three seeds;
#   with bf16 operands everywhere (DM_BF16_F16=0, the round-2 form) 1.33e-2 / 1.46e-2 / 1.25e-2; the round-1 build gave
#   1.95e-2 / 2.02e-2 / 1.98e-2 on the same chains


@pytest.mark.parametrize("seed", [42, 7, 123])
def test_bf16_parity_margin_three_seeds_b64_all_steps(seed):
    """bf16 (BASELINE configs[3]) teacher-forced on the CPU oracle's own DDIM-50 chain: 3 weight/noise seeds,
    batch 64, all 50 steps.  The round-1 build measured 1.95e-2 on one B=4 chain: 97.5 % of the budget."""
    d.set_seed(seed)
    m32 = d.UNetMini(image_size=32).eval()
    sdc = cpu_sd(m32)
    mb = d.UNetMini(image_size=32, precision="bf16").eval()
    mb.load_state_dict(sdc)
    B = 64
    x_T = torch.randn((B, 1, 32, 32), generator=torch.Generator().manual_seed(seed + 1))
    rec = []
    ref_out = O.sample(sdc, O.make_schedule(), x_T.shape, 50, 0.0, x_T=x_T, record=rec)
    sched = d.DDPMScheduler()
    smp = d.DDIMSampler(sched, num_inference_steps=50, eta=0.0)
    worst, worst32 = 0.0, 0.0
    for i, (t, x, eps, _) in enumerate(rec):
        e = mb(x.to(DEV), sched.timesteps[i]).cpu()
        worst = max(worst, (e - eps).abs().max().item())
        if i % 10 == 0:
            worst32 = max(worst32, (m32(x.to(DEV), sched.timesteps[i]).cpu() - eps).abs().max().item())
    out = smp.sample_from(mb, x_T).cpu()
    psnr = O.psnr(out, ref_out, float(ref_out.max() - ref_out.min()))
    print(f"bf16 seed {seed}: worst per-step eps err {worst:.4e} (fp32 path {worst32:.2e}), free-running PSNR min {psnr.min().item():.1f} dB")
    assert worst32 <= EPS_TOL
    assert worst <= BF16_EPS_MARGIN, worst
    assert psnr.min().item() >= 40.0


def test_bf16_half_operand_forms_agree_and_the_default_is_the_closest(monkeypatch):
    """bf16 path, MFMA operands of the layers behind a GroupNorm (conv_bf16_kernel's F16 forms, DM_BF16_F16 read when the model
    is created): 0 = bf16 operands with (hi, lo) bf16 weights, 1 = half operands with (hi, lo) half weights, 2 (default) = half
    operands with ONE half weight - and, in that form only, the fused low-resolution kernel with one half term per operand
    (lowres_fused_kernel<4>).  All three stay inside the bf16 bound against the fp32 oracle; the half forms are closer to
    it than the bf16 form, and the one-weight form is not worse than the weight pair (its 2^-12 weight rounding is the size
    of the activation operand's; the fused kernel's half planes more than make up for it)."""
    d.set_seed(3)
    sdc = cpu_sd(d.UNetMini(image_size=32).eval())
    x = torch.randn((16, 1, 32, 32), generator=torch.Generator().manual_seed(5))
    errs = {}
    for t in (30, 500, 970):
        with torch.no_grad():
            ref = O.unet_forward(sdc, x, torch.tensor(t))
        for mode in (0, 1, 2):
            monkeypatch.setenv("DM_BF16_F16", str(mode))
            mb = d.UNetMini(image_size=32, precision="bf16").eval()
            mb.load_state_dict(sdc)
            e = mb(x.to(DEV), torch.tensor(t)).cpu()
            errs.setdefault(mode, []).append(((e - ref).abs().max().item(), (e - ref).pow(2).mean().sqrt().item()))
    monkeypatch.delenv("DM_BF16_F16")
    print("bf16 operand forms (max, rms) per t:", errs)
    for mode in (0, 1, 2):
        assert max(m for m, _ in errs[mode]) <= BF16_EPS_TOL
    rms = {mode: sum(r for _, r in errs[mode]) / 3 for mode in errs}
    assert rms[1] < 0.8 * rms[0] and rms[2] < 0.8 * rms[0], rms
    assert rms[2] < 1.25 * rms[1], rms


@pytest.mark.parametrize("B,T,C", [(3, 256, 256), (2, 64, 64), (5, 128, 64), (2, 256, 64), (2, 64, 512), (130, 64, 64)])
def test_attention_fused_kernel(B, T, C):
    """Build-defined attention core (parity unpinned by the reference: it has no attention): fused QK^T -> softmax -> PV
    against torch-CPU on the same inputs."""
    import math
    torch.manual_seed(B + T + C)
    L = _lib.lib()
    qkv = torch.randn(B, T, 3 * C)
    qkv[..., :C] *= 2.0  # sharper score rows: the max-subtraction matters
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    ref = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), -1) @ v
    qd = qkv.to(DEV)
    o = torch.full((B, T, C), float("nan"), device=DEV)
    _lib.check(L.dm_attention(_lib.ptr(qd), _lib.ptr(o), B, T, C, torch.cuda.current_stream().cuda_stream))
    assert (o.cpu() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    rc = L.dm_attention(_lib.ptr(qd), _lib.ptr(o), B, 96, C, torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"fused kernel" in L.dm_last_error()


def test_steps_not_dividing_T_and_too_many_steps(golden, model, sd):
    """schedulers.py:146: stride T//n.  n=30 gives stride 33, i.e. 31 ascending steps 0, 33, ..., 990 (golden timesteps_30 holds the reference's list);
    n > T gives stride 0, which torch.arange rejects in the reference and here alike."""
    sched = d.DDPMScheduler()
    smp = d.DDIMSampler(sched, num_inference_steps=30, eta=0.0)
    assert torch.equal(sched.timesteps.cpu(), golden["timesteps_30"]) and sched.timesteps.numel() == 31
    x_T = golden["ddim50_x_T"]
    for use_graph in (True, False):
        smp.use_graph = use_graph
        out = smp.sample_from(model, x_T).cpu()
        ref = O.sample(sd, O.make_schedule(), x_T.shape, 30, 0.0, x_T=x_T)
        assert (out - ref).abs().max().item() <= 1e-3
    with pytest.raises(RuntimeError):
        d.DDIMSampler(d.DDPMScheduler(), num_inference_steps=2000)


def test_cosine_schedule_sampling_and_ragged_batches(model, sd):
    """Cosine betas (schedulers.py:65-72) through the whole loop; batch sizes that leave partial image groups in the
    multi-image tiles of the low-resolution layers (ni = 2 at 8x8, 4 at 4x4) and a batch of one."""
    sched = d.DDPMScheduler(beta_schedule="cosine")
    osch = O.make_schedule(beta_schedule="cosine")
    smp = d.DDIMSampler(sched, num_inference_steps=8, eta=0.3)
    for B in (1, 3, 7):
        g = torch.Generator().manual_seed(100 + B)
        x_T = torch.randn((B, 1, 32, 32), generator=g)
        nz = torch.randn((8, B, 1, 32, 32), generator=g)
        out = smp.sample_from(model, x_T, nz).cpu()
        ref = O.sample(sd, osch, x_T.shape, 8, 0.3, x_T=x_T, noises=nz)
        assert (out - ref).abs().max().item() <= 1e-3 * max(1.0, ref.abs().max().item()), B
