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


def test_results_do_not_depend_on_kernels_of_other_streams(sd):
    """A model forward and a sampling chain must give the same bits whether or not another stream keeps the matrix pipe busy.
    (Found with two sampling chains on two streams: packed fp32 instructions whose low result half reads the high half of a
    source - written by hand in init5_kernel, and generated by hipcc's SLP vectoriser elsewhere - returned wrong low halves
    in lanes 48-63 while a bf16-MFMA kernel of another stream shared the SIMD: 35 of 40 forwards differed.  The library now
    contains no such instruction.)  The co-runner is the library's own split-bf16 transposed conv, launched back to back."""
    import torch.nn.functional as F  # noqa: F401
    from diffusion_model_amd import _lib
    from tests.test_gpu_kernels import pack
    L = _lib.lib()
    m = d.UNetMini(image_size=32).eval()
    m.load_state_dict(sd)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.Generator().manual_seed(3)
    x = torch.randn((32, 1, 32, 32), generator=g).to(DEV)
    t = torch.tensor(321)
    # co-runner: ConvTranspose2d 4x4 s2 on the bf16 pipe (kind 9), 16x16 -> 32x32, B = 32
    xa = torch.randn(32, 16, 16, 64, device=DEV)
    wa = torch.randn(64, 64, 4, 4) * 0.05
    wp = torch.empty(L.dm_conv_packed_floats(9, 64, 64), device=DEV)
    wd = wa.to(DEV)
    _lib.check(L.dm_pack_conv_weight(9, _lib.ptr(wd), 64, 64, _lib.ptr(wp), torch.cuda.current_stream().cuda_stream))
    ba = torch.zeros(64, device=DEV)
    ya = torch.empty(32, 32, 32, 64, device=DEV)
    torch.cuda.synchronize()

    # the co-runner stays RESIDENT for the whole forward - a background thread keeps its stream fed (at most 40
    # launches queued) until the forwards under test have finished; 30 launches enqueued up front could drain before the
    # Python-driven forward had enqueued its own kernels.
    import threading
    stop = threading.Event()
    launched = [0]

    def feeder():
        while not stop.is_set():
            for _ in range(20):
                L.dm_conv2d(9, _lib.ptr(xa), 64, None, 0, _lib.ptr(wp), _lib.ptr(ba), 64, None, 0, 0.0, None, None, None, 0,
                            _lib.ptr(ya), None, 32, 16, 16, s2.cuda_stream)
            launched[0] += 20
            s2.synchronize()

    d.set_seed(7)
    wide = d.UNetMini(image_size=64, in_channels=3, out_channels=3, model_channels=256, attention=True).eval()
    mb = d.UNetMini(image_size=32, precision="bf16").eval()
    mb.load_state_dict(sd)
    xw = torch.randn((2, 3, 64, 64), generator=g).to(DEV)
    refs = []
    for name, mm, xx, trials in (("fp32", m, x, 20), ("bf16", mb, x, 10), ("wide + attention", wide, xw, 6)):
        with torch.cuda.stream(s1):
            refs.append(mm(xx, t))
    torch.cuda.synchronize()
    th = threading.Thread(target=feeder, daemon=True)
    th.start()
    try:
        for (name, mm, xx, trials), ref in zip((("fp32", m, x, 20), ("bf16", mb, x, 10), ("wide + attention", wide, xw, 6)), refs):
            for trial in range(trials):
                before = launched[0]
                with torch.cuda.stream(s1):
                    out = mm(xx, t)
                s1.synchronize()
                assert torch.equal(out, ref), (name, trial)
                while launched[0] == before and th.is_alive():  # the co-runner did run beside this forward
                    s1.synchronize()
    finally:
        stop.set()
        th.join()
        torch.cuda.synchronize()
    # two sampling chains at once (each on its own stream and model instance) == each alone
    import diffusion_model_amd.sampler as sampler_mod
    m2 = d.UNetMini(image_size=32).eval()
    m2.load_state_dict(sd)
    smps = [d.DDIMSampler(d.DDPMScheduler(), num_inference_steps=20, eta=0.0) for _ in range(2)]
    xs = [torch.randn((16, 1, 32, 32), generator=g) for _ in range(2)]
    alone = [smps[c].sample_from(mm, xs[c]) for c, mm in enumerate((m, m2))]
    torch.cuda.synchronize()
    saved = sampler_mod._side_stream
    try:
        outs = []
        for c, mm in enumerate((m, m2)):
            sampler_mod._side_stream = lambda dev, c=c: (s1, s2)[c]
            with torch.cuda.stream((s1, s2)[c]):
                outs.append(smps[c].sample_from(mm, xs[c]))
        torch.cuda.synchronize()
    finally:
        sampler_mod._side_stream = saved
    assert torch.equal(outs[0], alone[0]) and torch.equal(outs[1], alone[1])


def test_batch_65536_on_one_gpu_crosses_the_32_bit_element_boundary(sd):
    """BASELINE configs[2] unsharded: 65,536 images on ONE GPU are 4.3e9 activation elements per full-resolution layer
    (past 2^32; 60 GiB of workspace).  Rows at the start, on both sides of the boundary and at the end equal the same rows
    sampled in a batch of 16."""
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 2**30:
        pytest.skip("needs 80 GiB of free device memory")
    m = d.UNetMini(image_size=32).eval()
    m.load_state_dict(sd)
    smp = d.DDIMSampler(d.DDPMScheduler(), num_inference_steps=4, eta=0.0)
    B = 65536
    g = torch.Generator().manual_seed(7)
    x_T = torch.randn((B, 1, 32, 32), generator=g)
    full = smp.sample_from(m, x_T)
    assert torch.isfinite(full).all()
    for lo in (0, B // 2 - 8, B // 2, B - 16):
        assert torch.equal(smp.sample_from(m, x_T[lo:lo + 16].contiguous()), full[lo:lo + 16]), lo
    m._ws = None
    del full
    torch.cuda.empty_cache()


def test_guided_sampling_on_the_widened_model():
    """Classifier-free guidance (build-defined) with C = 256, 64x64x3: the first layers are evaluated once for both halves of the
    doubled batch through four 64-channel output tiles (ConvArgs.in_wrap with grid.y = 4); against the CPU restatement."""
    from tests.test_gpu_parity import _conditioner
    cc, cd = _conditioner()
    d.set_seed(7)
    m = d.UNetMini(image_size=64, in_channels=3, out_channels=3, model_channels=256).eval()
    sdw = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    B, n, w = 2, 3, 3.0
    g = torch.Generator().manual_seed(31)
    x_T = torch.randn((B, 3, 64, 64), generator=g)
    lab = torch.tensor([4, 8])
    smp = d.DDIMSampler(d.DDPMScheduler(), num_inference_steps=n, eta=0.0)
    ref = O.cfg_sample(sdw, cd, O.make_schedule(), x_T, lab, 10, w, n)
    for use_graph in (False, True):
        smp.use_graph = use_graph
        out = smp.sample_from(m, x_T, cond_emb=cc(lab), null_emb=cc(torch.tensor([10])), guidance_scale=w).cpu()
        assert (out - ref).abs().max().item() <= 1e-3, use_graph
