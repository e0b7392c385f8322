"""Per-kernel parity (GPU): each HIP kernel through the C ABI against the CPU oracle /
stock torch-CPU op on the same seeded inputs."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from diffusion_model_amd import _lib
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def stream():
    return torch.cuda.current_stream().cuda_stream


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def pack(kind, w):
    L = _lib.lib()
    cin, cout = (w.shape[0], w.shape[1]) if kind == 2 else (w.shape[1], w.shape[0])
    out = torch.empty(L.dm_conv_packed_floats(kind, cin, cout), device=DEV)
    wd = w.to(DEV).contiguous()
    _lib.check(L.dm_pack_conv_weight(kind, _lib.ptr(wd), cin, cout, _lib.ptr(out), stream()))
    return out


def run_conv(kind, x1, x2, w, b, tf=None, want_stats=False):
    """x1/x2 NCHW cpu; returns (y NCHW cpu, stats or None)."""
    L = _lib.lib()
    B, c1, H, W = x1.shape
    c2 = 0 if x2 is None else x2.shape[1]
    cout = w.shape[1] if kind == 2 else w.shape[0]
    OH, OW = (H, W) if kind in (0, 5) else (H // 2, W // 2) if kind in (1, 6) else (2 * H, 2 * W)
    wp = pack(kind, w)
    a1 = nhwc(x1).to(DEV)
    a2 = None if x2 is None else nhwc(x2).to(DEV)
    bd = b.to(DEV)
    y = torch.full((B, OH, OW, cout), float("nan"), device=DEV)
    pout = L.dm_conv_stats_partials(kind, H, W)
    so = torch.full((B, pout, 8, 2), float("nan"), device=DEV) if want_stats else None
    si = g = be = te = None
    pin, cnt, ts = 0, 0.0, 0
    if tf is not None:
        si, pin, cnt, g, be, te, ts = tf
    _lib.check(L.dm_conv2d(kind, _lib.ptr(a1), c1, _lib.ptr(a2), c2, _lib.ptr(wp), _lib.ptr(bd), cout,
                           _lib.ptr(si), pin, cnt, _lib.ptr(g), _lib.ptr(be), _lib.ptr(te), ts,
                           _lib.ptr(y), _lib.ptr(so), B, H, W, stream()))
    torch.cuda.synchronize()
    return nchw(y.cpu()), (None if so is None else so.cpu())


def merge_stats(st, cnt):
    """[B,P,8,2] partials -> mean, var (biased) per (B, group)."""
    mean_p, m2_p = st[..., 0], st[..., 1]
    mean = mean_p.mean(1)
    m2 = m2_p.sum(1) + cnt * ((mean_p - mean[:, None]) ** 2).sum(1)
    return mean, m2 / (cnt * st.shape[1])


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("B,H", [(3, 32), (2, 16), (5, 8), (9, 4), (1, 4)])
def test_conv_plain(kind, B, H):
    if kind == 1 and H < 8:
        H = 8
    torch.manual_seed(kind * 100 + B + H)
    x = torch.randn(B, 64, H, H)
    if kind == 2:
        w = torch.randn(64, 64, 4, 4) * 0.05
    else:
        k = 3 if kind == 0 else 4
        w = torch.randn(64, 64, k, k) * 0.05
    b = torch.randn(64)
    y, st = run_conv(kind, x, None, w, b, want_stats=True)
    ref = (F.conv2d(x, w, b, padding=1) if kind == 0 else F.conv2d(x, w, b, stride=2, padding=1) if kind == 1
           else F.conv_transpose2d(x, w, b, stride=2, padding=1))
    assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # GroupNorm(8) partial statistics of the output
    P = st.shape[1]
    cnt = ref.shape[2] * ref.shape[3] * 8 / P
    mean, var = merge_stats(st, cnt)
    rg = ref.reshape(B, 8, -1)
    assert (mean - rg.mean(2)).abs().max().item() <= 1e-5
    assert ((var - rg.var(2, unbiased=False)).abs() / rg.var(2, unbiased=False)).max().item() <= 1e-5


@pytest.mark.parametrize("B,H,W,cin,cout", [(3, 32, 32, 64, 64), (2, 16, 16, 64, 64), (5, 8, 8, 64, 64), (1, 8, 8, 64, 64),
                                            (2, 64, 64, 256, 256), (2, 16, 32, 128, 64), (3, 8, 16, 64, 128)])
def test_conv_winograd(B, H, W, cin, cout):
    """kind 5: the 3x3 conv as Winograd F(2x2,3x3) (fp32 MFMA) against torch's direct conv + statistics."""
    torch.manual_seed(B + H + cin)
    x = torch.randn(B, cin, H, W)
    w = torch.randn(cout, cin, 3, 3) * (1.0 / (3.0 * cin ** 0.5))
    b = torch.randn(cout)
    y, st = run_conv(5, x, None, w, b, want_stats=True)
    ref = F.conv2d(x, w, b, padding=1)
    assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    P = st.shape[1]
    cnt = H * W * (cout // 8) / P
    mean, var = merge_stats(st, cnt)
    rg = ref.reshape(B, 8, -1)
    assert (mean - rg.mean(2)).abs().max().item() <= 1e-5
    assert ((var - rg.var(2, unbiased=False)).abs() / rg.var(2, unbiased=False)).max().item() <= 1e-5


@pytest.mark.parametrize("B,H", [(3, 32), (2, 64), (130, 32)])
def test_conv_down_128_pixel_tiles(B, H):
    """kind 6: the 4x4 stride-2 conv on 128-pixel tiles with 16-channel K chunks == kind 1's result."""
    torch.manual_seed(B + H)
    x = torch.randn(B, 64, H, H)
    w, b = torch.randn(64, 64, 4, 4) * 0.05, torch.randn(64)
    y, st = run_conv(6, x, None, w, b, want_stats=True)
    ref = F.conv2d(x, w, b, stride=2, padding=1)
    assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    P = st.shape[1]
    mean, var = merge_stats(st, (H // 2) ** 2 * 8 / P)
    rg = ref.reshape(B, 8, -1)
    assert (mean - rg.mean(2)).abs().max().item() <= 1e-5
    assert ((var - rg.var(2, unbiased=False)).abs() / rg.var(2, unbiased=False)).max().item() <= 1e-5
    # with the fused input transform
    gamma, beta, temb = torch.rand(64) + 0.5, torch.randn(64) * 0.1, torch.randn(1, 64)
    h_raw, st0 = run_conv(5, x, None, torch.randn(64, 64, 3, 3) * 0.1, b + 3.0, want_stats=True)
    P0 = st0.shape[1]
    tf = (st0.to(DEV), P0, H * H * 8 / P0, gamma.to(DEV), beta.to(DEV), temb.to(DEV), 0)
    y2, _ = run_conv(6, h_raw, None, w, b, tf=tf)
    hin = F.silu(F.group_norm(h_raw, 8, gamma, beta, eps=1e-5)) + temb[:, :, None, None]
    ref2 = F.conv2d(hin, w, b, stride=2, padding=1)
    assert (y2 - ref2).abs().max().item() <= 3e-5 * max(1.0, ref2.abs().max().item())


def test_conv_winograd_concat_and_fused_transform():
    torch.manual_seed(77)
    B, H = 3, 16
    x1, x2 = torch.randn(B, 64, H, H), torch.randn(B, 64, H, H)
    w, b = torch.randn(64, 128, 3, 3) * 0.05, torch.randn(64)
    y, _ = run_conv(5, x1, x2, w, b)
    ref = F.conv2d(torch.cat([x1, x2], 1), w, b, padding=1)
    assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    for Hh, per_img in ((32, False), (8, True)):
        x0 = torch.randn(B, 64, Hh, Hh)
        w0, b0 = torch.randn(64, 64, 3, 3) * 0.1, torch.randn(64) + 3.0
        gamma, beta = torch.rand(64) + 0.5, torch.randn(64) * 0.1
        temb = torch.randn(B if per_img else 1, 64)
        w2, b2 = torch.randn(64, 64, 3, 3) * 0.05, torch.randn(64)
        h_raw, st = run_conv(5, x0, None, w0, b0, want_stats=True)
        P = st.shape[1]
        tf = (st.to(DEV), P, Hh * Hh * 8 / P, gamma.to(DEV), beta.to(DEV), temb.to(DEV), 64 if per_img else 0)
        y, _ = run_conv(5, h_raw, None, w2, b2, tf=tf)
        hin = F.silu(F.group_norm(h_raw, 8, gamma, beta, eps=1e-5)) + temb[:, :, None, None]
        ref = F.conv2d(hin, w2, b2, padding=1)
        assert (y - ref).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())
    L = _lib.lib()
    xx = torch.zeros(1, 4, 4, 64, device=DEV)
    rc = L.dm_conv2d(5, _lib.ptr(xx), 64, None, 0, _lib.ptr(xx), _lib.ptr(xx), 64, None, 0, 0.0, None, None, None, 0,
                     _lib.ptr(xx), None, 1, 4, 4, stream())
    assert rc != 0 and b"Winograd" in L.dm_last_error()


@pytest.mark.parametrize("kind", [0, 1, 2, 5])
def test_conv_large_batch_rows_match_small_batch(kind):
    """The fp32 kernels are persistent (each workgroup walks several tiles once the grid exceeds the
    resident slots): a batch large enough for several rounds must give, bit for bit, what the same
    images give alone.  (Regression: the fused transposed conv kept its parity state across tiles.)"""
    torch.manual_seed(3 + kind)
    B, H = 448, 16
    x = torch.randn(B, 64, H, H)
    k = 3 if kind in (0, 5) else 4
    w, b = torch.randn(64, 64, k, k) * 0.05, torch.randn(64)
    y, st = run_conv(kind, x, None, w, b, want_stats=True)
    for lo, hi in ((0, 32), (200, 232), (384, 448)):
        y2, st2 = run_conv(kind, x[lo:hi].contiguous(), None, w, b, want_stats=True)
        assert torch.equal(y[lo:hi], y2) and torch.equal(st[lo:hi], st2), (kind, lo)


@pytest.mark.parametrize("kind", [0, 1, 2, 5, 6])
def test_conv_xcd_tile_order_uneven_groups(kind):
    """131 images at 32x32: every kernel runs its XCD-aware tile order (image group g on XCD g & 7) with a
    different number of groups per XCD; all rows must equal what the images give alone, and the reference."""
    torch.manual_seed(40 + kind)
    B, H = 131, 32
    x = torch.randn(B, 64, H, H)
    k = 3 if kind in (0, 5) else 4
    w, b = torch.randn(64, 64, k, k) * 0.05, torch.randn(64)
    y, st = run_conv(kind, x, None, w, b, want_stats=True)
    assert torch.isfinite(y).all() and torch.isfinite(st).all()
    for lo, hi in ((0, 9), (60, 70), (120, 131)):
        y2, st2 = run_conv(kind, x[lo:hi].contiguous(), None, w, b, want_stats=True)
        assert torch.equal(y[lo:hi], y2) and torch.equal(st[lo:hi], st2), (kind, lo)
    xs = x[125:131]
    ref = (F.conv2d(xs, w, b, padding=1) if kind in (0, 5) else F.conv2d(xs, w, b, stride=2, padding=1) if kind in (1, 6)
           else F.conv_transpose2d(xs, w, b, stride=2, padding=1))
    assert (y[125:131] - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("H", [16, 4])
def test_conv_concat(H):
    torch.manual_seed(5)
    B = 3
    x1, x2 = torch.randn(B, 64, H, H), torch.randn(B, 64, H, H)
    w, b = torch.randn(64, 128, 3, 3) * 0.05, torch.randn(64)
    y, _ = run_conv(0, x1, x2, w, b)
    ref = F.conv2d(torch.cat([x1, x2], 1), w, b, padding=1)
    assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("H,per_image_temb", [(32, False), (8, True), (4, False)])
def test_conv_fused_groupnorm_silu_temb(kind, H, per_image_temb):
    """conv(SiLU(GN(x)) + temb): the producer conv emits the statistics, the consumer applies them."""
    if kind == 1 and H < 8:
        H = 8
    torch.manual_seed(11 + kind + H)
    B = 3
    x0 = torch.randn(B, 64, H, H)
    w0, b0 = torch.randn(64, 64, 3, 3) * 0.1, torch.randn(64) + 3.0  # non-zero mean: stresses the variance
    gamma, beta = torch.rand(64) + 0.5, torch.randn(64) * 0.1
    temb = torch.randn(B if per_image_temb else 1, 64)
    k = 3 if kind == 0 else 4
    w = torch.randn(64, 64, k, k) * 0.05
    b = torch.randn(64)
    h_raw, st = run_conv(0, x0, None, w0, b0, want_stats=True)
    P = st.shape[1]
    cnt = H * H * 8 / P
    tf = (st.to(DEV), P, cnt, gamma.to(DEV), beta.to(DEV), temb.to(DEV), 64 if per_image_temb else 0)
    y, _ = run_conv(kind, h_raw, None, w, b, tf=tf)
    hin = F.silu(F.group_norm(h_raw, 8, gamma, beta, eps=1e-5)) + temb[:, :, None, None]
    ref = (F.conv2d(hin, w, b, padding=1) if kind == 0 else F.conv2d(hin, w, b, stride=2, padding=1) if kind == 1
           else F.conv_transpose2d(hin, w, b, stride=2, padding=1))
    assert (y - ref).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())


def test_conv_rejects_bad_shapes():
    L = _lib.lib()
    x = torch.zeros(1, 24, 24, 64, device=DEV)
    rc = L.dm_conv2d(0, _lib.ptr(x), 64, None, 0, _lib.ptr(x), _lib.ptr(x), 64, None, 0, 0.0, None, None, None, 0,
                     _lib.ptr(x), None, 1, 24, 24, stream())
    assert rc != 0 and b"powers of two" in L.dm_last_error()
    rc = L.dm_conv2d(0, _lib.ptr(x), 48, None, 0, _lib.ptr(x), _lib.ptr(x), 64, None, 0, 0.0, None, None, None, 0,
                     _lib.ptr(x), None, 1, 16, 16, stream())
    assert rc != 0 and b"multiples of 64" in L.dm_last_error()


@pytest.mark.parametrize("eta", [0.0, 0.7])
def test_sampler_step_bit_exact(eta):
    torch.manual_seed(3)
    import diffusion_model_amd as d
    sch = d.DDPMScheduler()
    osch = O.make_schedule()
    x, e, nz = torch.randn(4, 1, 32, 32), torch.randn(4, 1, 32, 32), torch.randn(4, 1, 32, 32)
    for t in (0, 1, 20, 500, 999):
        out = sch.step(e.to(DEV), t, x.to(DEV), eta=eta, noise=nz).cpu()
        ref = O.sampler_step_np(osch, e, t, x, eta, nz)  # strict IEEE fp32, host-independent
        assert torch.equal(out, ref), (t, (out - ref).abs().max())
        ref_t = O.sampler_step(osch, e, torch.tensor(t), x, eta, nz)  # torch-CPU (MKL sqrt: +-1 ulp by host)
        assert (out - ref_t).abs().max().item() <= 1e-6 * max(1.0, ref_t.abs().max().item())


def test_time_embed(sd):
    import diffusion_model_amd as d
    m = d.UNetMini(image_size=32)
    m.load_state_dict(sd)
    L = _lib.lib()
    t = torch.tensor([0, 1, 20, 500, 980, 999])
    td = t.to(DEV)
    temb = torch.empty(len(t), 128, device=DEV)
    heads = torch.empty(len(t), 6, 64, device=DEV)
    _lib.check(L.dm_time_embed(m._native(), _lib.ptr(m.packed_weights()), _lib.ptr(td), len(t), None,
                               _lib.ptr(temb), _lib.ptr(heads), stream()))
    ref = O.time_mlp(sd, t, 128)
    assert (temb.cpu() - ref).abs().max().item() <= 2e-6
    for k, name in enumerate(["down1", "down2", "down3", "up1", "up2", "up3"]):
        rh = F.silu(F.linear(ref, sd[f"{name}.time_mlp.weight"], sd[f"{name}.time_mlp.bias"]))
        assert (heads[:, k].cpu() - rh).abs().max().item() <= 2e-6
