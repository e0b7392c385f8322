"""Per-kernel parity (GPU) of the split-bf16 convolutions (conv_x6_kernel: fp32 operands as three bf16 terms, six bf16
MFMAs per product, fp32 accumulation) through the C ABI against stock torch-CPU fp32 ops — to the SAME bounds as the
fp32-MFMA kernels (tests/test_gpu_kernels.py): the arithmetic is fp32-equivalent, not a reduced-precision mode.
kinds 7 / 8 / 9 = Conv2d 3x3 (model.py:41,45), Conv2d 4x4 stride 2 (model.py:46), ConvTranspose2d 4x4 stride 2 (model.py:43).
kinds 13 / 14 / 15 = kinds 8 / 9 / 7 with the operands as TWO fp16 terms (three MFMAs per product), for inputs behind a GroupNorm - the
model's default for its stride-2 / transposed convolutions; same bounds."""
import pytest
import torch
import torch.nn.functional as F

from diffusion_model_amd import _lib
from tests.test_gpu_kernels import DEV, merge_stats, nchw, nhwc, stream

pytestmark = pytest.mark.gpu


def run_x6(kind, x1, x2, w, b, tf=None, want_stats=True):
    L = _lib.lib()
    B, c1, H, W = x1.shape
    c2 = 0 if x2 is None else x2.shape[1]
    cin = c1 + c2
    cout = w.shape[1] if kind in (9, 14) else w.shape[0]
    OH, OW = (H, W) if kind in (7, 15) else (H // 2, W // 2) if kind in (8, 13) else (2 * H, 2 * W)
    wp = torch.empty(L.dm_conv_packed_floats(kind, cin, cout), device=DEV)
    wd = w.to(DEV).contiguous()
    _lib.check(L.dm_pack_conv_weight(kind, _lib.ptr(wd), cin, cout, _lib.ptr(wp), stream()))
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


def ref_conv(kind, x, w, b):
    return (F.conv2d(x, w, b, padding=1) if kind in (7, 15) else F.conv2d(x, w, b, stride=2, padding=1) if kind in (8, 13)
            else F.conv_transpose2d(x, w, b, stride=2, padding=1))


def check_stats(st, ref, B):
    P = st.shape[1]
    cout = ref.shape[1]
    cnt = ref.shape[2] * ref.shape[3] * (cout // 8) / P
    mean, var = merge_stats(st, cnt)
    rg = ref.reshape(B, 8, -1)
    assert (mean - rg.mean(2)).abs().max().item() <= 1e-5
    assert ((var - rg.var(2, unbiased=False)).abs() / rg.var(2, unbiased=False)).max().item() <= 1e-5


@pytest.mark.parametrize("kind", [7, 8, 9])
@pytest.mark.parametrize("B,H", [(3, 32), (2, 16), (5, 8), (9, 4), (1, 4), (130, 16)])
def test_x6_conv_plain(kind, B, H):
    """Every tile shape of the model's levels (128- and 64-pixel tiles, several images per tile, ragged last tile)."""
    if kind == 8 and H < 8:
        H = 8
    torch.manual_seed(kind * 100 + B + H)
    x = torch.randn(B, 64, H, H) * 3.0 + 0.5
    w = torch.randn(64, 64, 3 if kind == 7 else 4, 3 if kind == 7 else 4) * 0.05
    b = torch.randn(64)
    y, st = run_x6(kind, x, None, w, b)
    ref = ref_conv(kind, x, w, b)
    assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    check_stats(st, ref, B)


@pytest.mark.parametrize("kind,B,H,cin,cout", [(7, 2, 16, 128, 64), (7, 2, 64, 256, 256), (8, 2, 64, 256, 256), (7, 3, 8, 512, 128),
                                               (8, 2, 16, 128, 256), (7, 3, 4, 128, 64), (9, 2, 32, 256, 256), (9, 3, 8, 128, 64),
                                               (9, 5, 4, 256, 128)])
def test_x6_conv_wide_and_concat(kind, B, H, cin, cout):
    """Several K chunks, several 64-channel output tiles, GroupNorm groups of 8..32 channels; the two-pointer
    channel concat (model.py:146-148) when cin = 2 * 64; the transposed conv with more than 64 input channels runs one
    output parity per workgroup."""
    torch.manual_seed(kind + B + H + cin)
    x = torch.randn(B, cin, H, H)
    k = 3 if kind == 7 else 4
    w = (torch.randn(cin, cout, k, k) if kind == 9 else torch.randn(cout, cin, k, k)) * (1.0 / (k * cin ** 0.5))
    b = torch.randn(cout)
    if cin == 128 and kind == 7:
        y, st = run_x6(kind, x[:, :64].contiguous(), x[:, 64:].contiguous(), w, b)
    else:
        y, st = run_x6(kind, x, None, w, b)
    ref = ref_conv(kind, x, w, b)
    assert (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    check_stats(st, ref, B)


@pytest.mark.parametrize("kind,H", [(7, 32), (8, 32), (9, 16), (8, 16), (9, 8), (7, 4), (13, 32), (14, 16), (13, 16), (14, 8), (13, 64), (14, 32),
                                    (15, 32), (15, 16), (15, 8), (15, 64)])
def test_x6_conv_fused_groupnorm_silu_temb(kind, H):
    """Input transform SiLU(GroupNorm8(x)) + temb applied while staging (model.py:54-62), from producer partials."""
    torch.manual_seed(kind + H)
    B = 3
    h_raw = torch.randn(B, 64, H, H) * 2.0 + 1.5
    # producer partials: one (mean, M2) per image and group over the whole image (pin = 1)
    hg = h_raw.reshape(B, 8, -1)
    mean = hg.mean(2)
    st0 = torch.stack([mean, ((hg - mean[:, :, None]) ** 2).sum(2)], dim=-1).reshape(B, 1, 8, 2)
    gamma, beta, temb = torch.rand(64) + 0.5, torch.randn(64) * 0.1, torch.randn(1, 64)
    tf = (st0.to(DEV), 1, float(H * H * 8), gamma.to(DEV), beta.to(DEV), temb.to(DEV), 0)
    k = 3 if kind in (7, 15) else 4
    w, b = torch.randn(64, 64, k, k) * 0.05, torch.randn(64)
    y, st = run_x6(kind, h_raw, None, w, b, tf=tf)
    hin = F.silu(F.group_norm(h_raw, 8, gamma, beta, eps=1e-5)) + temb[:, :, None, None]
    ref = ref_conv(kind, hin, w, b)
    assert (y - ref).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())
    check_stats(st, ref, B)


@pytest.mark.parametrize("C,H", [(128, 16), (128, 32), (256, 16)])
def test_fp16_3x3_at_wider_channel_counts(C, H):
    """3x3 in the two-fp16-term form behind a GroupNorm at the widened model's channel counts (conv_x6_c3_kernel: four and eight
    K chunks per tile, groups of 16 and 32 channels) against the fp32 reference, GroupNorm partials included (model.py:54-62)."""
    torch.manual_seed(C + H)
    B = 3
    h_raw = torch.randn(B, C, H, H) * 1.5 - 0.5
    hg = h_raw.reshape(B, 8, -1)
    mean = hg.mean(2)
    st0 = torch.stack([mean, ((hg - mean[:, :, None]) ** 2).sum(2)], dim=-1).reshape(B, 1, 8, 2)
    gamma, beta, temb = torch.rand(C) + 0.5, torch.randn(C) * 0.1, torch.randn(1, C)
    tf = (st0.to(DEV), 1, float(H * H * C // 8), gamma.to(DEV), beta.to(DEV), temb.to(DEV), 0)
    w, b = torch.randn(C, C, 3, 3) * 0.03, torch.randn(C)
    y, st = run_x6(15, h_raw, None, w, b, tf=tf)
    hin = F.silu(F.group_norm(h_raw, 8, gamma, beta, eps=1e-5)) + temb[:, :, None, None]
    ref = ref_conv(15, hin, w, b)
    assert (y - ref).abs().max().item() <= 3e-5 * max(1.0, ref.abs().max().item())
    check_stats(st, ref, B)


def test_x6_split_keeps_fp32_accuracy_on_badly_scaled_inputs():
    """Operands spanning many binades (1e-6 .. 1e4): the three-term split carries 24 mantissa bits whatever the
    magnitude, so the result stays within fp32 rounding of the float64 convolution, relative to sum |x||w|."""
    torch.manual_seed(5)
    B, H = 2, 16
    x = torch.randn(B, 64, H, H) * torch.exp(torch.randn(B, 64, H, H) * 4.0)
    w = torch.randn(64, 64, 3, 3) * 0.05 * torch.exp(torch.randn(64, 64, 3, 3) * 2.0)
    b = torch.zeros(64)
    y, _ = run_x6(7, x, None, w, b, want_stats=False)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    scale = F.conv2d(x.abs().double(), w.abs().double(), None, padding=1)
    assert ((y.double() - ref).abs() / scale).max().item() <= 2e-6


def test_x6_rows_of_a_large_batch_match_a_small_batch():
    """Persistent workgroups walk many tiles (XCD-aware order): rows of a B=600 launch == the same rows alone, bit for bit."""
    torch.manual_seed(9)
    for kind, H in ((7, 16), (8, 32), (9, 16), (13, 32), (14, 16), (15, 32)):
        k = 3 if kind in (7, 15) else 4
        x = torch.randn(600, 64, H, H)
        w, b = torch.randn(64, 64, k, k) * 0.05, torch.randn(64)
        tf = tf2 = None
        if kind >= 13:  # the fp16 form reads SiLU(GroupNorm(.)): unit-variance partials
            st0 = torch.zeros(600, 1, 8, 2)
            st0[..., 1] = H * H * 8
            mk = lambda sx: (sx.to(DEV), 1, float(H * H * 8), torch.ones(64, device=DEV), torch.zeros(64, device=DEV), torch.zeros(1, 64, device=DEV), 0)
            tf, tf2 = mk(st0), mk(st0[592:].contiguous())
        y, st = run_x6(kind, x, None, w, b, tf=tf)
        y2, st2 = run_x6(kind, x[592:].contiguous(), None, w, b, tf=tf2)
        assert torch.equal(y[592:], y2) and torch.equal(st[592:], st2)


def test_fp16_form_needs_a_groupnorm_in_front():
    """kinds 13 / 14 without the fused GroupNorm transform are refused (un-normalised inputs could leave fp16's range)."""
    L = _lib.lib()
    x = torch.zeros(1, 16, 16, 64, device=DEV)
    w = torch.zeros(64, 64, 4, 4, device=DEV)
    wp = torch.empty(L.dm_conv_packed_floats(13, 64, 64), device=DEV)
    _lib.check(L.dm_pack_conv_weight(13, _lib.ptr(w), 64, 64, _lib.ptr(wp), stream()))
    y = torch.empty(1, 8, 8, 64, device=DEV)
    rc = L.dm_conv2d(13, _lib.ptr(x), 64, None, 0, _lib.ptr(wp), _lib.ptr(torch.zeros(64, device=DEV)), 64, None, 0, 0.0, None, None,
                     None, 0, _lib.ptr(y), None, 1, 16, 16, stream())
    assert rc != 0 and b"GroupNorm" in L.dm_last_error()


@pytest.mark.parametrize("scale", [1e-3, 1.0, 30.0])
def test_fp16_form_weight_scaling(scale):
    """The packed weights are scaled by a power of two into fp16's normal range and un-scaled in the epilogue: tiny and large
    weights give the same relative accuracy."""
    torch.manual_seed(3)
    B, H = 2, 32
    h_raw = torch.randn(B, 64, H, H)
    hg = h_raw.reshape(B, 8, -1)
    mean = hg.mean(2)
    st0 = torch.stack([mean, ((hg - mean[:, :, None]) ** 2).sum(2)], dim=-1).reshape(B, 1, 8, 2)
    gamma, beta, temb = torch.rand(64) + 0.5, torch.randn(64) * 0.1, torch.randn(1, 64)
    tf = (st0.to(DEV), 1, float(H * H * 8), gamma.to(DEV), beta.to(DEV), temb.to(DEV), 0)
    hin = F.silu(F.group_norm(h_raw, 8, gamma, beta, eps=1e-5)) + temb[:, :, None, None]
    for kind in (13, 14, 15):
        k = 3 if kind == 15 else 4
        w, b = torch.randn(64, 64, k, k) * 0.05 * scale, torch.randn(64) * scale
        y, _ = run_x6(kind, h_raw, None, w, b, tf=tf, want_stats=False)
        ref = ref_conv(kind, hin.double(), w.double(), b.double())
        assert (y.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_x6_model_forward_matches_fp32_mfma_forward_and_oracle(monkeypatch):
    """UNetMini.forward (model.py:120-151) with the 4x4 convolutions on the bf16 pipe (three-way bf16 split, DM_X6=7; the
    default DM_X6=487 takes two fp16 terms: stride-2 / transposed convs and every conv2 behind a GroupNorm, and all twelve
    layers of the fused low-resolution kernel) against the same weights on the fp32-MFMA
    kernels only (DM_X6=0) and against the CPU oracle: the split forms move eps by less than the fp32 kernels' own distance
    from the oracle."""
    import diffusion_model_amd as d
    from oracle import unet_oracle as O
    d.set_seed(11)
    monkeypatch.setenv("DM_X6", "0")
    m0 = d.UNetMini(image_size=32).eval()
    m0._native()                                          # (the handle reads DM_X6 when it is created)
    monkeypatch.setenv("DM_X6", "7")
    m7 = d.UNetMini(image_size=32).eval()
    m7._native()
    m7.load_state_dict(m0.state_dict())
    monkeypatch.setenv("DM_X6", "487")
    m39 = d.UNetMini(image_size=32).eval()
    m39._native()
    m39.load_state_dict(m0.state_dict())
    x = torch.randn(24, 1, 32, 32)
    t = torch.tensor(437)
    with torch.no_grad():
        e0 = m0(x.to(DEV), t.to(DEV)).cpu()
        e7 = m7(x.to(DEV), t.to(DEV)).cpu()
        e39 = m39(x.to(DEV), t.to(DEV)).cpu()
        ref = O.unet_forward({k: v.detach().cpu() for k, v in m0.state_dict().items()}, x, t)
    assert not torch.equal(e0, e7) and not torch.equal(e7, e39)   # (different kernels did run)
    assert (e0 - e7).abs().max().item() <= 2e-5 and (e0 - e39).abs().max().item() <= 2e-5
    for e in (e0, e7, e39):
        assert (e - ref).abs().max().item() <= 1e-4
    print("eps error vs oracle: fp32-MFMA %.2e, bf16x6 %.2e, f16x3 %.2e" % tuple((e - ref).abs().max().item() for e in (e0, e7, e39)))


@pytest.mark.parametrize("size,cin,ch", [(32, 1, 64), (16, 1, 64), (64, 3, 128)])
def test_init5_composed_first_layers_match_two_layer_form_and_oracle(monkeypatch, size, cin, ch):
    """init_conv (model.py:135) and down1.conv1 (model.py:54) evaluated as ONE 5x5 convolution with border-class weight sets
    (default) against the two-layer evaluation (DM_INIT5=0) and the CPU oracle; an input with large values along the image
    border makes a wrong border class visible."""
    import diffusion_model_amd as d
    from oracle import unet_oracle as O
    d.set_seed(21)
    monkeypatch.setenv("DM_INIT5", "0")
    m0 = d.UNetMini(image_size=size, in_channels=cin, out_channels=cin, model_channels=ch).eval()
    m0._native()
    monkeypatch.setenv("DM_INIT5", "1")
    m1 = d.UNetMini(image_size=size, in_channels=cin, out_channels=cin, model_channels=ch).eval()
    m1._native()
    m1.load_state_dict(m0.state_dict())
    x = torch.randn(5, cin, size, size)
    x[:, :, 0, :] += 3.0; x[:, :, -1, :] -= 2.0; x[:, :, :, 0] += 2.5; x[:, :, :, -1] -= 3.5
    t = torch.tensor(901)
    with torch.no_grad():
        e0 = m0(x.to(DEV), t.to(DEV)).cpu()
        e1 = m1(x.to(DEV), t.to(DEV)).cpu()
        ref = O.unet_forward({k: v.detach().cpu() for k, v in m0.state_dict().items()}, x, t)
    assert not torch.equal(e0, e1)                        # (two different evaluations did run)
    assert (e0 - e1).abs().max().item() <= 2e-5
    assert (e1 - ref).abs().max().item() <= 1e-4 and (e0 - ref).abs().max().item() <= 1e-4


@pytest.mark.parametrize("H,c2,mag", [(32, 0, 1.0), (16, 64, 1.0), (8, 0, 1.0), (16, 64, 1e-4), (16, 0, 3e4), (64, 0, 7.0)])
def test_fp16_form_on_raw_inputs_scaled_by_their_maxima(H, c2, mag):
    """kind 15 without a GroupNorm in front (a block's conv1, model.py:54: its inputs are raw conv outputs): the image is scaled
    by a power of two taken from its producers' maxima before it is split into two fp16 terms, so tiny and large activations
    keep the accuracy of the fp32 kernels.  The maxima come from a producer launch (kind 14 writes them for its output) or,
    here for the skip input, from any partition of the image into partial maxima."""
    L = _lib.lib()
    torch.manual_seed(H + c2)
    B = 3
    x1 = torch.randn(B, 64, H, H) * mag
    x1[1] *= 37.0                                # images of one batch at different scales
    x2 = torch.randn(B, c2, H, H) * mag * 0.1 if c2 else None
    am1 = x1.abs().amax(dim=(1, 3)).contiguous().to(DEV)          # [B][H] partial maxima (rows)
    am2 = None if x2 is None else x2.abs().amax(dim=(1, 2, 3)).reshape(B, 1).contiguous().to(DEV)
    w, b = torch.randn(64, 64 + c2, 3, 3) * 0.05, torch.randn(64) * mag
    L.dm_debug_conv_amax(_lib.ptr(am1), H, _lib.ptr(am2), 1 if c2 else 0, None)
    y, st = run_x6(15, x1, x2, w, b)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = F.conv2d(xin.double(), w.double(), b.double(), padding=1)
    for i in range(B):   # per image: each has its own scale
        assert (y[i].double() - ref[i]).abs().max().item() <= 2e-5 * ref[i].abs().max().item(), i
    if mag == 1.0:
        check_stats(st, ref.float(), B)


def test_x6_kernels_write_the_maxima_of_their_output():
    """amax_out of the stride-2 (pipelined kernel, both tile sizes) and transposed convolutions: the maximum over an image's
    partial maxima equals max |y| of that image, exactly."""
    L = _lib.lib()
    torch.manual_seed(21)
    for kind, B, H in ((13, 3, 32), (13, 300, 32), (14, 3, 16), (8, 2, 32), (9, 2, 16)):
        h_raw = torch.randn(B, 64, H, H)
        hg = h_raw.reshape(B, 8, -1)
        mean = hg.mean(2)
        st0 = torch.stack([mean, ((hg - mean[:, :, None]) ** 2).sum(2)], dim=-1).reshape(B, 1, 8, 2)
        tf = (st0.to(DEV), 1, float(H * H * 8), torch.ones(64, device=DEV), torch.zeros(64, device=DEV), torch.zeros(1, 64, device=DEV), 0)
        w, b = torch.randn(64, 64, 4, 4) * 0.05, torch.randn(64)
        n = L.dm_conv_stats_partials(kind, H, H) * 4
        am = torch.full((B, n), float("nan"), device=DEV)
        L.dm_debug_conv_amax(None, 0, None, 0, _lib.ptr(am))
        y, _ = run_x6(kind, h_raw, None, w, b, tf=tf, want_stats=False)
        assert torch.equal(am.cpu().amax(1), y.abs().amax(dim=(1, 2, 3))), kind
