#!/usr/bin/env python3
"""bench.py — sampled images/s of the MI355X-native DDIM-50 sampler (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--total-images T]

A "step" is ONE full DDIM-50 sampling pass (50 denoiser evaluations + 50 sampler updates) over
one batch of B=1024 synthetic 32x32 images per GPU (BASELINE.json configs[1]); x_T and the
weights are resident in HBM when the timed region starts.  N>1 is launched by the driver with
torch.distributed.run, one rank per GPU: every rank samples its own B images (weak scaling,
no data-path collective inside the loop) and the finished samples are all-gathered over
RCCL inside the timed region.  `--total-images T` instead fixes the whole job at T images
(BASELINE configs[2]: 65536) and gives every rank T/N of them (strong scaling).
Rank 0 prints ONE JSON line.

Extra legs on rank 0 at N=1 (outside the timed region):
  roofline     — dm_unet_profile: hipEvent pairs around every kernel launch of the denoiser on
                 the launch stream; the dominant kernel class is the Winograd 3x3 conv.
                 `frac` is the EXECUTED MFMA fraction of the fp32-MFMA peak (the Winograd form
                 executes 4/9 of the direct-conv FLOPs); `algorithmic_frac` prices the
                 algorithmic direct-conv FLOPs and may exceed 1.
  end_to_end   — the same pass with the x_T upload (H2D) and the sample download (D2H) inside the
                 timed region, next to the resident number (`value`).
  configs      — short runs of the other BASELINE.json configurations (cfg1 ancestral-1000 B=64,
                 cfg3's per-GPU shard B=8192, cfg4 bf16 + guidance, cfg5 64x64x3 C=256 with and
                 without attention), each with images/s, its dominant kernel class with the
                 executed fraction, and a CPU-oracle check on a small slice.
  cpu_baseline — the CPU oracle (oracle/unet_oracle.py, kind "port") on a bounded sample; its
                 output also checks the GPU samples.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMAGE_STEP = 364_249_088  # SURVEY 8(d), equals torch.utils.flop_counter on model.py:120-151
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA peak
PEAK_HBM_GBPS = 8000.0
K_NAMES = ["init_conv", "conv3x3_mfma_t128", "conv3x3_mfma_t64", "conv4x4s2_mfma", "convT4x4s2_mfma_t128",
           "convT4x4s2_mfma_t64", "final_conv", "time_embed", "sampler_step", "attention", "conv3x3_winograd_mfma",
           "conv3x3_split16_mfma", "conv4x4s2_split16_mfma", "convT4x4s2_split16_mfma", "conv_any_geometry_mfma", "groupnorm_silu_any_geometry",
           "lowres_fused_split16_mfma"]
NK = len(K_NAMES)  # DM_K_COUNT
K_WINO = 10
MFMA_CLASSES = (1, 2, 3, 4, 5, 9, 10, 11, 12, 13)
MFMA_CLASSES = MFMA_CLASSES + (14,)
# fp32 convolutions on the 16-bit matrix pipe, operands split into 16-bit terms: class -> the DM_X6 bit that selects the two-fp16-term
# form (THREE v_mfma_f32_32x32x16_f16 per fp32 product; default) in place of the three-bf16-term form (SIX bf16 MFMAs)
X6_CLASSES = {11: 64, 12: 32, 13: 32, 16: 128}
DM_X6 = int(os.environ.get("DM_X6", "487"))


def split16_mfmas_per_product(k):
    return 3 if (DM_X6 & 16) or (DM_X6 & X6_CLASSES[k]) else 6
# PMC / rocprofv3 summaries of THIS build (profiles/README.md): each file carries the key of the library it was measured on
# (`build_key`, __graft_entry__.build_key()); a figure whose key differs from the loaded library's is reported as null
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r3_traffic.json")
ROCPROF_FILE = os.path.join(ROOT, "profiles", "r3_rocprof_summary.json")


def keyed_profile(path):
    """A committed profile summary, or None when it was measured on another build of the library."""
    from diffusion_model_amd import _lib
    if not os.path.exists(path):
        return None
    j = json.load(open(path))
    return j if j.get("build_key") == _lib.lib().dm_build_key().decode() else None


def flop_per_image_step(C_, S, cin, cout):
    """SURVEY 8(a) closed form: MACs/img = 43.265625 C^2 S + 9 Cin C S + C Cout S."""
    return 2.0 * (43.265625 * C_ * C_ * S + 9.0 * cin * C_ * S + C_ * cout * S)


def profile_model(model, sched, B, H, W, reps=3):
    """Per-kernel-class device time of `reps` denoiser evaluations (dm_unet_profile)."""
    from diffusion_model_amd import _lib
    L = _lib.lib()
    dev = next(model.parameters()).device
    ws = model.workspace(B, H, W)
    packed = model.packed_weights()
    t_dev = torch.tensor([500], device=dev)
    coef = sched.step_coefs(torch.tensor([500]), 0.0)[0]
    cbuf = (C.c_float * 5)(*[float(v) for v in coef[:5]])
    counts, ms = (C.c_int * NK)(), (C.c_float * NK)()
    fl, by = (C.c_double * NK)(), (C.c_double * NK)()
    xs = torch.randn((B, model.in_channels, H, W), generator=torch.Generator().manual_seed(1)).to(dev)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for r in (1, reps):  # first call warms up
        _lib.check(L.dm_unet_profile(model._native(), _lib.ptr(packed), _lib.ptr(xs), _lib.ptr(t_dev), cbuf, B, H, W,
                                     _lib.ptr(ws), ws.numel(), r, side.cuda_stream, NK, counts, ms, fl, by))
    bf16 = model.precision == "bf16"
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
    tot_ms = sum(ms[k] for k in range(NK))
    per = {}
    for k in range(NK):
        if counts[k]:
            tf = fl[k] / (ms[k] * 1e-3) / 1e12
            e = {"launches": counts[k] // reps, "avg_us": 1e3 * ms[k] / counts[k], "share": ms[k] / tot_ms, "tflops": tf,
                 "alg_gbps": by[k] / (ms[k] * 1e-3) / 1e9}
            if k in X6_CLASSES and not bf16:  # executed 16-bit MFMA FLOPs = 3 (6) x the algorithmic fp32 FLOPs, priced on the dense bf16 / fp16 peak
                nm = split16_mfmas_per_product(k)
                e["executed_mfma_frac"] = tf * nm / PEAK_BF16_MFMA_TFLOPS
                e["pipe"] = ("fp16 MFMA, operands as two fp16 terms, three MFMAs per fp32 product" if nm == 3 else
                             "bf16 MFMA, operands as three bf16 terms, six MFMAs per fp32 product") + " (fp32-equivalent accuracy)"
            elif k in MFMA_CLASSES:  # executed MFMA fraction (the Winograd class executes 4/9 of its algorithmic FLOPs)
                e["executed_mfma_frac"] = tf * (4.0 / 9.0 if k == K_WINO else 1.0) / peak
            per[K_NAMES[k]] = e
    raw = {"counts": counts, "ms": ms, "fl": fl, "by": by, "tot_ms": tot_ms}
    return per, raw


def dominant(per):
    name = max(per, key=lambda k_: per[k_]["share"])
    e = per[name]
    return {"kernel": name, "share": e["share"], "avg_us": e["avg_us"], "tflops": e["tflops"],
            "executed_mfma_frac": e.get("executed_mfma_frac"), "alg_gbps": e["alg_gbps"]}


def timed_passes(fn, passes, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / passes, out


def run_configs(d, O, dev, sd32):
    """Short, bounded runs of BASELINE.json's other configurations on ONE GPU."""
    res = {}
    sched = d.DDPMScheduler()
    osch = O.make_schedule()

    # ---- cfg1: DDPM ancestral T=1000, batch 64, eta=1, fp32 (configs[0])
    m32 = d.UNetMini(image_size=32).eval()
    m32.load_state_dict(sd32)
    B, n = 64, 1000
    smp = d.AncestralSampler(sched, num_inference_steps=n, eta=1.0)
    g = torch.Generator().manual_seed(11)
    x_T = torch.randn((B, 1, 32, 32), generator=g)
    nz = torch.randn((n, B, 1, 32, 32), generator=g).pin_memory()  # host noise, staged 50 steps at a time
    xd = x_T.to(dev)
    dt, _ = timed_passes(lambda: smp.sample_from(m32, xd, nz), 2)
    per, _ = profile_model(m32, sched, B, 32, 32)
    # oracle check on the first 100 of the 1000 steps, 4 images (the free-running 1000-step chain on random weights
    # grows to |x| ~ 800 and is ill-conditioned: SURVEY 7)
    k, ns = 4, 100
    full_ts = sched.timesteps
    sched.timesteps = full_ts[:ns]
    got = smp.sample_from(m32, x_T[:k].contiguous(), nz[:ns, :k].contiguous()).cpu()
    sched.timesteps = full_ts
    xr = x_T[:k].clone()
    with torch.no_grad():
        for i in range(ns):
            t = torch.tensor(i)
            xr = O.sampler_step(osch, O.unet_forward(sd32, xr, t), t, xr, 1.0, nz[i, :k])
    # the public entry point, AncestralSampler.__call__ (sampler.py:45-76): x_T and ALL per-step noise drawn from torch's CPU
    # generator inside the timed region - chunk by chunk in the library's chunk callback, so the draw of chunk k+1 overlaps the
    # GPU's work on chunk k - plus the H2D copies.  torch.randn itself (the reference pays it too) bounds this from above.
    torch.manual_seed(11)
    dt_call, _ = timed_passes(lambda: smp(m32, (B, 1, 32, 32), progress_bar=False), 2)
    t0 = time.perf_counter()
    torch.randn((n, B, 1, 32, 32))
    rng_s = time.perf_counter() - t0
    res["cfg1_ancestral1000_b64_eta1_fp32"] = {
        "images_per_s": B / dt, "ms_per_pass": 1e3 * dt, "batch": B, "steps": n, "noise": "host tensor, staged per 50-step graph chunk",
        "end_to_end_call": {"images_per_s": B / dt_call, "ms_per_pass": 1e3 * dt_call,
                            "includes": "AncestralSampler.__call__: x_T draw, the 1000 per-step noise draws (torch CPU generator, one 50-step chunk at a "
                                        "time into pinned buffers while the GPU runs the previous chunk), H2D, sampling",
                            "host_rng_ms_per_pass": 1e3 * rng_s, "host_rng_ceiling_images_per_s": B / rng_s,
                            "note": "torch.randn of the n x B x 1024 noise values alone takes host_rng_ms_per_pass on this host (single-threaded "
                                    "mt19937 + Box-Muller, bit-identical to the reference's draws): no sampler that keeps the reference's noise can "
                                    "run faster than host_rng_ceiling_images_per_s; `images_per_s` above is the same loop with the noise pre-drawn"},
        "frac_fp32_mfma_peak_whole_path_algorithmic": B / dt * n * FLOP_PER_IMAGE_STEP / 1e12 / PEAK_FP32_MFMA_TFLOPS,
        "dominant": dominant(per),
        "check_max_abs_vs_cpu_oracle": (got - xr).abs().max().item(), "check": f"first {ns} steps, {k} images, |ref|max {xr.abs().max().item():.3g}"}
    del nz

    # ---- cfg3 shard: DDIM-50, 8192 images on one GPU (= 65536 / 8), fp32 (configs[2])
    B, n = 8192, 50
    smp = d.DDIMSampler(sched, num_inference_steps=n, eta=0.0)
    x_T = torch.randn((B, 1, 32, 32), generator=torch.Generator().manual_seed(12))
    xd = x_T.to(dev)
    dt, out = timed_passes(lambda: smp.sample_from(m32, xd), 2)
    k = 8  # the LAST images of the shard: every persistent kernel has walked all its tiles by then
    ref = O.sample(sd32, osch, (k, 1, 32, 32), n, 0.0, x_T=x_T[-k:])
    res["cfg3_shard_ddim50_b8192_fp32"] = {
        "images_per_s": B / dt, "ms_per_pass": 1e3 * dt, "batch": B, "steps": n,
        "frac_fp32_mfma_peak_whole_path_algorithmic": B / dt * n * FLOP_PER_IMAGE_STEP / 1e12 / PEAK_FP32_MFMA_TFLOPS,
        "check_max_abs_vs_cpu_oracle": (out[-k:].cpu() - ref).abs().max().item(), "check": f"last {k} images of the shard, all {n} steps",
        "note": "strong-scaling form: python -m torch.distributed.run ... bench.py --gpus N --total-images 65536"}
    del out, xd
    m32._ws = None
    torch.cuda.empty_cache()

    # ---- configs[1] again with the OPT-IN two-term operand split of the 4x4 convolutions (DM_X6 bit 4: three bf16 MFMAs per
    # product instead of six; ~7x the rounding error of an fp32 dot product in those layers).  NOT the default and not `value`.
    if "DM_X6" not in os.environ:
        os.environ["DM_X6"] = "23"
        try:
            m2t = d.UNetMini(image_size=32).eval()
            m2t._native()  # (the handle reads DM_X6 when it is created)
        finally:
            del os.environ["DM_X6"]
        m2t.load_state_dict(sd32)
        B, n = 1024, 50
        x_T = torch.randn((B, 1, 32, 32), generator=torch.Generator().manual_seed(15))
        xd = x_T.to(dev)
        dt, out = timed_passes(lambda: smp.sample_from(m2t, xd), 3)
        k = 8
        ref = O.sample(sd32, osch, (k, 1, 32, 32), n, 0.0, x_T=x_T[-k:])
        dflt = smp.sample_from(m32, xd)
        res["cfg2_optin_two_term_split_fp32"] = {
            "images_per_s": B / dt, "ms_per_pass": 1e3 * dt, "batch": B, "steps": n,
            "check_max_abs_vs_cpu_oracle": (out[-k:].cpu() - ref).abs().max().item(),
            "default_build_same_inputs_check_max_abs_vs_cpu_oracle": (dflt[-k:].cpu() - ref).abs().max().item(),
            "check": f"last {k} images, all {n} steps (tolerance 1e-4 on eps per step; PSNR >= 40 dB on the samples)",
            "note": "opt-in (DM_X6=23): operands of the 4x4 convolutions as two bf16 terms, 3 MFMAs per product; the default (`value`) uses "
                    "three terms and six MFMAs, whose error is at the level of fp32 rounding"}
        del out, dflt, xd, m2t
        torch.cuda.empty_cache()

    # ---- cfg4: class-conditional + classifier-free guidance (doubled batch), bf16 (configs[3]; build-defined semantics)
    B, n, w = 1024, 50, 3.0
    mb = d.UNetMini(image_size=32, precision="bf16").eval()
    mb.load_state_dict(sd32)
    d.set_seed(43)
    cc = d.ClassConditioner(11, 128)
    labels = torch.arange(B) % 10
    cond, null = cc(labels), cc(torch.tensor([10]))
    x_T = torch.randn((B, 1, 32, 32), generator=torch.Generator().manual_seed(13))
    xd = x_T.to(dev)
    smp = d.DDIMSampler(sched, num_inference_steps=n, eta=0.0)
    dt, out = timed_passes(lambda: smp.sample_from(mb, xd, cond_emb=cond, null_emb=null, guidance_scale=w), 3)
    per, _ = profile_model(mb, sched, 2 * B, 32, 32)
    k = 4
    cd = {k_: v.detach().cpu() for k_, v in cc.state_dict().items()}
    ref = O.cfg_sample(sd32, cd, osch, x_T[:k], labels[:k], 10, w, n)
    res["cfg4_classcond_cfg_bf16_b1024"] = {
        "images_per_s": B / dt, "ms_per_pass": 1e3 * dt, "batch": B, "denoiser_batch": 2 * B, "steps": n, "guidance": w,
        "dominant": dominant(per), "kernels": {k_: {"share": v["share"], "avg_us": v["avg_us"], "alg_gbps": v["alg_gbps"]} for k_, v in per.items()},
        "check_max_abs_vs_cpu_oracle": (out[:k].cpu() - ref).abs().max().item(),
        "check": f"first {k} images, all {n} steps, fp32 CPU oracle (bf16 tolerance: per-step eps 2e-2)"}
    del mb, out

    # ---- a geometry outside the tuned kernels' set (the reference accepts any H, W % 8 == 0 and any C % 8 == 0, model.py:72-73):
    # 40x40x3 images, C = 96, on the any-geometry kernels (dm_kernels_generic.hip)
    B, n = 256, 50
    d.set_seed(9)
    mg = d.UNetMini(image_size=40, in_channels=3, out_channels=3, model_channels=96).eval()
    sdg = {k_: v.detach().cpu() for k_, v in mg.state_dict().items()}
    x_T = torch.randn((B, 3, 40, 40), generator=torch.Generator().manual_seed(16))
    xd = x_T.to(dev)
    smp = d.DDIMSampler(sched, num_inference_steps=n, eta=0.0)
    dt, out = timed_passes(lambda: smp.sample_from(mg, xd), 1)
    per, _ = profile_model(mg, sched, B, 40, 40, reps=2)
    fl = flop_per_image_step(96, 1600, 3, 3)
    with torch.no_grad():
        ref = O.unet_forward(sdg, x_T[-2:], torch.tensor(500))
    got = mg(xd[-2:].contiguous(), torch.tensor(500)).cpu()
    res["any_geometry_40x40x3_c96"] = {
        "images_per_s": B / dt, "ms_per_pass": 1e3 * dt, "batch": B, "steps": n,
        "frac_fp32_mfma_peak_whole_path_algorithmic": B / dt * n * fl / 1e12 / PEAK_FP32_MFMA_TFLOPS,
        "dominant": dominant(per), "kernels": {k_: {"share": v["share"], "avg_us": v["avg_us"]} for k_, v in per.items()},
        "check_max_abs_vs_cpu_oracle": (got - ref).abs().max().item(), "check": "one denoiser evaluation (t=500) of the last 2 images",
        "note": "not a BASELINE configuration: image sides that are not powers of two / channel counts outside {64,128,256,512} run on "
                "plain fp32-MFMA kernels with stand-alone GroupNorm passes"}
    del mg, out, xd
    torch.cuda.empty_cache()

    # ---- cfg5: widened UNet 64x64x3, C=256, attention at 16x16 on / off, fp32 (configs[4]; attention build-defined)
    B, n = 256, 50
    for attn in (False, True):
        d.set_seed(7)
        mw = d.UNetMini(image_size=64, in_channels=3, out_channels=3, model_channels=256, attention=attn).eval()
        sdw = {k_: v.detach().cpu() for k_, v in mw.state_dict().items()}
        x_T = torch.randn((B, 3, 64, 64), generator=torch.Generator().manual_seed(14))
        xd = x_T.to(dev)
        dt, out = timed_passes(lambda: smp.sample_from(mw, xd), 1)
        per, _ = profile_model(mw, sched, B, 64, 64, reps=2)
        fl = flop_per_image_step(256, 4096, 3, 3)
        with torch.no_grad():
            ref = O.unet_forward(sdw, x_T[-2:], torch.tensor(500))
        got = mw(xd[-2:].contiguous(), torch.tensor(500)).cpu()
        res["cfg5_wide64x64x3_c256" + ("_attention" if attn else "")] = {
            "images_per_s": B / dt, "ms_per_pass": 1e3 * dt, "batch": B, "steps": n,
            "frac_fp32_mfma_peak_whole_path_algorithmic": B / dt * n * fl / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "dominant": dominant(per),
            "attention": ({"share": per["attention"]["share"], "avg_us": per["attention"]["avg_us"], "tflops": per["attention"]["tflops"],
                           "executed_mfma_frac": per["attention"]["executed_mfma_frac"]} if attn else None),
            "check_max_abs_vs_cpu_oracle": (got - ref).abs().max().item(), "check": "one denoiser evaluation (t=500) of the last 2 images"}
        del mw, out, xd
        torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="images per GPU per pass")
    ap.add_argument("--total-images", type=int, default=0,
                    help="strong scaling: the whole job is this many images per pass, T/N per GPU (BASELINE configs[2]: 65536)")
    ap.add_argument("--ddim-steps", type=int, default=50, help="number of sampler steps (both samplers)")
    ap.add_argument("--sampler", default="ddim", choices=["ddim", "ancestral"],
                    help="ancestral with --ddim-steps 1000 --batch 64 --eta 1 = BASELINE configs[0] (not the default metric)")
    ap.add_argument("--eta", type=float, default=None, help="default 0 for ddim, 1 for ancestral")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="bf16 = BASELINE configs[3] arithmetic (not the default metric)")
    ap.add_argument("--cfg", action="store_true", help="class-conditional + classifier-free guidance (doubled batch)")
    ap.add_argument("--model", default="mnist", choices=["mnist", "wide", "wide-attn"],
                    help="wide = BASELINE configs[4]: 64x64x3, C=256 (use --batch 256); wide-attn adds the build-defined attention blocks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the bounded runs of the other BASELINE configurations")
    args = ap.parse_args()

    if os.environ.get("DM_BENCH_WATCHDOG"):  # rehearsal aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DM_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    # rehearsal aid: DM_BENCH_BACKEND=gloo DM_BENCH_ONE_DEVICE=1 runs N ranks on ONE GPU (RCCL refuses
    # two ranks on a device); the driver's real runs use the defaults (nccl = RCCL, one GPU per rank)
    backend = os.environ.get("DM_BENCH_BACKEND", "nccl")
    if os.environ.get("DM_BENCH_ONE_DEVICE"):
        local_rank = 0
    # plumbing rehearsal WITHOUT a GPU (tests/test_distributed_cpu.py): gloo, host tensors, and a stand-in for the sampler
    # (x -> -x: no denoiser runs, `value` means nothing and the line says so) - exercises the launch contract, the host-side
    # shard slicing, the gather and the JSON line of the N > 1 path
    rehearse = bool(os.environ.get("DM_BENCH_REHEARSE_CPU"))
    if rehearse:
        backend = "gloo"
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist = None
    force_dist = bool(os.environ.get("DM_BENCH_FORCE_DIST"))  # rehearsal: a 1-rank RCCL communicator + all-gather
    if force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    saved_stdout = None
    if world > 1 or force_dist:
        # RCCL prints a version banner on stdout when its communicator is created (NCCL_DEBUG=VERSION on some
        # boxes): keep stdout for the ONE JSON line by pointing fd 1 at stderr until that line is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import diffusion_model_amd as d

    strong = args.total_images > 0
    if strong and args.total_images % world:
        raise SystemExit(f"--total-images {args.total_images} is not divisible by {world} GPUs")
    B, n = (args.total_images // world if strong else args.batch), args.ddim_steps
    d.set_seed(42)
    IMG, CIN, CH = (32, 1, 64) if args.model == "mnist" else (64, 3, 256)
    flop_img_step = FLOP_PER_IMAGE_STEP if args.model == "mnist" else flop_per_image_step(CH, IMG * IMG, CIN, CIN)
    sched = d.DDPMScheduler()
    eta = args.eta if args.eta is not None else (0.0 if args.sampler == "ddim" else 1.0)
    smp = d.create_sampler(args.sampler, sched, num_inference_steps=n, eta=eta)
    # synthetic x_T: the full [world*B] draw (the same on every rank), sliced ON THE HOST: each rank uploads its own rows only
    # (4 KiB per image - at 65,536 images the full tensor is 256 MiB per rank, the shard 32 MiB at N = 8)
    from diffusion_model_amd.distributed import gather_shards, shard_rows, sampler_fn
    g = torch.Generator().manual_seed(42)
    x_T_host = torch.randn((world * B, CIN, IMG, IMG), generator=g)
    x_T = shard_rows(x_T_host).to(dev)
    assert x_T.shape[0] == B
    model = None
    if rehearse:
        fn = lambda x, nz: x.neg()
    else:
        model = d.UNetMini(image_size=IMG, in_channels=CIN, out_channels=CIN, model_channels=CH, precision=args.precision,
                           attention=args.model == "wide-attn").eval()  # random-init weights, reference architecture
        fn = sampler_fn(smp, model)
    nz_local = None
    if eta > 0 and not rehearse:  # per-step noise of this rank's rows, resident in HBM like x_T
        nz_local = torch.randn((n, B, CIN, IMG, IMG), generator=torch.Generator().manual_seed(43 + rank)).to(dev)
        fn = lambda x, nz: smp.sample_from(model, x, nz_local)
    if args.cfg and not rehearse:  # build-defined guided sampling: labels arange % 10, null class 10, w = 3
        cc = d.ClassConditioner(11, 128)
        cond_full = cc(torch.arange(world * B) % 10)
        null = cc(torch.tensor([10]))
        fn = lambda x, nz: smp.sample_from(model, x, cond_emb=cond_full[rank * B:(rank + 1) * B], null_emb=null,
                                           guidance_scale=3.0)

    def one_pass():
        # rank r samples rows [r*B, (r+1)*B) (resident in HBM); one RCCL all-gather of the finished samples into a
        # buffer that is allocated once
        return gather_shards(fn(x_T, None))

    def barrier():
        if world > 1:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    if force_dist and world == 1:  # the collective sample_sharded issues for N > 1, on a 1-rank communicator
        o1 = one_pass()
        full = torch.empty_like(o1)
        dist.all_gather_into_tensor(full, o1)
        torch.cuda.synchronize()
        assert torch.equal(full, o1)
        print("rehearsal: RCCL all_gather_into_tensor ok", file=sys.stderr)
    if rehearse:
        assert torch.equal(one_pass(), x_T_host.neg())  # every rank's rows landed where the unsharded job has them
    for _ in range(args.warmup):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_pass()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    assert torch.isfinite(out).all() and out.shape[0] == world * B
    out = out[rank * B:(rank + 1) * B]
    images = world * B * args.steps
    value = images / dt
    default_cfg = (args.sampler, n, eta, B, args.precision, args.cfg, args.model) == ("ddim", 50, 0.0, 1024, "fp32", False, "mnist")
    res = {
        "metric": f"sampled images/sec ({'DDIM' if args.sampler == 'ddim' else 'DDPM-ancestral'}-{n}, {IMG}x{IMG} {'MNIST' if args.model == 'mnist' else 'widened'} UNet)", "value": value, "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
        "config": {"workload": f"{args.sampler.upper()}-{n} eta={eta:g}, UNetMini({IMG}x{IMG}x{CIN}, C={CH}{', attention' if args.model == 'wide-attn' else ''}) random-init seed 42, batch {B} per GPU, "
                               "x_T resident in HBM"
                               + (" (BASELINE configs[1])" if default_cfg and not strong else "")
                               + (f" (BASELINE configs[2]: {args.total_images} images sharded over {world} GPUs)" if strong else ""),
                   "images_per_gpu": B, "ddim_steps": n, "guidance": bool(args.cfg), "parallelism": f"dp{world} (batch-sharded, final all-gather)"},
        # ALGORITHMIC direct-conv FLOPs per second, and their ratio to the fp32-MFMA peak: the Winograd layers
        # execute 4/9 of them and the 4x4 convolutions run on the bf16 pipe, so the ratio may exceed 1 - the executed
        # fractions per pipe are in `roofline` and `kernels`
        "achieved_tflops_whole_path_algorithmic": value * n * flop_img_step / 1e12,
        "frac_fp32_mfma_peak_whole_path_algorithmic": value * n * flop_img_step / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world),
    }
    from diffusion_model_amd import _lib as _l
    res["build_key"] = _l.lib().dm_build_key().decode()  # sha256 of the library's sources + flags (profiles/ summaries are keyed by it)
    if rehearse:
        res["rehearsal"] = "CPU plumbing rehearsal (DM_BENCH_REHEARSE_CPU): stand-in sampler, `value` is not a measurement"
        args.no_roofline = args.no_cpu_baseline = args.no_configs = True

    if model is not None:
        # dm_sample runs large batches as parallel sub-batch chains on branches of the captured graph:
        # a scheduling decision, bit-identical samples; `single_chain` is the same pass with DM_SPLIT=1 beside it
        nb = _l.lib().dm_sample_branches(model._native(), B, IMG, IMG, 1 if args.cfg else 0)
        res["config"]["parallel_branches"] = nb
        if rank == 0 and world == 1 and nb > 1:
            prev = os.environ.get("DM_SPLIT")
            os.environ["DM_SPLIT"] = "1"
            try:
                dts, _ = timed_passes(lambda: fn(x_T, None), max(2, args.steps), warm=1)
            finally:
                if prev is None:
                    del os.environ["DM_SPLIT"]
                else:
                    os.environ["DM_SPLIT"] = prev
            res["single_chain"] = {"value": B / dts, "unit": "images/s", "ms_per_step": 1e3 * dts,
                                   "note": f"the same pass as ONE chain of {B} images (DM_SPLIT=1); `value` runs {nb} chains of {B // nb} images on "
                                           "parallel branches of the graph. `roofline` / `kernels` time single-chain launches of the whole batch"}

    if rank == 0 and world == 1 and not rehearse:
        # end to end: x_T uploaded from (pinned) host memory and the finished samples downloaded inside the timed region
        xh = x_T_host.pin_memory()
        oh = torch.empty_like(xh).pin_memory()

        def e2e_pass():
            o = fn(xh.to(dev, non_blocking=True), None)
            oh.copy_(o, non_blocking=True)
            return o
        dte, _ = timed_passes(e2e_pass, max(2, args.steps), warm=1)
        res["end_to_end"] = {"value": B / dte, "unit": "images/s", "ms_per_step": 1e3 * dte,
                             "includes": "x_T H2D (pinned host, 4 KiB/image) + sampling + D2H of the samples; `value` above is the HBM-resident rate"}

    if rank == 0 and not args.no_roofline:  # (any N: rank 0's own kernels; the other ranks wait at the final barrier)
        per, raw = profile_model(model, sched, B, IMG, IMG, reps=10 if args.model == "mnist" else 3)
        counts, ms, fl, by, tot_ms = raw["counts"], raw["ms"], raw["fl"], raw["by"], raw["tot_ms"]
        # dominant kernel class of the step (largest share of the denoiser's device time)
        k = 1 if args.precision == "bf16" else max((q for q in range(NK) if counts[q] and q in MFMA_CLASSES), key=lambda q: ms[q])
        ach = fl[k] / (ms[k] * 1e-3) / 1e12
        # HBM bytes per launch from the committed PMC passes and the rocprofv3 class average, both only if measured on THIS build
        traffic = rocprof_avg = None
        if B == 1024 and args.precision == "fp32" and args.model == "mnist":
            tj, rj = keyed_profile(TRAFFIC_FILE), keyed_profile(ROCPROF_FILE)
            traffic = tj.get("classes", {}).get(K_NAMES[k], {}).get("hbm_bytes_per_launch") if tj else None
            rocprof_avg = rj.get("classes", {}).get(K_NAMES[k], {}).get("avg_us") if rj else None
        if args.precision == "bf16":
            # at 16x the MFMA rate the bf16 conv kernel is bound by its activation traffic
            gbps = by[k] / (ms[k] * 1e-3) / 1e9
            res["roofline"] = {"kernel": "conv_bf16_kernel<2,0> (3x3 implicit GEMM, v_mfma_f32_32x32x16_bf16)",
                               "bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                               "traffic": None, "avg_launch_us": 1e3 * ms[k] / counts[k],
                               "algorithmic_bytes_per_launch": by[k] / counts[k], "share_of_denoiser_time": ms[k] / tot_ms}
        else:
            if k in X6_CLASSES:
                nm = split16_mfmas_per_product(k)
                exe, peak = ach * nm, PEAK_BF16_MFMA_TFLOPS
                kname = {11: "conv_x6_kernel<2,0,true,2,true> / conv_x6_c3_kernel (3x3 conv, direct form", 12: "conv_x6_down_kernel (4x4 stride-2 conv",
                         13: "conv_x6_kernel<2,4> (transposed 4x4 conv", 16: "lowres_fused_kernel (twelve layers at <= 8x8"}[k] + \
                    (", fp32 operands as two fp16 terms on v_mfma_f32_32x32x16_f16)" if nm == 3 else ", fp32 operands as three bf16 terms on v_mfma_f32_32x32x16_bf16)")
                note = (f"achieved/frac = EXECUTED 16-bit MFMA FLOPs ({nm} MFMAs per fp32 product) against the dense bf16/fp16 MFMA peak; "
                        "algorithmic_* price the direct-conv fp32 FLOPs of SURVEY 8(d) against the fp32-MFMA peak and may exceed it")
            else:
                wino = k == K_WINO
                exe, peak = ach * (4.0 / 9.0 if wino else 1.0), PEAK_FP32_MFMA_TFLOPS
                kname = ("conv_mfma_kernel<2,5,64> (3x3 conv as Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32)" if wino else K_NAMES[k] + " (v_mfma_f32_32x32x2_f32)")
                note = ("achieved/frac = EXECUTED MFMA FLOPs (the Winograd form executes 4/9 of the direct-conv FLOPs); "
                        "algorithmic_* price the direct-conv FLOPs of SURVEY 8(d) and may exceed the peak")
            res["roofline"] = {"kernel": kname, "class": K_NAMES[k],
                               "bound": "mfma", "achieved": exe, "peak": peak, "unit": "TFLOP/s",
                               "frac": exe / peak, "note": note,
                               "algorithmic_achieved": ach, "algorithmic_frac": ach / PEAK_FP32_MFMA_TFLOPS,
                               "traffic": traffic,
                               "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, B=1024)",
                               "algorithmic_bytes_per_launch": by[k] / counts[k],
                               "avg_launch_us": 1e3 * ms[k] / counts[k], "flop_per_launch": fl[k] / counts[k],
                               "rocprof_avg_launch_us": rocprof_avg,
                               "rocprof_note": "rocprofv3 --kernel-trace --stats class average of the same command on this build "
                                               "(profiles/r3_rocprof_summary.json); null when the committed profile is of another build",
                               "launches_per_evaluation": counts[k] // (10 if args.model == "mnist" else 3),
                               "share_of_denoiser_time": ms[k] / tot_ms}
        res["kernels"] = per

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.cfg:
        from oracle import unet_oracle as O
        sd = {k_: v.detach().cpu() for k_, v in model.state_dict().items()}
        cb = min(512 if args.model == "mnist" else 4, B)  # ~10 s of CPU work on the 16 host cores of the GPU box
        xc = x_T[:cb].cpu()
        nzc = nz_local[:, :cb].cpu() if eta > 0 else None
        O.sample(sd, O.make_schedule(), xc.shape, 2, 0.0, x_T=xc)  # warm-up
        t0 = time.perf_counter()
        ref = O.sample(sd, O.make_schedule(), xc.shape, n, eta, x_T=xc, noises=nzc)
        ct = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": cb / ct, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"CPU oracle (torch-CPU restatement of the reference sampler), {args.sampler}-{n} eta={eta:g}, batch {cb}, "
                                         f"{ct:.1f} s"}
        err = (out[:cb].cpu() - ref).abs().max().item()
        res["check_max_abs_vs_cpu_oracle"] = err
        if eta == 0 and n <= 50 and args.precision == "fp32" and not err <= 1e-3:
            # a throughput number for wrong samples is worthless: refuse to print one
            raise SystemExit(f"bench: GPU samples differ from the CPU oracle by {err:.3g} (> 1e-3) on the first {cb} images")

    if rank == 0 and world == 1 and default_cfg and not strong and not args.no_configs:
        from oracle import unet_oracle as O
        sd32 = {k_: v.detach().cpu() for k_, v in model.state_dict().items()}
        del out
        model._ws = None
        torch.cuda.empty_cache()
        res["configs"] = run_configs(d, O, dev, sd32)

    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        if world > 1:
            dist.barrier()  # rank 0 may still be in its roofline leg
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
