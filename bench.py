#!/usr/bin/env python3
"""bench.py — sampled images/s of the MI355X-native DDIM-50 sampler (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

A "step" is ONE full DDIM-50 sampling pass (50 denoiser evaluations + 50 sampler updates) over
one batch of B=1024 synthetic 32x32 images per GPU (BASELINE.json configs[1]); x_T and the
weights are resident in HBM when the timed region starts.  N>1 is launched by the driver with
torch.distributed.run, one rank per GPU: every rank samples its own B images (weak scaling,
no data-path collective inside the loop) and the finished samples are all-gathered over
RCCL inside the timed region.  Rank 0 prints ONE JSON line.

Extra legs on rank 0 at N=1 (outside the timed region):
  roofline     — dm_unet_profile: hipEvent pairs around every kernel launch of the denoiser on
                 the launch stream; the dominant kernel class is the 3x3 implicit-GEMM conv.
  cpu_baseline — the CPU oracle (oracle/unet_oracle.py, kind "port") on a bounded sample.
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
K_NAMES = ["init_conv", "conv3x3_mfma_t128", "conv3x3_mfma_t64", "conv4x4s2_mfma", "convT4x4s2_mfma_t128",
           "convT4x4s2_mfma_t64", "final_conv", "time_embed", "sampler_step", "attention", "conv3x3_winograd_mfma"]
NK = len(K_NAMES)  # DM_K_COUNT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="images per GPU per pass")
    ap.add_argument("--ddim-steps", type=int, default=50, help="number of sampler steps (both samplers)")
    ap.add_argument("--sampler", default="ddim", choices=["ddim", "ancestral"],
                    help="ancestral with --ddim-steps 1000 --batch 64 --eta 1 = BASELINE configs[0] (not the default metric)")
    ap.add_argument("--eta", type=float, default=None, help="default 0 for ddim, 1 for ancestral")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="bf16 = BASELINE configs[3] arithmetic (not the default metric)")
    ap.add_argument("--cfg", action="store_true", help="class-conditional + classifier-free guidance (doubled batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
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
    from diffusion_model_amd import _lib

    B, n = args.batch, args.ddim_steps
    d.set_seed(42)
    model = d.UNetMini(image_size=32, precision=args.precision).eval()  # random-init weights, reference architecture
    sched = d.DDPMScheduler()
    eta = args.eta if args.eta is not None else (0.0 if args.sampler == "ddim" else 1.0)
    smp = d.create_sampler(args.sampler, sched, num_inference_steps=n, eta=eta)
    # synthetic x_T: the full [world*B] draw, each rank takes its row slice
    from diffusion_model_amd.distributed import sample_sharded, sampler_fn
    g = torch.Generator().manual_seed(42)
    x_T_full = torch.randn((world * B, 1, 32, 32), generator=g).to(dev)  # same draw on every rank
    x_T = x_T_full[rank * B:(rank + 1) * B]
    fn = sampler_fn(smp, model)
    if eta > 0:  # per-step noise of this rank's rows, resident in HBM like x_T
        nz_local = torch.randn((n, B, 1, 32, 32), generator=torch.Generator().manual_seed(43 + rank)).to(dev)
        fn = lambda x, nz: smp.sample_from(model, x, nz_local)
    if args.cfg:  # build-defined guided sampling: labels arange % 10, null class 10, w = 3
        cc = d.ClassConditioner(11, 128)
        cond_full = cc(torch.arange(world * B) % 10)
        null = cc(torch.tensor([10]))
        fn = lambda x, nz: smp.sample_from(model, x, cond_emb=cond_full[rank * B:(rank + 1) * B], null_emb=null,
                                           guidance_scale=3.0)

    def one_pass():
        # rank r samples rows [r*B, (r+1)*B); one RCCL all-gather of the finished samples
        return sample_sharded(fn, x_T_full)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if force_dist and world == 1:  # the collective sample_sharded issues for N > 1, on a 1-rank communicator
        o1 = one_pass()
        full = torch.empty_like(o1)
        dist.all_gather_into_tensor(full, o1)
        torch.cuda.synchronize()
        assert torch.equal(full, o1)
        print("rehearsal: RCCL all_gather_into_tensor ok", file=sys.stderr)
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
    res = {
        "metric": f"sampled images/sec ({'DDIM' if args.sampler == 'ddim' else 'DDPM-ancestral'}-{n}, 32x32 MNIST UNet)", "value": value, "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
        "config": {"workload": f"{args.sampler.upper()}-{n} eta={eta:g}, UNetMini(32x32, C=64) random-init seed 42, batch {B} per GPU, "
                               "x_T resident in HBM"
                               + (" (BASELINE configs[1])" if (args.sampler, n, eta, B) == ("ddim", 50, 0.0, 1024) else ""),
                   "images_per_gpu": B, "ddim_steps": n, "guidance": bool(args.cfg), "parallelism": f"dp{world} (batch-sharded, final all-gather)"},
        "achieved_tflops_whole_path": value * n * FLOP_PER_IMAGE_STEP / 1e12,
        "frac_fp32_mfma_peak_whole_path": value * n * FLOP_PER_IMAGE_STEP / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world),
    }

    if rank == 0 and not args.no_roofline:  # (any N: rank 0's own kernels; the other ranks wait at the final barrier)
        L = _lib.lib()
        ws = model.workspace(B, 32, 32)
        packed = model.packed_weights()
        t_dev = torch.tensor([500], device=dev)
        coef = sched.step_coefs(torch.tensor([500]), 0.0)[0]
        cbuf = (C.c_float * 5)(*[float(v) for v in coef[:5]])
        counts, ms = (C.c_int * NK)(), (C.c_float * NK)()
        fl, by = (C.c_double * NK)(), (C.c_double * NK)()
        xs = x_T.clone()
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        for reps in (2, 10):  # first call warms up
            _lib.check(L.dm_unet_profile(model._native(), _lib.ptr(packed), _lib.ptr(xs), _lib.ptr(t_dev), cbuf, B, 32, 32,
                                         _lib.ptr(ws), ws.numel(), reps, side.cuda_stream, counts, ms, fl, by))
        per = {}
        tot_ms = sum(ms[k] for k in range(NK))
        for k in range(NK):
            if counts[k]:
                per[K_NAMES[k]] = {"launches": counts[k], "avg_us": 1e3 * ms[k] / counts[k], "share": ms[k] / tot_ms,
                                   "tflops": fl[k] / (ms[k] * 1e-3) / 1e12, "alg_gbps": by[k] / (ms[k] * 1e-3) / 1e9}
        k = 1 if args.precision == "bf16" else 10  # dominant: conv_bf16_kernel<2,0> / conv_mfma_kernel<2,5,64> (DM_K_WINO)
        ach = fl[k] / (ms[k] * 1e-3) / 1e12
        traffic = None  # HBM bytes per launch from the committed PMC passes (profiles/README.md)
        tpath = os.path.join(ROOT, "profiles", "r1_traffic_wino.json")
        if os.path.exists(tpath) and B == 1024:
            traffic = json.load(open(tpath))["hbm_bytes_per_launch"]
        if args.precision == "bf16":
            # at 16x the MFMA rate the bf16 conv kernel is bound by its activation traffic
            gbps = by[k] / (ms[k] * 1e-3) / 1e9
            res["roofline"] = {"kernel": "conv_bf16_kernel<2,0> (3x3 implicit GEMM, v_mfma_f32_32x32x16_bf16)",
                               "bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
                               "traffic": None, "avg_launch_us": 1e3 * ms[k] / counts[k],
                               "algorithmic_bytes_per_launch": by[k] / counts[k], "share_of_denoiser_time": ms[k] / tot_ms}
        else:
          res["roofline"] = {"kernel": "conv_mfma_kernel<2,5,64> (3x3 conv as Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32; "
                                       "achieved = ALGORITHMIC direct-conv FLOPs / time, the kernel executes 4/9 of them)",
                             "executed_mfma_frac": ach * (4.0 / 9.0) / PEAK_FP32_MFMA_TFLOPS,
                             "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                             "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, B=1024)",
                             "algorithmic_bytes_per_launch": by[k] / counts[k],
                             "avg_launch_us": 1e3 * ms[k] / counts[k], "flop_per_launch": fl[k] / counts[k],
                             "share_of_denoiser_time": ms[k] / tot_ms}
        res["kernels"] = per

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.cfg:
        from oracle import unet_oracle as O
        sd = {k_: v.detach().cpu() for k_, v in model.state_dict().items()}
        cb = 512  # ~10 s of CPU work on the 16 host cores of the GPU box
        xc = x_T[:cb].cpu()
        cb = min(cb, B)
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
