# ancestral-1000, B=64, eta=1 (BASELINE configs[0]) with the per-step noise in pinned / pageable host memory or on the device:
# images/s per library build (DM_SAMPLER_LIB), alternating.  usage: cfg1_ab.py libA.so:libB.so
import os, sys, subprocess, json, time
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, '.')
    import torch
    import diffusion_model_amd as d
    d.set_seed(42)
    m = d.UNetMini(image_size=32).eval()
    smp = d.AncestralSampler(d.DDPMScheduler(), num_inference_steps=1000, eta=1.0)
    g = torch.Generator().manual_seed(1)
    B = 64
    x_T = torch.randn((B, 1, 32, 32), generator=g)
    nz = torch.randn((1000, B, 1, 32, 32), generator=g)
    out = {}
    for name, src in (("pinned", nz.pin_memory()), ("pageable", nz), ("device", nz.cuda())):
        smp.sample_from(m, x_T, src); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): smp.sample_from(m, x_T, src)
        torch.cuda.synchronize()
        out[name] = 3 * B / (time.perf_counter() - t0)
    print(json.dumps(out)); sys.exit(0)
libs = sys.argv[1].split(":")
for rnd in range(2):
    for l in libs:
        env = dict(os.environ); env["DM_SAMPLER_LIB"] = l
        o = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        if o.returncode: print(o.stderr[-2000:]); sys.exit(1)
        print(os.path.basename(l), o.stdout.strip().splitlines()[-1], flush=True)
