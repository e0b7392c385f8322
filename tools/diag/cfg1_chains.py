# ancestral-1000, 64 images, eta=1 (BASELINE configs[0]) as 1 / 2 / 4 independent sub-batch chains on separate streams:
# at B=64 most layers occupy a fraction of the CUs and every launch is latency-bound, so concurrent chains fill the chip.
import sys, time, threading
sys.path.insert(0, '.')
import torch
import diffusion_model_amd as d
import diffusion_model_amd.sampler as sampler_mod
d.set_seed(42)
m0 = d.UNetMini(image_size=32).eval()
sd = m0.state_dict()
N, n = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 1000
g = torch.Generator().manual_seed(1)
x_T = torch.randn((N, 1, 32, 32), generator=g)
nz = torch.randn((n, N, 1, 32, 32), generator=g).cuda()
ref = None
for nc in (1, 2, 4, 8):
    models, smps, streams = [], [], []
    for c in range(nc):
        mm = d.UNetMini(image_size=32).eval(); mm.load_state_dict(sd); models.append(mm)
        smps.append(d.AncestralSampler(d.DDPMScheduler(), num_inference_steps=n, eta=1.0) if n > 50 else d.DDIMSampler(d.DDPMScheduler(), num_inference_steps=n, eta=1.0))
        streams.append(torch.cuda.Stream())
    b = N // nc
    parts_x = [x_T[c * b:(c + 1) * b].contiguous() for c in range(nc)]
    parts_z = [nz[:, c * b:(c + 1) * b].contiguous() for c in range(nc)]
    def run():
        outs = []
        for c in range(nc):
            sampler_mod._side_stream = lambda dev, c=c: streams[c]  # (the package runs every loop on ONE side stream)
            with torch.cuda.stream(streams[c]):
                outs.append(smps[c].sample_from(models[c], parts_x[c], parts_z[c]))
        torch.cuda.synchronize()
        return torch.cat(outs)
    out = run()
    t0 = time.perf_counter()
    for _ in range(3): out = run()
    dt = (time.perf_counter() - t0) / 3
    if ref is None: ref = out
    print(f"chains={nc} batch/chain={b}: {N / dt:8.1f} images/s  ({dt * 1e3:.1f} ms)  bit-identical to one chain: {torch.equal(out, ref)}", flush=True)
