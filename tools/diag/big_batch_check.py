# B = 65,536 on ONE GPU (BASELINE configs[2] unsharded): 4.3e9 activation elements per layer - 32-bit index overflow check.
# Rows at the start, across the 2^32-element boundary and at the end must equal the same rows sampled in a small batch.
import sys, time
sys.path.insert(0, '.')
import torch
import diffusion_model_amd as d
d.set_seed(42)
m = d.UNetMini(image_size=32).eval()
smp = d.DDIMSampler(d.DDPMScheduler(), num_inference_steps=int(sys.argv[2]) if len(sys.argv) > 2 else 10, eta=0.0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
g = torch.Generator().manual_seed(7)
x_T = torch.randn((B, 1, 32, 32), generator=g)
t0 = time.perf_counter(); full = smp.sample_from(m, x_T); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"B={B}: {dt:.2f} s, finite: {bool(torch.isfinite(full).all())}, workspace {m._ws.numel() / 2**30:.1f} GiB")
for lo in (0, B // 2 - 8, B // 2, 3 * B // 4 - 4, B - 16):
    part = smp.sample_from(m, x_T[lo:lo + 16].contiguous())
    print(f"   rows {lo}..{lo + 15} equal the small-batch run: {torch.equal(part, full[lo:lo + 16])}")
