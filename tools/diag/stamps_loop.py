import os, sys, ctypes as C
os.environ["DM_SAMPLER_LIB"] = "tools/diag/build/libdm_stamps.so"
sys.path.insert(0, '.')
import torch, numpy as np
import diffusion_model_amd as d
from diffusion_model_amd import _lib
L = _lib.lib()
L.dm_debug_set_stamps.argtypes = [C.c_void_p]
d.set_seed(42)
m = d.UNetMini(image_size=32).eval()
smp = d.DDIMSampler(d.DDPMScheduler(), num_inference_steps=50, eta=0.0)
x = torch.randn(1024, 1, 32, 32)
smp.use_graph = False   # stamps pointer is read at launch time: eager launches, still back to back
smp.sample_from(m, x); torch.cuda.synchronize()
st = torch.zeros(1 << 21, dtype=torch.int64, device='cuda')
L.dm_debug_set_stamps(st.data_ptr())
smp.sample_from(m, x); torch.cuda.synchronize()
L.dm_debug_set_stamps(None)
a = st.cpu().numpy().reshape(-1, 8)
a = a[(a[:, 0] != 0) & (a[:, 5] > a[:, 4])]
clk = (a[:, 3] - a[:, 0]) / (a[:, 5] - a[:, 4]) * 100.0
print("WGs", len(a), "steady-state in-kernel clock MHz: median %.0f p10 %.0f p90 %.0f" % (np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
b = a[a[:, 6] > a[:, 0]]
for nm, v in (("issue", b[:, 6] - b[:, 0]), ("tables", b[:, 7] - b[:, 6]), ("xform+bar", b[:, 1] - b[:, 7])):
    print(f"   {nm:9s} median {np.median(v):9.0f}  p10 {np.percentile(v,10):9.0f}  p90 {np.percentile(v,90):9.0f}")
for nm, v in (("stage", a[:, 1] - a[:, 0]), ("kloop", a[:, 2] - a[:, 1]), ("epi", a[:, 3] - a[:, 2]), ("life", a[:, 3] - a[:, 0])):
    print(f"   {nm:6s} median {np.median(v):9.0f}  p10 {np.percentile(v,10):9.0f}  p90 {np.percentile(v,90):9.0f}")
