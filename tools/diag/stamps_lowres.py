# phase durations (cycles, s_memtime) of lowres_fused_kernel for the LAST image of every workgroup; -DDM_STAMPS build:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -shared -fPIC -DDM_STAMPS diffusion_model_amd/csrc/dm_api.hip -o tools/diag/bin/libdm_stamps.so
import os, sys, ctypes as C
os.environ["DM_SAMPLER_LIB"] = "tools/diag/bin/libdm_stamps.so"
sys.path.insert(0, '.')
import torch, numpy as np
import diffusion_model_amd as d
from diffusion_model_amd import _lib
L = _lib.lib()
L.dm_debug_set_stamps.argtypes = [C.c_void_p]
NAMES = ["A d2.transform 16->8", "B d3.conv1", "C d3.conv2", "D d3.transform 8->4", "E bott.0", "F bott.3", "G u1.conv1", "H u1.conv2",
         "I u1.convT 4->8", "J u2.conv1", "K u2.conv2", "L u2.convT 8->16"]
for B in [int(a) for a in sys.argv[1:]] or [64, 1024]:
    m = d.UNetMini(image_size=32).eval()
    x = torch.randn(B, 1, 32, 32, device="cuda")
    t = torch.tensor(500)
    for _ in range(5): m(x, t)
    torch.cuda.synchronize()
    st = torch.zeros((1 << 20) + (1 << 15), dtype=torch.int64, device="cuda")
    L.dm_debug_set_stamps(st.data_ptr())
    m(x, t); torch.cuda.synchronize()
    L.dm_debug_set_stamps(None)
    a = st[1 << 20:].cpu().numpy().reshape(-1, 32)
    a = a[a[:, 0] != 0]
    b = a[:, 13:16]
    print(f"   phase B split: pre-K {np.median(b[:,0]-a[:,1]):.0f}  K loop {np.median(b[:,1]-b[:,0]):.0f}  bias+GN+SiLU {np.median(b[:,2]-b[:,1]):.0f}  planes {np.median(a[:,2]-b[:,2]):.0f}")
    a = a[:, :13]
    dlt = np.diff(a, axis=1)
    print(f"B={B}: {len(a)} workgroups; last image: median {np.median(a[:,12]-a[:,0]):.0f} cycles")
    for i, nm in enumerate(NAMES):
        print(f"   {nm:24s} median {np.median(dlt[:, i]):8.0f}  p10 {np.percentile(dlt[:, i], 10):8.0f}  p90 {np.percentile(dlt[:, i], 90):8.0f}")
