# model forward beside a bf16 MFMA burner: which block-0 outputs (T1 tail = init5, T2 tail = down1.conv2) differ from the solo run?
import sys, os, ctypes as C
sys.path.insert(0, '.')
import torch
import diffusion_model_amd as d
A = C.CDLL(os.path.abspath("tools/variants/libaggressors.so"))
A.aggressor.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
dev = "cuda"
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
d.set_seed(42)
m = d.UNetMini(image_size=32).eval()
B = 32
x = torch.randn(B, 1, 32, 32, device=dev); t = torch.tensor(321)
buf = torch.zeros(1 << 20, device=dev)
def al(n): return (n * 4 + 255) // 256 * 256 // 4
full = B * 1024 * 64
off = al(64) + 3 * al(B * 1024) + al(B * 128)
A0, T1, T2 = off, off + al(full), off + 2 * al(full)
with torch.cuda.stream(s1): ref = m(x, t)
torch.cuda.synchronize()
ws_ref = m._ws.view(torch.float32).clone()
bad = t1 = t2 = 0; pat = {}
for trial in range(40):
    assert A.aggressor(1, 2048, 3000, buf.data_ptr(), buf.numel(), s2.cuda_stream) == 0
    with torch.cuda.stream(s1): o = m(x, t)
    torch.cuda.synchronize()
    ws = m._ws.view(torch.float32)
    if not torch.equal(o, ref): bad += 1
    d1 = (ws[T1 + full // 4:T1 + full] != ws_ref[T1 + full // 4:T1 + full]).nonzero().flatten()
    d2 = (ws[T2 + full // 4:T2 + full] != ws_ref[T2 + full // 4:T2 + full]).nonzero().flatten()
    t1 += int(d1.numel() > 0); t2 += int(d2.numel() > 0)
    for c in (d1 % 4).tolist(): pat[c] = pat.get(c, 0) + 1
print(f"{os.environ.get('DM_SAMPLER_LIB', 'default')} DM_INIT5={os.environ.get('DM_INIT5', '-')}: output differs {bad}/40; init5/conv1 output differs {t1}/40 (component histogram {pat}); conv2 output differs {t2}/40")
