# fp32 accuracy of Winograd F(4x4,3x3) against F(2x2,3x3) and the direct form on layer-like data (truth: float64 direct)
import numpy as np, torch, torch.nn.functional as F
torch.manual_seed(0)
def mats(m):
    if m == 2:
        Bt = np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], float)
        G = np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], float)
        At = np.array([[1,1,1,0],[0,1,-1,-1]], float)
    else:
        Bt = np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], float)
        G = np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]], float)
        At = np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]], float)
    return Bt, G, At
def wino(x, w, m, dt):
    # x [C,H,W], w [K,C,3,3]; output [K,H,W]; all arithmetic in dtype dt, the channel sum in chunks of 2 (MFMA-like)
    Bt, G, At = [torch.tensor(a, dtype=dt) for a in mats(m)]
    C, H, W = x.shape; K = w.shape[0]; a = m + 2
    xp = F.pad(x.to(dt), (1, 1, 1, 1))
    U = torch.einsum('ia,kcab,jb->kcij', G, w.to(dt), G)                 # [K,C,a,a]
    out = torch.zeros(K, H, W, dtype=dt)
    for ty in range(0, H, m):
        for tx in range(0, W, m):
            d = xp[:, ty:ty + a, tx:tx + a]
            V = torch.einsum('ia,cab,jb->cij', Bt, d, Bt)                 # [C,a,a]
            M = torch.zeros(K, a, a, dtype=dt)
            for c in range(0, C, 2):                                      # two products per accumulate
                M = M + torch.einsum('kcij,cij->kij', U[:, c:c + 2], V[c:c + 2])
            out[:, ty:ty + m, tx:tx + m] = torch.einsum('ia,kab,jb->kij', At, M, At)
    return out
C, K, H = 64, 64, 16
for trial in range(3):
    x = F.silu(torch.randn(C, H, H) * 1.5 + 0.2) + 0.3 * torch.randn(C, 1, 1)
    w = torch.randn(K, C, 3, 3) * 0.06
    ref = F.conv2d(x.double()[None], w.double(), padding=1)[0]
    d32 = F.conv2d(x[None], w, padding=1)[0]
    e2 = wino(x, w, 2, torch.float32); e4 = wino(x, w, 4, torch.float32)
    chk = (wino(x, w, 4, torch.float64) - ref).abs().max().item()
    print(f"|ref|max {ref.abs().max():.2f}: max-abs error direct fp32 {(d32-ref).abs().max():.2e}, F(2x2,3x3) fp32 {(e2-ref).abs().max():.2e}, "
          f"F(4x4,3x3) fp32 {(e4-ref).abs().max():.2e}  (F(4x4) in float64: {chk:.1e}); rms F2 {(e2-ref).pow(2).mean().sqrt():.2e} F4 {(e4-ref).pow(2).mean().sqrt():.2e}")
