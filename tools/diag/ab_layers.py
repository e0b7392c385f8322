# A/B of two builds of the library on one box: per-layer conv timing (kind H cin) alternating between the builds
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, '.')
    import torch
    from diffusion_model_amd import _lib
    L = _lib.lib(); dev = 'cuda'
    def run(kind, H, B=int(os.environ.get('AB_B', 1024)), c2=0, reps=40):
        base = {5: 0, 7: 0, 6: 1, 8: 1, 9: 2}.get(kind, kind); ks = 3 if base == 0 else 4
        OH = H // 2 if base == 1 else 2 * H if base == 2 else H
        x1 = torch.randn(B, H, H, 64, device=dev); x2 = torch.randn(B, H, H, 64, device=dev) if c2 else None
        cin = 64 + c2
        w = torch.randn(64, cin, ks, ks, device=dev) * 0.05
        wp = torch.empty(L.dm_conv_packed_floats(kind, cin, 64), device=dev)
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(L.dm_pack_conv_weight(kind, _lib.ptr(w), cin, 64, _lib.ptr(wp), s))
        b = torch.randn(64, device=dev); g = torch.rand(64, device=dev) + .5; be = torch.randn(64, device=dev); te = torch.randn(64, device=dev)
        pin = 4; si = torch.zeros(B, pin, 8, 2, device=dev); si[..., 1] = H * H * 8 / pin
        y = torch.empty(B, OH, OH, 64, device=dev); so = torch.empty(B, 16, 8, 2, device=dev)
        tf = not c2
        def call():
            _lib.check(L.dm_conv2d(kind, _lib.ptr(x1), 64, _lib.ptr(x2), c2, _lib.ptr(wp), _lib.ptr(b), 64, _lib.ptr(si) if tf else None, pin,
                                   H * H * 8 / pin, _lib.ptr(g), _lib.ptr(be), _lib.ptr(te), 0, _lib.ptr(y), _lib.ptr(so), B, H, H, s))
        for _ in range(30): call()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(reps): call()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    out = {}
    for spec in sys.argv[2:]:
        k, H, c2 = (int(v) for v in spec.split(","))
        out[spec] = run(k, H, c2=c2)
    print(json.dumps(out)); sys.exit(0)
libs = sys.argv[1].split(":"); specs = sys.argv[2:]
res = {l: {s: [] for s in specs} for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ); env["DM_SAMPLER_LIB"] = l
        o = subprocess.run([sys.executable, __file__, "--child"] + specs, env=env, capture_output=True, text=True)
        if o.returncode: print(o.stderr[-2000:]); sys.exit(1)
        d = json.loads(o.stdout.strip().splitlines()[-1])
        for s in specs: res[l][s].append(d[s])
for s in specs:
    print(s, "  ".join(f"{os.path.basename(l)}: " + "/".join(f"{v:.1f}" for v in res[l][s]) for l in libs))
