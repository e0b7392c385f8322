import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, diffusion_model_amd as d
d.set_seed(42)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
m = d.UNetMini(image_size=32, precision=prec).eval()
sched = d.DDPMScheduler()
per, raw = bench.profile_model(m, sched, B, 32, 32, reps=5)
tot = sum(v["avg_us"] * v["launches"] for v in per.values())
print(os.environ.get("DM_SAMPLER_LIB", "default"), "total us/eval", round(tot, 1))
for k, v in per.items():
    print("   ", k.ljust(24), v["launches"], round(v["avg_us"], 1), round(v["share"], 3), round(v["tflops"], 1), round(v["alg_gbps"]))
