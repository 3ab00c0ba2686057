#!/usr/bin/env python3
"""Tower latency at the small per-rank crop counts (1 ... 12 crops; encode()'s own stream policy) for product-library variants, one
process per library (SLIME_HIP_LIBRARY), rounds interleaved; prints ms per pass and a checksum of the 12-crop features (the tile
choice must be bit-invisible).  usage: small_latency_ab.py [--rounds R] name ...   (name 'product' = slime_amd/libslime_hip.so)"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = tuple(int(x) for x in os.environ["AB_SIZES"].split(",")) if os.environ.get("AB_SIZES") else (1, 2, 3, 4, 5, 7, 9, 12)   # AB_SIZES=4,5,6,...: other crop counts
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import time, hashlib, torch
    sys.path.insert(0, ROOT)
    from slime_amd import weights as W
    from slime_amd.model.multimodal_encoder.clip_encoder import HipCLIPVisionModel
    dev, dt = torch.device("cuda:0"), torch.bfloat16
    vm = HipCLIPVisionModel(W.CLIP_L_336); vm.load_state_dict(W.make_tower_state_dict(W.CLIP_L_336, seed=1234)); vm.to(dev).to(dt)
    px = W.synthetic_pixels(max(12, max(SIZES)), seed=0).to(dev).to(dt)
    out = {}
    for n in SIZES:
        x = px[:n].contiguous()
        best = []
        for rep in range(3):
            for _ in range(3): vm.encode(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): vm.encode(x)
            torch.cuda.synchronize(); best.append((time.perf_counter() - t0) / 10 * 1e3)
        out[n] = round(min(best), 3)
    f = vm.encode(px[:12].contiguous()); torch.cuda.synchronize()
    print(json.dumps({"ms": out, "sha": hashlib.sha1(f.float().cpu().numpy().tobytes()).hexdigest()[:12]}))
    sys.exit(0)
args = sys.argv[1:]; rounds = 2
if args and args[0] == "--rounds": rounds = int(args[1]); args = args[2:]
print("variant: ms per tower pass at " + " / ".join(str(n) for n in SIZES) + " crops | sha1 of the 12-crop features")
for r in range(rounds):
    for n in args:
        env = dict(os.environ)
        if n != "product": env["SLIME_HIP_LIBRARY"] = os.path.join(ROOT, "slime_amd", "variants", f"libslime_hip_{n}.so")
        o = subprocess.run([sys.executable, __file__, "--child"], capture_output=True, text=True, timeout=600, env=env)
        try:
            d = json.loads(o.stdout.strip().splitlines()[-1])
            print(f"{n:10s}: " + " ".join(f"{d['ms'][str(k)] if str(k) in d['ms'] else d['ms'][k]:6.3f}" for k in SIZES) + f" | {d['sha']}", flush=True)
        except Exception:
            print(n, "FAILED", o.stderr[-500:], flush=True)
