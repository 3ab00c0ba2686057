#!/usr/bin/env python3
"""A/B of product-library variants (tools/build_variants.sh) on the 40-crop tower: each variant runs in its own process (one library
per process), rounds interleaved; prints ms per 40 crops (two streams x 20 | one stream x 40) and a checksum of the features so that
bit-equality across variants is visible.  usage: lib_variant_ab.py [--rounds R] name ...   (name 'product' = slime_amd/libslime_hip.so)"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import time, hashlib, torch
    sys.path.insert(0, ROOT)
    from slime_amd import _lib
    name = sys.argv[2]
    if name != "product":
        _lib.LIB_PATH = os.path.join(ROOT, "slime_amd", "variants", f"libslime_hip_{name}.so")
    from slime_amd import ops, weights as W
    dev = torch.device("cuda:0"); dt = torch.bfloat16
    tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
    px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
    pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
    side = torch.cuda.Stream(); parts = list(px.chunk(2))

    def run2():
        cur = torch.cuda.current_stream(); side.wait_stream(cur)
        with torch.cuda.stream(side): b = ops.tower_forward(pts[1], parts[1])
        a = ops.tower_forward(pts[0], parts[0]); cur.wait_stream(side)
        return a, b

    def run1():
        return ops.tower_forward(pts[0], px)
    a, b = run2(); torch.cuda.synchronize()
    h = hashlib.sha1(torch.cat([a, b]).float().cpu().numpy().tobytes()).hexdigest()[:12]
    ts = []
    for fn in (run2, run1):
        best = []
        for rep in range(3):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(12): fn()
            torch.cuda.synchronize(); best.append((time.perf_counter() - t0) / 12 * 1e3)
        ts.append(sorted(best)[1])
    print(json.dumps({"name": name, "two": ts[0], "one": ts[1], "sha": h}))
    sys.exit(0)

args = sys.argv[1:]; rounds = 3
if args and args[0] == "--rounds": rounds = int(args[1]); args = args[2:]
print("variant: two streams x 20 crops ms | one stream x 40 crops ms | sha1 of the 40-crop features (median of 3 x 12 steps per round)")
for r in range(rounds):
    for n in args:
        o = subprocess.run([sys.executable, __file__, "--child", n], capture_output=True, text=True, timeout=600)
        try:
            d = json.loads(o.stdout.strip().splitlines()[-1])
            print(f"{d['name']:24s}: {d['two']:6.2f} | {d['one']:6.2f} | {d['sha']}", flush=True)
        except Exception:
            print(n, "FAILED", o.stderr[-600:], flush=True)
