#!/usr/bin/env python3
"""Two-stream tower (40 crops) with per-shape tile overrides (diagnostic build), interleaved rounds: does any alternative tile for a
tower GEMM beat the shipped dispatch INSIDE the step?  usage: tile_rule_ab.py "N,K,tile[;N,K,tile...]" ..."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
side = torch.cuda.Stream()
parts = list(px.chunk(2))


def run2():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side): b = ops.tower_forward(pts[1], parts[1])
    a = ops.tower_forward(pts[0], parts[0])
    cur.wait_stream(side)
    return a, b


def run1():
    return ops.tower_forward(pts[0], px)


def rules(spec):
    lib.slime_gemm_set_shape_tile(0, 0, 0)
    for r in filter(None, spec.split(";")):
        n, k, t = (int(x) for x in r.split(","))
        lib.slime_gemm_set_shape_tile(n, k, t)


configs = [""] + sys.argv[1:]
rules(""); ref = run2(); torch.cuda.synchronize()
print("rules: two streams x 20 ms | one stream x 40 ms | bit-equal to shipped")
for rnd in range(3):
    for c in configs:
        rules(c)
        out = run2(); eq = all(torch.equal(x, y) for x, y in zip(out, ref))
        ts = []
        for fn in (run2, run1):
            for _ in range(2): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8): fn()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 8 * 1e3)
        print(f"{c or 'shipped':32s}: {ts[0]:6.2f} | {ts[1]:6.2f} | {eq}", flush=True)
rules("")
