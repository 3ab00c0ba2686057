#!/usr/bin/env python3
"""40-crop tower over two streams with UNEQUAL halves (a / 40-a crops): do two streams that drift through each other's kernel
sequence overlap better than two that run it in lock step?  Product library, interleaved rounds."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, weights as W
dev = torch.device("cuda:0"); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
side = torch.cuda.Stream()
out = torch.empty((40, 576, 1024), dtype=dt, device=dev)


def run(a):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side): ops.tower_forward(pts[1], px[a:], dt, False, out=out[a:])
    ops.tower_forward(pts[0], px[:a], dt, False, out=out[:a])
    cur.wait_stream(side)


run(20); torch.cuda.synchronize(); ref = out.clone()
print("first-stream crops: ms per 40 crops (three rounds) | bit-equal")
for a in (20, 21, 22, 24, 26, 28, 32):
    ts = []
    for rnd in range(3):
        for _ in range(2): run(a)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): run(a)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 8 * 1e3)
    print(f"{a:2d} + {40-a:2d}: " + " ".join(f"{t:6.2f}" for t in ts) + f" | {torch.equal(out, ref)}", flush=True)
