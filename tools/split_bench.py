#!/usr/bin/env python3
"""Tower time for different ways of cutting the 40-crop batch over concurrent streams."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
def run(split):
    cur = torch.cuda.current_stream()
    parts = list(px.split(split))
    for s in streams[:len(parts)]: s.wait_stream(cur)
    for pt, s, p in zip(pts, streams, parts):
        with torch.cuda.stream(s): ops.tower_forward(pt, p)
    for s in streams[:len(parts)]: cur.wait_stream(s)
SPLITS = [[20, 20], [24, 16], [22, 18], [28, 12], [14, 13, 13], [16, 12, 12], [10, 10, 10, 10], [40]]
for rep in range(2):
    for sp in SPLITS:
        for _ in range(2): run(sp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): run(sp)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 8
        print(f"split {sp}: {t*1e3:.2f} ms {40/t:.0f} crops/s", flush=True)
