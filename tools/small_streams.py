#!/usr/bin/env python3
"""Where does splitting a tower call over two streams start to pay?  (HipCLIPVisionModel.encode's `n >= 8` threshold)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, weights as W
dev = torch.device("cuda:0"); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def one(px): ops.tower_forward(pts[0], px)
def two(px):
    cur = torch.cuda.current_stream(); h = (px.shape[0] + 1) // 2
    for s in streams: s.wait_stream(cur)
    for pt, s, p in zip(pts, streams, (px[:h], px[h:])):
        with torch.cuda.stream(s): ops.tower_forward(pt, p)
    for s in streams: cur.wait_stream(s)
for n in (4, 5, 8, 10, 12, 16, 20, 30):
    px = W.synthetic_pixels(n, seed=n).to(dev).to(dt)
    res = []
    for fn in (one, two):
        for _ in range(3): fn(px)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn(px)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"n={n:3d}: one stream {res[0]:6.2f} ms ({n/res[0]*1e3:5.0f} crops/s) | two streams {res[1]:6.2f} ms ({n/res[1]*1e3:5.0f} crops/s)", flush=True)
