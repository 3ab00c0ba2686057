#!/usr/bin/env python3
"""How much does the per-step join of the two tower streams cost?  (a) join after every step (the product's encode()),
(b) each stream runs its half-batch tower K times back to back, one join at the end."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, weights as W
dev = torch.device("cuda:0"); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
parts = list(px.chunk(2))
K = 10
def joined():
    cur = torch.cuda.current_stream()
    for _ in range(K):
        for s in streams: s.wait_stream(cur)
        for pt, s, p in zip(pts, streams, parts):
            with torch.cuda.stream(s): ops.tower_forward(pt, p)
        for s in streams: cur.wait_stream(s)
def free():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for _ in range(K):
        for pt, s, p in zip(pts, streams, parts):
            with torch.cuda.stream(s): ops.tower_forward(pt, p)
    for s in streams: cur.wait_stream(s)
for fn in (joined, free): fn()
torch.cuda.synchronize()
for rep in range(3):
    for name, fn in (("join every step", joined), ("join once per 10 steps", free)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(); torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / K
        print(f"{name:24s}: {t*1e3:.2f} ms/step {40/t:.0f} crops/s", flush=True)
