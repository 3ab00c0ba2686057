#!/usr/bin/env python3
"""Two whole 40-crop batches in flight (one per stream) vs one batch split over two streams: throughput only."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, weights as W
dev = torch.device("cuda:0"); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
px = W.synthetic_pixels(80, seed=0).to(dev).to(dt)
def run(parts):
    cur = torch.cuda.current_stream()
    for s in streams[:len(parts)]: s.wait_stream(cur)
    for pt, s, p in zip(pts, streams, parts):
        with torch.cuda.stream(s): ops.tower_forward(pt, p)
    for s in streams[:len(parts)]: cur.wait_stream(s)
cases = [("one batch: 2 x 20", [px[:20], px[20:40]], 40), ("two batches: 2 x 40", [px[:40], px[40:]], 80),
         ("two batches: 4 x 20", [px[:20], px[20:40], px[40:60], px[60:]], 80), ("80 crops: 3 x 27", [px[:27], px[27:54], px[54:]], 80)]
for rep in range(2):
    for name, parts, n in cases:
        for _ in range(2): run(parts)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): run(parts)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 6
        print(f"{name:22s}: {t*1e3:6.2f} ms {n/t:5.0f} crops/s", flush=True)
