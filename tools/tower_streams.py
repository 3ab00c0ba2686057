#!/usr/bin/env python3
"""Tower time for 40 crops split over 1..5 HIP streams (each stream runs the whole 24-layer tower on its share of the crops),
interleaved rounds.  bench.py's encode() uses two."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); _lib.load(); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
px = W.synthetic_pixels(N, seed=0).to(dev).to(dt)
MAXS = 5
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(MAXS)]
streams = [torch.cuda.Stream() for _ in range(MAXS)]
def run(ns):
    cur = torch.cuda.current_stream()
    parts = list(px.chunk(ns))
    for s in streams[:ns]: s.wait_stream(cur)
    for pt, s, p in zip(pts, streams, parts):
        with torch.cuda.stream(s): ops.tower_forward(pt, p)
    for s in streams[:ns]: cur.wait_stream(s)
def timed(ns, n=6):
    for _ in range(2): run(ns)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run(ns)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for rnd in range(3):
    for ns in (1, 2, 3, 4, 5):
        print(f"round {rnd} {N} crops over {ns} streams: {timed(ns)*1e3:7.3f} ms", flush=True)
