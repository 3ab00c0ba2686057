#!/usr/bin/env python3
"""A/B: one tower call on N crops vs S concurrent streams of N/S crops (diagnostic)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W

dev = torch.device("cuda:0")
lib = _lib.load_diag()
dt = torch.bfloat16
N = int(os.environ.get("CROPS", "40"))
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(N, seed=0).to(dev).to(dt)

def bench(nstreams, tile, it=6):
    lib.slime_gemm_force_tile(tile)
    pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    parts = list(px.chunk(nstreams))
    def run():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        outs = []
        for pt, s, p in zip(pts, streams, parts):
            with torch.cuda.stream(s):
                outs.append(ops.tower_forward(pt, p))
        for s in streams:
            cur.wait_stream(s)
        return outs
    for _ in range(2): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it): run()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / it
    print(f"streams={nstreams} tile={tile}: {t*1e3:.2f} ms  {N/t:.0f} crops/s  {N*366.034e9/t/1e12:.0f} TF/s", flush=True)

for rep in range(2):
    for tile in (0, 1, 3, 4):
        for ns in (1, 2, 4):
            bench(ns, tile)
