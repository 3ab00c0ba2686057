#!/usr/bin/env python3
"""Marginal cost of each per-layer kernel inside the two-stream tower: tower time with that kernel skipped (wrong results)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
parts = list(px.chunk(2))
def run():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for pt, s, p in zip(pts, streams, parts):
        with torch.cuda.stream(s): ops.tower_forward(pt, p)
    for s in streams: cur.wait_stream(s)
NAMES = {1: "qkv", 2: "attention", 3: "out_proj", 5: "fc1", 6: "fc2"}     # (0 / 4 were the LayerNorm launches: folded into the GEMMs)
CASES = [("full", 0)] + [(f"without {n}", 1 << k) for k, n in NAMES.items()] + [("GEMMs only", 0b0000100), ("full", 0)]
for _ in range(3): run()
base = None
for name, mask in CASES:
    lib.slime_vit_set_skip_mask(mask)
    for _ in range(2): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): run()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 8
    if base is None: base = t
    print(f"{name:20s}: {t*1e3:6.2f} ms  (marginal {((base - t)*1e3):5.2f} ms = {(base - t)/base*100:4.1f} %)", flush=True)
lib.slime_vit_set_skip_mask(0)
