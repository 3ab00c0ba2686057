#!/usr/bin/env python3
"""Two-stream tower time with per-GEMM-shape tile overrides (slime_gemm_set_shape_tile): which of the tower's
four GEMM shapes, if any, prefers the 128x128 (2 workgroups/CU) tile inside the real launch mix?"""
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
SHAPES = {"qkv": (3072, 1024), "out": (1024, 1024), "fc1": (4096, 1024), "fc2": (1024, 4096)}
CONFIGS = [(), ("qkv", "out", "fc2", "fc1"), ("out", "fc2"), ("qkv",)]
for _ in range(3): run()
for rep in range(2):
    for tile in (4,):          # auto = cost-model rule (192-row tile for qkv/out/fc2); listed shapes forced back to 256 rows
        for cfg in CONFIGS:
            lib.slime_gemm_set_shape_tile(0, 0, 0)
            for name in cfg: lib.slime_gemm_set_shape_tile(*SHAPES[name], tile)
            for _ in range(2): run()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8): run()
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 8
            print(f"tile {tile} for {'+'.join(cfg) or 'none':12s}: {t*1e3:.2f} ms {40/t:.0f} crops/s", flush=True)
lib.slime_gemm_set_shape_tile(0, 0, 0)

lib.slime_gemm_set_shape_tile(0, 0, 0)
