#!/usr/bin/env python3
"""Two-stream tower time with the stream GEMM (tiles 10/11) vs the ping-pong GEMM (4/9) per shape (same-box A/B)."""
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
SH = {"qkv": (3072, 1024), "out": (1024, 1024), "fc1": (4096, 1024), "fc2": (1024, 4096)}
CONFIGS = [("auto (fc1 11, qkv 10, out/fc2 pp4)", {}),
           ("split barrier: fc1 13 qkv 12", {"fc1": 13, "qkv": 12}),
           ("split barrier: fc1 13", {"fc1": 13}),
           ("split barrier: fc1 13 qkv 13", {"fc1": 13, "qkv": 13})]
for _ in range(3): run()
for rep in range(2):
    for name, cfg in CONFIGS:
        lib.slime_gemm_set_shape_tile(0, 0, 0)
        for k, tile in cfg.items(): lib.slime_gemm_set_shape_tile(*SH[k], tile)
        for _ in range(2): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): run()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 8
        print(f"{name:36s}: {t*1e3:.2f} ms {40/t:.0f} crops/s", flush=True)
lib.slime_gemm_set_shape_tile(0, 0, 0)
