#!/usr/bin/env python3
"""Two-stream tower time (40 crops) under diagnostic-build switches, interleaved rounds.  Usage:
   python tools/tower_ab.py attn      attention variant 0 (attn64r) vs 2 (round-1 attn64)
   python tools/tower_ab.py attn32    attn64r vs attn32 (one wave per SIMD, 32x32 MFMAs), one / two workgroups per (crop, head)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(int(os.environ.get('NSTREAMS', '2')))]
NS = int(os.environ.get('NSTREAMS', '2'))
streams = [torch.cuda.Stream() for _ in range(NS)]
parts = list(px.chunk(NS))
def run():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for pt, s, p in zip(pts, streams, parts):
        with torch.cuda.stream(s): ops.tower_forward(pt, p)
    for s in streams: cur.wait_stream(s)
def timed(n=8):
    for _ in range(2): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
what = sys.argv[1] if len(sys.argv) > 1 else "attn"
cases = {"attn": [("attn64r", lambda: lib.slime_attention_set_variant(7)), ("attn64 (r1)", lambda: lib.slime_attention_set_variant(2)),
                  ("attn64w 12wave", lambda: lib.slime_attention_set_variant(3))],
         "attn32": [("attn64r", lambda: lib.slime_attention_set_variant(7)), ("attn32", lambda: lib.slime_attention_set_variant(4)),
                    ("attn32 uncut", lambda: lib.slime_attention_set_variant(6))]}[what]
for _ in range(3): run()
for rnd in range(4):
    for name, setup in cases:
        setup()
        print(f"round {rnd} {name:14s}: {timed()*1e3:7.3f} ms", flush=True)
cases[0][1]()
