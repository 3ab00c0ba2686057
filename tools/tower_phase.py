#!/usr/bin/env python3
"""Does it help to run the two tower streams out of phase?  Stream B starts `d` microseconds after stream A (a spin kernel in front
of its tower), so that A's full-machine GEMMs meet B's sub-round kernels instead of B's GEMMs.  40 crops, interleaved rounds."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); _lib.load(); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
parts = list(px.chunk(2))
CLK = 2.0e3          # spin cycles per microsecond (approximate: torch.cuda._sleep counts GPU clock cycles)
def run(d_us):
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for i, (pt, s, p) in enumerate(zip(pts, streams, parts)):
        with torch.cuda.stream(s):
            if i == 1 and d_us > 0: torch.cuda._sleep(int(d_us * CLK))
            ops.tower_forward(pt, p)
    for s in streams: cur.wait_stream(s)
def timed(d, n=6):
    for _ in range(2): run(d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run(d)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for rnd in range(3):
    for d in (0, 40, 80, 120, 160, 200, 300):
        print(f"round {rnd} offset {d:3d} us: {timed(d)*1e3:7.3f} ms", flush=True)
