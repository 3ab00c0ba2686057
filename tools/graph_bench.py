#!/usr/bin/env python3
"""Host enqueue time of the tower and HIP-graph replay of the two-stream step (diagnostic)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
parts = [px[:20].contiguous(), px[20:].contiguous()]
outs = [None, None]
def run():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for i, (pt, s, p) in enumerate(zip(pts, streams, parts)):
        with torch.cuda.stream(s): outs[i] = ops.tower_forward(pt, p)
    for s in streams: cur.wait_stream(s)
for _ in range(3): run()
torch.cuda.synchronize()
t0 = time.perf_counter(); run(); t_host = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"eager: host enqueue {t_host*1e3:.2f} ms, total {t_all*1e3:.2f} ms")
t0 = time.perf_counter()
for _ in range(10): run()
torch.cuda.synchronize(); print(f"eager steady: {(time.perf_counter()-t0)/10*1e3:.2f} ms/step")
# graph capture
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2): run()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        run()
    torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); print(f"graph replay: {(time.perf_counter()-t0)/10*1e3:.2f} ms/step")
    print("graph output equal:", all(torch.equal(a, b) for a, b in zip(outs, ref)))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
