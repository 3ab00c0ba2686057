#!/usr/bin/env python3
"""Round 4: tower latency of small, unsplit batches -- eager launch sequence vs HIP-graph replay of the same
sequence (ops.tower_forward, ViT-L/14-336, bf16, product library, one stream).  Interleaved rounds, median (min) of 20 calls.
Result (profiles/r04_small_batch_graph_replay.txt): identical -- the 2.4 ms floor is dependent kernel time, not launch overhead."""
import os, sys, time, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, weights as W
dev = torch.device("cuda:0"); dt = torch.bfloat16
pt = ops.pack_tower(W.strip_tower_prefix(W.make_tower_state_dict(W.CLIP_L_336, seed=1234)), W.CLIP_L_336, dt, dev)
px = W.synthetic_pixels(12, seed=0).to(dev).to(dt)
graphs = {}


def graphed(x):
    n = x.shape[0]
    if n not in graphs:
        xs, ys = torch.zeros_like(x), torch.empty((n, 576, 1024), dtype=dt, device=dev)
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side): ops.tower_forward(pt, xs, dt, False, out=ys)          # warm-up outside the capture
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): ops.tower_forward(pt, xs, dt, False, out=ys)
        graphs[n] = (g, xs, ys, pt.ws.buf)                                                  # keep the captured workspace alive
    g, xs, ys, _ = graphs[n]
    xs.copy_(x); g.replay()
    return ys.clone()


def ms(fn, x, reps=20):
    for _ in range(3): fn(x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(x); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts), min(ts)


print("crops: eager median (min) ms | graph replay median (min) ms | equal outputs")
for rnd in range(2):
    for n in (1, 2, 3, 5, 7, 9, 12):
        x = px[:n].contiguous()
        e = ms(lambda t: ops.tower_forward(pt, t, dt), x); g = ms(graphed, x)
        print(f"{n:3d}: {e[0]:6.3f} ({e[1]:6.3f}) | {g[0]:6.3f} ({g[1]:6.3f}) | {torch.equal(ops.tower_forward(pt, x, dt), graphed(x))}", flush=True)
