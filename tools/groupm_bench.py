#!/usr/bin/env python3
"""Two-stream tower time vs the GEMM L2 patch height (slime_gemm_set_group_m): does L2-miss traffic matter?"""
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
for _ in range(3): run()
for rep in range(3):
    for rule in (8, 2, 4, 16, 46):
        lib.slime_gemm_set_group_m(rule)
        for _ in range(2): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): run()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 8
        print(f"group_m {rule}: {t*1e3:.2f} ms {40/t:.0f} crops/s", flush=True)

# standalone GEMMs at the half-batch shapes
M = 11540
for name, N, K in (("qkv", 3072, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096), ("out", 1024, 1024)):
    a = torch.randn(M, K, device=dev).to(dt); w = torch.randn(N, K, device=dev).to(dt) * 0.03
    b = torch.zeros(N, device=dev); c = torch.empty(M, N, device=dev, dtype=dt)
    for gm in (8, 2, 4, 16, 46):
        lib.slime_gemm_set_group_m(gm)
        for _ in range(3): ops.gemm(a, w, b, 0, out=c)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm(a, w, b, 0, out=c)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20
        print(f"{name} group_m {gm}: {t*1e3:.1f} us {2*M*N*K/t/1e9:.0f} TF/s", flush=True)
