#!/usr/bin/env python3
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
TILES = [int(x) for x in os.environ.get("TILES", "4,5").split(",")]
def timeit(fn, warm=3, it=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
shapes = {"qkv": (3072, 1024, _lib.EPI_BIAS_T), "out": (1024, 1024, _lib.EPI_BIAS_RESID_F32),
          "fc1": (4096, 1024, _lib.EPI_BIAS_QUICKGELU_T), "fc2": (1024, 4096, _lib.EPI_BIAS_RESID_F32),
          "proj2": (4096, 4096, _lib.EPI_BIAS_F32)}
for M in (11540, 23080):
    for name, (N, K, epi) in shapes.items():
        a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
        b = torch.randn(N, device=dev); out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi >= 3 else dt)
        line = f"M={M:6d} {name:5s}: "
        for tile in TILES:
            lib.slime_gemm_force_tile(tile)
            t = timeit(lambda: ops.gemm(a, w, b, epi, out=out))
            line += f"tile{tile} {2.0*M*N*K/t/1e12:7.1f} | "
        print(line, flush=True)
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
def tower_bench(ns, tile):
    lib.slime_gemm_force_tile(tile)
    pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(ns)]
    streams = [torch.cuda.Stream() for _ in range(ns)]
    parts = list(px.chunk(ns))
    def run():
        cur = torch.cuda.current_stream()
        for s in streams: s.wait_stream(cur)
        for pt, s, p in zip(pts, streams, parts):
            with torch.cuda.stream(s): ops.tower_forward(pt, p)
        for s in streams: cur.wait_stream(s)
    for _ in range(2): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): run()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 6
    print(f"tower streams={ns} tile={tile}: {t*1e3:.2f} ms {40/t:.0f} crops/s {40*366.034e9/t/1e12:.0f} TF/s", flush=True)
for rep in range(2):
    for ns in (1, 2):
        for tile in TILES:
            tower_bench(ns, tile)
