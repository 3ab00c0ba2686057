#!/usr/bin/env python3
"""Stream GEMM (tiles 10 = 192x256, 11 = 256x256, four waves) vs ping-pong (4 = 256x256, 9 = 192x256) on the tower's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
def timeit(fn, warm=3, it=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
for M in (11540, 16384):
    for name, N, K, epi in (("qkv", 3072, 1024, 0), ("out", 1024, 1024, 4), ("fc1", 4096, 1024, 1), ("fc2", 1024, 4096, 4), ("mlp2", 4096, 4096, 3)):
        a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
        b = torch.randn(N, device=dev)
        c = torch.zeros(M, N, device=dev, dtype=dt if epi <= 2 else torch.float32)
        line = f"M={M:6d} {name:5s}: "
        for tile in (10, 12, 11, 13):
            lib.slime_gemm_force_tile(tile)
            t = timeit(lambda: ops.gemm(a, w, b, epi, out=c))
            line += f"tile {tile:2d} {2.0*M*N*K/t/1e12:7.1f} | "
        lib.slime_gemm_force_tile(0)
        print(line, flush=True)
