#!/usr/bin/env python3
"""Tile choice for the adapter's GEMM shapes (stacked projection MLP, resampler projections)."""
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
for (name, M, N, K, epi) in (("mlp1 stacked", 13824, 4096, 1024, 2), ("mlp2 stacked", 13824, 4096, 4096, 3), ("kv local", 18432, 1024, 1024, 0),
                            ("kv global", 4608, 1024, 1024, 0), ("o local", 4608, 1024, 1024, 3)):
    a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt); b = torch.randn(N, device=dev)
    c = torch.zeros(M, N, device=dev, dtype=dt if epi <= 2 else torch.float32)
    line = f"{name:13s} M={M} N={N} K={K}: "
    for tile in (0, 4, 9, 10, 11, 3):
        lib.slime_gemm_force_tile(tile)
        t = timeit(lambda: ops.gemm(a, w, b, epi, out=c))
        line += f"tile {tile:2d} {t*1e6:6.1f} us {2.0*M*N*K/t/1e12:6.0f} | "
    lib.slime_gemm_force_tile(0)
    print(line, flush=True)
