#!/usr/bin/env python3
"""Timing ablations of the ping-pong GEMM (wrong results by construction; diagnostic)."""
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
TILE = int(os.environ.get("TILE", "4"))
lib.slime_gemm_force_tile(TILE)
for M in (256 * 64, 11540):          # 64 row tiles: exact rounds for N multiples of 1024
    for N, K in ((4096, 1024), (1024, 4096), (4096, 4096)):
        a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
        b = torch.randn(N, device=dev); out = torch.zeros(M, N, device=dev, dtype=dt)
        line = f"M={M:6d} N={N} K={K}: "
        for abl, name in ((0, "full"), (1, "noDMA"), (2, "noLDSread"), (3, "noDMA+noRead"), (4, "noBarrier"), (7, "mfmaOnly")):
            lib.slime_gemm_set_ablation(abl)
            t = timeit(lambda: ops.gemm(a, w, b, _lib.EPI_BIAS_T, out=out))
            line += f"{name} {2.0*M*N*K/t/1e12:7.1f} | "
        lib.slime_gemm_set_ablation(0)
        print(line, flush=True)
