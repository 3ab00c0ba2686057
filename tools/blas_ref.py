#!/usr/bin/env python3
"""Reference point, not a product path: torch.matmul (hipBLASLt/rocBLAS) vs slime_gemm on the tower's GEMM shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib
dev = torch.device("cuda:0"); dt = torch.bfloat16
def timeit(fn, warm=5, it=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
for M in (11540, 23080):
    for name, N, K in (("qkv", 3072, 1024), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096), ("mlp2", 4096, 4096)):
        a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
        b = torch.zeros(N, device=dev); c = torch.empty(M, N, device=dev, dtype=dt)
        t_lib = timeit(lambda: torch.matmul(a, w.t(), out=c))
        t_lin = timeit(lambda: torch.nn.functional.linear(a, w))
        t_ours = timeit(lambda: ops.gemm(a, w, b, 0, out=c))
        fl = 2.0 * M * N * K
        print(f"M={M:6d} {name:5s} N={N} K={K}: hipBLASLt matmul {fl/t_lib/1e12:7.1f} TF/s, F.linear {fl/t_lin/1e12:7.1f} TF/s | slime_gemm (bias epilogue) {fl/t_ours/1e12:7.1f} TF/s", flush=True)
