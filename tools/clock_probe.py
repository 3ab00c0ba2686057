#!/usr/bin/env python3
"""Effective shader clock under sustained GEMM load: exactly one round of 256 workgroups, block ticks vs wall."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
lib.slime_gemm_force_tile(4)
for M, N, K, reps in ((16384, 1024, 4096, 1), (16384, 1024, 4096, 20), (16384, 1024, 16384, 10), (16384, 1024, 1024, 20), (32768, 1024, 1024, 20), (16384 * 4, 1024, 1024, 20)):
    a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    b = torch.randn(N, device=dev); out = torch.zeros(M, N, device=dev, dtype=dt)
    nblk = (M // 256) * (N // 256)
    dbg = torch.zeros(nblk * 4, dtype=torch.int64, device=dev)
    for mode in (8, 0):
        lib.slime_gemm_set_ablation(mode); lib.slime_gemm_set_debug(dbg.data_ptr() if mode else None)
        for _ in range(3): ops.gemm(a, w, b, _lib.EPI_BIAS_T, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): ops.gemm(a, w, b, _lib.EPI_BIAS_T, out=out)
        e1.record(); torch.cuda.synchronize()
        wall_us = e0.elapsed_time(e1) * 1e3 / reps
        if mode:
            d = dbg.view(nblk, 4).cpu().double()
            tot = (d[:, 3] - d[:, 0])
            print(f"M={M} N={N} K={K} blocks={nblk} rounds={nblk/256:.2f} reps={reps}: debug wall/launch {wall_us:8.1f} us; block ticks mean {tot.mean():9.0f} max {tot.max():9.0f} -> {tot.max()*max(1,nblk/256)/wall_us/1e3:.3f} GHz-equivalent if gapless", flush=True)
        else:
            print(f"      production build wall/launch {wall_us:8.1f} us  {2.0*M*N*K/wall_us/1e6:7.1f} TF/s", flush=True)
    lib.slime_gemm_set_ablation(0); lib.slime_gemm_set_debug(None)
