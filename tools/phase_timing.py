#!/usr/bin/env python3
"""Per-workgroup phase timing of the ping-pong GEMM via s_memtime stamps (diagnostic build ABL=8)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
lib.slime_gemm_force_tile(4)
def q(x): return f"mean {x.mean():8.0f} p10 {x.quantile(0.1):8.0f} p50 {x.median():8.0f} p90 {x.quantile(0.9):8.0f} max {x.max():8.0f}"
for M, N, K in ((11540, 4096, 1024), (23080, 4096, 1024), (23080, 1024, 4096)):
    a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    b = torch.randn(N, device=dev); out = torch.zeros(M, N, device=dev, dtype=dt)
    nblk = ((M + 255) // 256) * (N // 256)
    dbg = torch.zeros(nblk * 4, dtype=torch.int64, device=dev)
    lib.slime_gemm_set_ablation(8); lib.slime_gemm_set_debug(dbg.data_ptr())
    for _ in range(5): ops.gemm(a, w, b, _lib.EPI_BIAS_T, out=out)      # warm + sustained load
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    ops.gemm(a, w, b, _lib.EPI_BIAS_T, out=out)
    e1.record(); torch.cuda.synchronize()
    wall_us = e0.elapsed_time(e1) * 1e3
    lib.slime_gemm_set_ablation(0); lib.slime_gemm_set_debug(None)
    d = dbg.view(nblk, 4).cpu().double()
    pro, loop, epi = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]
    print(f"M={M} N={N} K={K} blocks={nblk}: wall {wall_us:.1f} us (last launch, events)")
    print("  prologue :", q(pro)); print("  main loop:", q(loop)); print("  epilogue :", q(epi))
    xcd = torch.arange(nblk) % 8
    for x in range(8):
        sel = xcd == x
        ds = d[sel]
        span = ds[:, 3].max() - ds[:, 0].min()
        order = torch.argsort(ds[:, 0])
        starts = ds[order, 0] - ds[:, 0].min()
        ends = ds[order, 3] - ds[:, 0].min()
        # gap between a block's end and the start of the block that reuses its CU: approximate by sorting
        # starts of round r+1 against ends of round r (32 CUs per XCD)
        n = ds.shape[0]
        gaps = []
        es = torch.sort(ends)[0]
        for i in range(32, n):
            gaps.append(float(starts[i] - es[i - 32]))
        g = torch.tensor(gaps) if gaps else torch.zeros(1)
        print(f"   xcd {x}: {n:3d} blocks, span {span:9.0f} ticks -> {span / wall_us:7.1f} ticks/us; "
              f"next-block gap mean {g.mean():7.0f} p50 {g.median():7.0f} max {g.max():7.0f}; first-round start spread {starts[:32].max():6.0f}")
