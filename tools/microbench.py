#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X (diagnostic; bench.py is the contract benchmark)."""
import json
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W   # noqa: E402


def timeit(fn, warm=3, it=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load_diag()
    res = {}
    dt = torch.bfloat16
    n_crops = int(os.environ.get("CROPS", "40"))
    M = n_crops * 577
    shapes = {"qkv": (M, 3072, 1024), "out": (M, 1024, 1024), "fc1": (M, 4096, 1024), "fc2": (M, 1024, 4096)}
    epi = {"qkv": _lib.EPI_BIAS_T, "out": _lib.EPI_BIAS_RESID_F32, "fc1": _lib.EPI_BIAS_QUICKGELU_T, "fc2": _lib.EPI_BIAS_RESID_F32}
    for name, (m, n, k) in shapes.items():
        a = torch.randn(m, k, device=dev).to(dt)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(dt)
        b = torch.randn(n, device=dev)
        out = torch.zeros(m, n, device=dev, dtype=torch.float32 if epi[name] >= _lib.EPI_BIAS_F32 else dt)
        for tile, sched in ((1, 1), (3, 1), (4, 1)):
            if True:
                lib.slime_gemm_force_tile(tile)
                lib.slime_gemm_set_sched(sched)
                t = timeit(lambda: ops.gemm(a, w, b, epi[name], out=out))
                tf = 2.0 * m * n * k / t / 1e12
                res[f"gemm_{name}_tile{tile}_sched{sched}"] = {"ms": t * 1e3, "tflops": tf}
                print(f"gemm {name:4s} M={m} N={n} K={k} tile={tile} sched={sched}: {t*1e3:8.3f} ms {tf:8.1f} TF/s", flush=True)
    lib.slime_gemm_force_tile(0)
    lib.slime_gemm_set_sched(1)
    # attention, CLIP geometry
    qkv = torch.randn(n_crops, 577, 3 * 1024, device=dev).to(dt)
    qkv[..., :1024] *= 0.125
    q, k, v = qkv[..., :1024], qkv[..., 1024:2048], qkv[..., 2048:]
    t = timeit(lambda: ops.attention(q, k, v, 16, 64))
    fl = 4.0 * n_crops * 16 * 577 * 577 * 64
    res["attn_vit"] = {"ms": t * 1e3, "tflops": fl / t / 1e12}
    print(f"attention vit B={n_crops}: {t*1e3:.3f} ms {fl/t/1e12:.1f} TF/s", flush=True)
    # layernorm
    x = torch.randn(M, 1024, device=dev)
    wln = torch.ones(1024, device=dev)
    t = timeit(lambda: ops.layernorm(x, wln, wln, 1e-5, dt))
    res["layernorm"] = {"ms": t * 1e3, "GBps": M * 1024 * 6 / t / 1e9}
    print(f"layernorm rows={M}: {t*1e3:.3f} ms {M*1024*6/t/1e9:.0f} GB/s", flush=True)
    # tower end to end
    tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
    pt = ops.pack_tower(tsd, W.CLIP_L_336, dt, dev)
    px = W.synthetic_pixels(n_crops, seed=0).to(dev).to(dt)
    t = timeit(lambda: ops.tower_forward(pt, px), warm=2, it=5)
    res["tower"] = {"ms": t * 1e3, "crops_per_s": n_crops / t, "tflops": n_crops * 366.034e9 / t / 1e12}
    print(f"tower N={n_crops}: {t*1e3:.2f} ms  {n_crops/t:.1f} crops/s  {n_crops*366.034e9/t/1e12:.1f} TF/s", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/microbench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
