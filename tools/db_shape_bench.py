#!/usr/bin/env python3
"""Stand-alone A/B of the direct-B kernel vs the LDS-staged dispatch on arbitrary shapes (the adapter's stacked MLP, the Llama-3-8B
projections of configs 4 / 5) and the direct-B kernel's L2 patch height (GROUP_M).  Diagnostic build, hot operands, interleaved."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
E = _lib
SHAPES = [("adapter mlp1", 13824, 4096, 1024, E.EPI_BIAS_GELU_T), ("adapter mlp2", 13824, 4096, 4096, E.EPI_BIAS_F32),
          ("llama qkv cfg4", 9728, 6144, 4096, E.EPI_BIAS_T), ("llama o cfg4", 9728, 4096, 4096, E.EPI_BIAS_T),
          ("llama qkv cfg5", 9280, 6144, 4096, E.EPI_BIAS_T), ("resampler kv", 18432, 1024, 1024, E.EPI_BIAS_T),
          ("tower qkv", 11540, 3072, 1024, E.EPI_BIAS_T), ("tower fc1", 11540, 4096, 1024, E.EPI_BIAS_QUICKGELU_T),
          ("tower out", 11540, 1024, 1024, E.EPI_BIAS_RESID_F32), ("tower fc2", 11540, 1024, 4096, E.EPI_BIAS_RESID_F32)]


def rnd(shape, seed, scale=1.0, dtype=dt):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


def time_ms(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("shape: LDS-staged | direct-B GROUP_M 8 (default) | 4 | 16 | 2     TF/s (us), two rounds")
for name, M, N, K, epi in SHAPES:
    a, w, b = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5), rnd((N,), 3, 1.0, torch.float32)
    wf = ops.pack_b_frag(w)
    out = torch.zeros((M, N), dtype=dt if epi <= E.EPI_BIAS_GELU_T else torch.float32, device=dev)
    fl = 2.0 * M * N * K
    for rep in range(2):
        lib.slime_gemm_force_tile(0)
        row = [time_ms(lambda: ops.gemm(a, w, b, epi, out=out))]
        lib.slime_gemm_force_tile(12)
        for gm in (0, 4, 16, 2):
            lib.slime_gemm_set_group_m(gm)
            row.append(time_ms(lambda: ops.gemm(a, w, b, epi, out=out, w_frag=wf)))
        lib.slime_gemm_set_group_m(0); lib.slime_gemm_force_tile(0)
        print(f"{name:15s} M{M} N{N} K{K}: " + " | ".join(f"{fl/t/1e9:5.0f} ({t*1e3:6.1f})" for t in row), flush=True)
