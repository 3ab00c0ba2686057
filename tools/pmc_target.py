#!/usr/bin/env python3
"""Small profiling target: a few launches of each hot kernel at the bench launch shapes (single stream)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W

dev = torch.device("cuda:0")
lib = _lib.load_diag()
dt = torch.bfloat16
n = int(os.environ.get("CROPS", "20"))
M = n * 577
shapes = {"qkv": (3072, 1024, _lib.EPI_BIAS_T), "out": (1024, 1024, _lib.EPI_BIAS_RESID_F32),
          "fc1": (4096, 1024, _lib.EPI_BIAS_QUICKGELU_T), "fc2": (1024, 4096, _lib.EPI_BIAS_RESID_F32)}
for name, (N, K, epi) in shapes.items():
    a = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    b = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi >= _lib.EPI_BIAS_F32 else dt)
    for _ in range(4):
        ops.gemm(a, w, b, epi, out=out)
qkv = torch.randn(n, 577, 3072, device=dev).to(dt)
qkv[..., :1024] *= 0.125
for _ in range(4):
    ops.attention(qkv[..., :1024], qkv[..., 1024:2048], qkv[..., 2048:], 16, 64)
x = torch.randn(M, 1024, device=dev)
wln = torch.ones(1024, device=dev)
for _ in range(4):
    ops.layernorm(x, wln, wln, 1e-5, dt)
torch.cuda.synchronize()
print("pmc target done")
