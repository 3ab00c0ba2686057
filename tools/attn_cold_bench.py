#!/usr/bin/env python3
"""Attention variants on COLD inputs: the timing loop walks over enough distinct qkv buffers (> 512 MB) that nothing is left
in L2 / the 256 MB Infinity Cache from the previous visit -- as in the tower, where qkv was just written by a GEMM that streamed
70 MB through the caches.  Compare with tools/attn32_check.py (one hot buffer)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
for B in (20, 40):
    per = B * 577 * 3072 * 2
    nbuf = max(2, (600 << 20) // per + 1)
    bufs = [torch.randn(B, 577, 3072, device=dev).to(dt) for _ in range(nbuf)]
    for b_ in bufs: b_[..., :1024] *= 0.125
    def call(i):
        x = bufs[i % nbuf]
        return ops.attention(x[..., :1024], x[..., 1024:2048], x[..., 2048:], 16, 64)
    for rnd in range(2):
        for var in (0, 4, 6):
            lib.slime_attention_set_variant(var)
            for i in range(nbuf): call(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 3 * nbuf
            e0.record()
            for i in range(n): call(i)
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / n * 1e-3
            print(f"cold B={B:2d} ({nbuf} buffers) variant {var}: {t*1e6:7.1f} us {4.0*B*16*577*577*64/t/1e12:6.1f} TF/s", flush=True)
lib.slime_attention_set_variant(0)
