#!/usr/bin/env python3
"""Attention kernel variants at the tower's shapes, interleaved rounds: 0 = attn64r (one workgroup per (crop, head), granule DMA,
two passes), 2 = round-1 attn64 (two workgroups per (crop, head)), 1 = generic kernel; plus the Resampler shapes."""
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
NAMES = {0: "attn64r", 2: "attn64 (r1)", 3: "attn64w 12wave", 1: "generic"}
for B in (5, 20, 40):
    qkv = torch.randn(B, 577, 3072, device=dev).to(dt); qkv[..., :1024] *= 0.125
    q, k, v = qkv[..., :1024], qkv[..., 1024:2048], qkv[..., 2048:]
    outs = {}
    for rnd in range(3):
        for var in (0, 2, 3, 1):
            if var == 1 and rnd > 0: continue
            lib.slime_attention_set_variant(var)
            t = timeit(lambda: ops.attention(q, k, v, 16, 64))
            outs[var] = ops.attention(q, k, v, 16, 64)
            print(f"attention vit B={B:2d} {NAMES[var]:12s}: {t*1e6:7.1f} us {4.0*B*16*577*577*64/t/1e12:6.1f} TF/s", flush=True)
    lib.slime_attention_set_variant(0)
    print(f"   bit-equal: attn64r vs r1 {torch.equal(outs[0], outs[2])}, attn64w vs r1 {torch.equal(outs[3], outs[2])}")
for B, nq in ((32, 144), (8, 576)):
    qq = (torch.randn(1, nq, 1024, device=dev) * 0.088).to(dt); kk = torch.randn(B, 576, 1024, device=dev).to(dt); vv = torch.randn(B, 576, 1024, device=dev).to(dt)
    t = timeit(lambda: ops.attention(qq, kk, vv, 8, 128))
    print(f"attention resampler B={B} nq={nq}: {t*1e3:.3f} ms {4.0*B*8*nq*576*128/t/1e12:.1f} TF/s", flush=True)
