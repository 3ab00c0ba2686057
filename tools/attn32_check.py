#!/usr/bin/env python3
"""attn32 (one wave per SIMD, 32x32x16 MFMAs; diag variants 4 = one workgroup per (crop, head), 5 = two) against an fp32
reference and against attn64r on the clock."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
LOG2E = 1.4426950408889634
def ref_attn(q, k, v, heads, dh):
    B, S, E = k.shape
    qf = q.float().view(q.shape[0], -1, heads, dh).transpose(1, 2) / LOG2E
    kf = k.float().view(B, S, heads, dh).transpose(1, 2); vf = v.float().view(B, S, heads, dh).transpose(1, 2)
    p = torch.softmax(qf @ kf.transpose(-1, -2), dim=-1)
    return (p @ vf).transpose(1, 2).reshape(B, -1, E)
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
def timeit(fn, warm=3, it=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
bad = 0
g = torch.Generator(device="cpu").manual_seed(1)
for (B, heads, nq, nkv, gain, spike) in [(2, 2, 577, 577, 1.0, False), (3, 4, 577, 577, 4.0, False), (3, 4, 577, 577, 12.0, False), (2, 2, 577, 577, 1.0, True),
                                          (1, 1, 577, 321, 1.0, False), (2, 3, 100, 400, 1.0, False), (2, 2, 33, 608, 2.0, False), (1, 2, 640, 576, 1.0, False),
                                          (2, 2, 300, 577, 6.0, False), (5, 16, 577, 577, 1.0, False)]:
    E = heads * 64
    qkv = torch.randn(B, max(nq, nkv), 3 * E, generator=g).to(dt)
    qkv[..., :E] *= gain * 0.125 * LOG2E
    if spike: qkv[0, 500, E:2 * E] = (qkv[0, 17, :E].float() * 60.0).to(dt)
    qkv = qkv.to(dev)
    q, k, v = qkv[:, :nq, :E], qkv[:, :nkv, E:2 * E], qkv[:, :nkv, 2 * E:]
    want = ref_attn(q, k, v, heads, 64)
    line = f"B={B} heads={heads} nq={nq} nkv={nkv} gain={gain} spike={spike}:"
    for var in (0, 4, 6):
        lib.slime_attention_set_variant(var)
        out = ops.attention(q, k, v, heads, 64)
        torch.cuda.synchronize()
        r = rel(out.float(), want); fin = bool(torch.isfinite(out.float()).all())
        line += f"  v{var} {r:.2e}{'' if fin else ' NONFINITE'}"
        if not fin or r > 6e-3: bad += 1
    print(line, flush=True)
lib.slime_attention_set_variant(0)
print("FAILURES:", bad, flush=True)
for B in (5, 10, 20, 40):
    qkv = torch.randn(B, 577, 3072, device=dev).to(dt); qkv[..., :1024] *= 0.125
    q, k, v = qkv[..., :1024], qkv[..., 1024:2048], qkv[..., 2048:]
    for rnd in range(2):
        for var in (0, 4, 6):
            lib.slime_attention_set_variant(var)
            t = timeit(lambda: ops.attention(q, k, v, 16, 64))
            print(f"B={B:2d} variant {var}: {t*1e6:7.1f} us {4.0*B*16*577*577*64/t/1e12:6.1f} TF/s", flush=True)
lib.slime_attention_set_variant(0)
