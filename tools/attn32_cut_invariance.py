#!/usr/bin/env python3
"""Which launch forms of attn32 agree bit for bit?  Uncut (variant 6) vs every item cut in 2..5 (variants 5 / 12..15) on the CLIP
shape, and their timing at 5 / 10 / 20 crops next to attn64r (variant 7).  Diagnostic build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
FORMS = {"attn64r": 7, "uncut": 6, "cut2": 12, "cut3": 13, "cut4": 14, "cut5": 15, "model": 4}


def t_us(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for gain in (1.0, 8.0):
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(3, 577, 3072, generator=g)).to(dt).to(dev)
    x[..., :1024] *= 0.125 * 1.4427 * gain
    outs = {}
    for name, v in FORMS.items():
        lib.slime_attention_set_variant(v)
        outs[name] = ops.attention(x[..., :1024], x[..., 1024:2048], x[..., 2048:], 16, 64)
    torch.cuda.synchronize()
    ref = outs["uncut"]
    print(f"logit gain {gain}: " + "  ".join(f"{n}: {'EQUAL' if torch.equal(o, ref) else 'diff %.3g in %d elems' % (float((o.float() - ref.float()).abs().max()), int((o != ref).sum()))}" for n, o in outs.items()), flush=True)
    print("   cut forms among themselves: " + "  ".join(f"{n}=cut2:{torch.equal(outs[n], outs['cut2'])}" for n in ("cut3", "cut4", "cut5")), flush=True)
for n in (5, 10, 20, 40):
    x = (torch.randn(n, 577, 3072, device=dev) * 0.5).to(dt); x[..., :1024] *= 0.125
    row = []
    for name, v in FORMS.items():
        lib.slime_attention_set_variant(v)
        row.append((name, t_us(lambda: ops.attention(x[..., :1024], x[..., 1024:2048], x[..., 2048:], 16, 64))))
    print(f"{n:2d} crops: " + "  ".join(f"{k} {v:.1f}" for k, v in row) + " us", flush=True)
lib.slime_attention_set_variant(0)
