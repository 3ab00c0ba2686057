#!/usr/bin/env python3
"""Round 4: attn64g (variant 30: K/V ring, four waves, two workgroups per CU; diagnostic build) against attn64r (variant 0) --
bit-equality, stand-alone timing at the tower's shapes, and the 40-crop tower (two streams x 20, one stream x 40)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); lib = _lib.load_diag()


def timeit(fn, warm=3, it=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for dt in (torch.bfloat16, torch.float16):
    for B in (1, 5, 9, 20, 40):
        qkv = torch.randn(B, 577, 3072, device=dev).to(dt); qkv[..., :1024] *= 0.125
        q, k, v = qkv[..., :1024], qkv[..., 1024:2048], qkv[..., 2048:]
        row, outs = [], {}
        for rnd in range(2 if dt == torch.bfloat16 else 1):
            for var in (0, 30):
                lib.slime_attention_set_variant(var)
                row.append(timeit(lambda: ops.attention(q, k, v, 16, 64)))
                outs[var] = ops.attention(q, k, v, 16, 64)
        lib.slime_attention_set_variant(0)
        print(f"{str(dt)[6:]:8s} B={B:2d}: " + " | ".join(f"{t:6.1f}" for t in row) + f" us (attn64r | attn64g ...)  bit-equal {torch.equal(outs[0], outs[30])}", flush=True)

dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
side = torch.cuda.Stream(); parts = list(px.chunk(2))


def run2():
    cur = torch.cuda.current_stream(); side.wait_stream(cur)
    with torch.cuda.stream(side): b = ops.tower_forward(pts[1], parts[1])
    a = ops.tower_forward(pts[0], parts[0]); cur.wait_stream(side)
    return torch.cat([a, b])


def run1(): return ops.tower_forward(pts[0], px)


lib.slime_attention_set_variant(0); ref = run2(); torch.cuda.synchronize()
print("tower, 40 crops: two streams x 20 | one stream x 40 (ms) | bit-equal")
for rnd in range(3):
    for var in (0, 30):
        lib.slime_attention_set_variant(var)
        eq = torch.equal(run2(), ref)
        ts = []
        for fn in (run2, run1):
            for _ in range(2): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8): fn()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 8 * 1e3)
        print(f"variant {var:2d}: {ts[0]:6.2f} | {ts[1]:6.2f} | {eq}", flush=True)
lib.slime_attention_set_variant(0)
