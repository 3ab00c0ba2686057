#!/usr/bin/env python3
"""Round 6 same-process A/Bs on the diagnostic library (kill rules: two A/Bs, <= 12 GPU-minutes each):

  tiles   per-shape tile overrides (slime_gemm_set_shape_tile) on the 20-crop half batch: q/k/v, out_proj, fc1 on the 96-row
          direct-B tile (19) instead of the 128-row one (12), fc2 on the 192-row ping-pong tile (9) instead of 256 (4) --
          stand-alone launch times (HIP-event probe of the tower driver, mean over the 23 layers) AND the two-stream 40-crop
          tower, which is where a tile has to win (a sub-round launch's idle CUs are taken by the other stream's kernels);
          bit-equality of the tower output against the shipped dispatch.
  attn    attention variant 40: the V^T transposed reads replaced by conflict-free plain reads (wrong results, right timing) --
          the exact price of the 2-way LDS bank conflict, stand-alone at 5 / 20 / 40 crops and inside the two-stream tower.

usage: r6_tile_ab.py [tiles] [attn] [--rounds R]
"""
import hashlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W

dev = torch.device("cuda:0")
lib = _lib.load_diag()
dt = torch.bfloat16
ROUNDS = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
SHAPES = {"qkv": (3072, 1024, 1), "out": (1024, 1024, 3), "fc1": (4096, 1024, 5), "fc2": (1024, 4096, 6)}     # N, K, probe kernel id

tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
side = torch.cuda.Stream()
parts = list(px.chunk(2))


def run2():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        b = ops.tower_forward(pts[1], parts[1])
    a = ops.tower_forward(pts[0], parts[0])
    cur.wait_stream(side)
    return a, b


def tower_ms(reps=12):
    for _ in range(3):
        run2()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run2()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def sha():
    a, b = run2()
    torch.cuda.synchronize()
    return hashlib.sha1(torch.cat([a, b]).float().cpu().numpy().tobytes()).hexdigest()[:12]


def probe_us(kid, crops=20):
    """Mean launch duration of tower kernel `kid` over the 23 layers of one pass over `crops` crops (other stream idle)."""
    pt = pts[0]
    x = px[:crops].contiguous()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record(); torch.cuda.synchronize()
    ms = []
    for layer in range(pt.layers_run):
        pt.probe = (layer, kid, e0, e1)
        ops.tower_forward(pt, x)
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    pt.probe = None
    return sum(ms) / len(ms) * 1e3


def set_tiles(over):
    lib.slime_gemm_set_shape_tile(0, 0, 0)                       # clear the table
    for name, tile in over.items():
        N, K, _ = SHAPES[name]
        lib.slime_gemm_set_shape_tile(N, K, tile)


if "tiles" in sys.argv or len([a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()]) == 0:
    configs = [("shipped", {}), ("out->96", {"out": 19}), ("qkv->96", {"qkv": 19}), ("qkv,out->96", {"qkv": 19, "out": 19}),
               ("qkv,out,fc1->96", {"qkv": 19, "out": 19, "fc1": 19}), ("fc2->192pp", {"fc2": 9}), ("qkv,out->96 fc2->192pp", {"qkv": 19, "out": 19, "fc2": 9})]
    print("== stand-alone launch, 20-crop half batch (us, mean over 23 layers): shipped tile | alternative ==", flush=True)
    for name, alt in (("qkv", 19), ("out", 19), ("fc1", 19), ("fc2", 9)):
        row = []
        for rnd in range(2):
            for tile in (None, alt):
                set_tiles({} if tile is None else {name: tile})
                row.append(probe_us(SHAPES[name][2]))
        N, K, _ = SHAPES[name]
        fl = 2.0 * 11540 * N * K
        print(f"{name:4s}: shipped {row[0]:6.1f} {row[2]:6.1f} us ({fl / row[0] / 1e6:5.0f} TF/s) | tile {alt:2d} {row[1]:6.1f} {row[3]:6.1f} us ({fl / row[1] / 1e6:5.0f} TF/s)", flush=True)
    set_tiles({})
    ref = sha()
    print(f"== two-stream tower, 40 crops (ms), {ROUNDS} interleaved rounds; sha of the features (shipped: {ref}) ==", flush=True)
    res = {n: [] for n, _ in configs}
    shas = {}
    for rnd in range(ROUNDS):
        for n, over in configs:
            set_tiles(over)
            res[n].append(tower_ms())
            if rnd == 0:
                shas[n] = sha()
    for n, _ in configs:
        v = res[n]
        print(f"{n:28s}: " + " ".join(f"{x:6.3f}" for x in v) + f" | median {sorted(v)[len(v) // 2]:6.3f} | bit-equal {shas[n] == ref}", flush=True)
    set_tiles({})

if "attn" in sys.argv:
    print("== attention: attn64r (0) vs variant 40 = V^T transpose reads -> conflict-free plain reads (wrong results, right timing) ==", flush=True)
    for B in (5, 20, 40):
        qkv = torch.randn(B, 577, 3072, device=dev).to(dt)
        qkv[..., :1024] *= 0.125
        q, k, v = qkv[..., :1024], qkv[..., 1024:2048], qkv[..., 2048:]
        rows = {0: [], 40: []}
        for rnd in range(3):
            for var in (0, 40):
                lib.slime_attention_set_variant(var)
                for _ in range(3):
                    ops.attention(q, k, v, 16, 64)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.attention(q, k, v, 16, 64)
                e1.record(); torch.cuda.synchronize()
                rows[var].append(e0.elapsed_time(e1) / 20 * 1e3)
        lib.slime_attention_set_variant(0)
        print(f"B = {B:2d}: attn64r " + " ".join(f"{x:6.1f}" for x in rows[0]) + " us | conflict-free reads " + " ".join(f"{x:6.1f}" for x in rows[40]) +
              f" us | price of the conflict {sorted(rows[0])[1] - sorted(rows[40])[1]:5.1f} us", flush=True)
    res = {0: [], 40: []}
    for rnd in range(ROUNDS):
        for var in (0, 40):
            lib.slime_attention_set_variant(var)
            res[var].append(tower_ms())
    lib.slime_attention_set_variant(0)
    print("two-stream tower, 40 crops (ms): attn64r " + " ".join(f"{x:6.3f}" for x in res[0]) + " | conflict-free reads " + " ".join(f"{x:6.3f}" for x in res[40]), flush=True)
