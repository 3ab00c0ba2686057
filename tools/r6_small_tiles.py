#!/usr/bin/env python3
"""Round 6: the rank-sized tower passes (5 / 9 crops: what one of 8 GPUs runs in configs 2 / 5 and 3).  Per-shape tile overrides on
the diagnostic build: out_proj (N = K = 1024) and fc2 (N = 1024, K = 4096) on the direct-B kernel at 64 / 96 / 128 rows (tiles 13 /
19 / 12) against the shipped dispatch (128 x 128 ring / two-stage, or direct-B 128): per-kernel launch time (probe, mean of 23 layers),
tower pass latency, bit-equality.  usage: r6_small_tiles.py [crops ...]"""
import hashlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
pt = ops.pack_tower(tsd, W.CLIP_L_336, dt, dev)
SH = {"qkv": (3072, 1024, 1), "out": (1024, 1024, 3), "fc1": (4096, 1024, 5), "fc2": (1024, 4096, 6)}


def set_tiles(over):
    lib.slime_gemm_set_shape_tile(0, 0, 0)
    for name, tile in over.items():
        lib.slime_gemm_set_shape_tile(SH[name][0], SH[name][1], tile)


def probe_us(px, kid):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record(); torch.cuda.synchronize()
    ms = []
    for layer in range(pt.layers_run):
        pt.probe = (layer, kid, e0, e1)
        ops.tower_forward(pt, px, out_dtype=dt); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    pt.probe = None
    return sum(ms) / len(ms) * 1e3


def pass_ms(px, reps=20):
    for _ in range(3): ops.tower_forward(pt, px, out_dtype=dt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ops.tower_forward(pt, px, out_dtype=dt)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3


for n in [int(a) for a in sys.argv[1:]] or [5, 9]:
    px = W.synthetic_pixels(n, seed=n).to(dev).to(dt)
    set_tiles({})
    ref = hashlib.sha1(ops.tower_forward(pt, px, out_dtype=dt).float().cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"== {n} crops (M = {577 * n}); shipped kernels: " + ", ".join(f"{k}: {v}" for k, v in ops.tower_kernel_names(pt, n).items()), flush=True)
    for name in ("out", "fc2", "qkv", "fc1"):
        row = []
        for tile in (None, 13, 19, 12, 15, 3):
            set_tiles({} if tile is None else {name: tile})
            row.append((tile, probe_us(px, SH[name][2])))
        print(f"  {name:4s} launch us: " + "  ".join(f"{'shipped' if t is None else 'tile %d' % t}: {u:6.1f}" for t, u in row), flush=True)
    configs = [("shipped", {}), ("fc2->64db", {"fc2": 13}), ("fc2->96db", {"fc2": 19}), ("out,fc2->64db", {"out": 13, "fc2": 13}), ("out,fc2->96db", {"out": 19, "fc2": 19}),
               ("all four->96db", {"qkv": 19, "out": 19, "fc1": 19, "fc2": 19}), ("all four->64db", {"qkv": 13, "out": 13, "fc1": 13, "fc2": 13})]
    for rnd in range(2):
        parts = []
        for cname, over in configs:
            set_tiles(over)
            ms = pass_ms(px)
            same = hashlib.sha1(ops.tower_forward(pt, px, out_dtype=dt).float().cpu().numpy().tobytes()).hexdigest()[:12] == ref
            parts.append(f"{cname} {ms:.3f}{'' if same else ' (NOT bit-equal)'}")
        print("  pass ms: " + " | ".join(parts), flush=True)
set_tiles({})
