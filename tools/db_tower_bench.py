#!/usr/bin/env python3
"""Round 3: the direct-B GEMM kernel (tile 12: fragment-order weights, 128x256 tiles, two workgroups per CU) against the
LDS-staged kernels -- stand-alone per tower shape, and inside the tower (two streams x 20 crops and one stream x 40 crops) with
per-shape overrides (diagnostic build).  Same process, interleaved rounds."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
SH = {"qkv": (3072, 1024, _lib.EPI_BIAS_T), "out": (1024, 1024, _lib.EPI_BIAS_RESID_F32), "fc1": (4096, 1024, _lib.EPI_BIAS_QUICKGELU_T),
      "fc2": (1024, 4096, _lib.EPI_BIAS_RESID_F32)}


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


print("== stand-alone, hot operands, TF/s (us): LDS-staged auto | direct-B | direct-B, epilogue stores kept in L2 | direct-B, no epilogue ==", flush=True)
for M in (11540, 23080):
    for name, (N, K, epi) in SH.items():
        a, w, b = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5), rnd((N,), 3, 1.0, torch.float32)
        wf = ops.pack_b_frag(w)
        out = torch.zeros((M, N), dtype=dt if epi <= _lib.EPI_BIAS_GELU_T else torch.float32, device=dev)
        fl = 2.0 * M * N * K
        for rep in range(2):
            row = [time_ms(lambda: ops.gemm(a, w, b, epi, out=out))]
            for abl in (0, 1, 2):
                lib.slime_gemm_set_db_ablation(abl)
                row.append(time_ms(lambda: ops.gemm(a, w, b, epi, out=out, w_frag=wf)))
            lib.slime_gemm_set_db_ablation(0)
            print(f"M {M:6d} {name}: " + " | ".join(f"{fl/t/1e9:6.0f} ({t*1e3:5.1f})" for t in row), flush=True)

tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
parts = list(px.chunk(2))
FRAGS = ("w_qkv_frag", "w_o_frag", "w_fc1_frag", "w_fc2_frag")
saved = [{f: getattr(pt.desc, f) for f in FRAGS} for pt in pts]


def set_frags(on):
    for pt, sv in zip(pts, saved):
        for f in FRAGS: setattr(pt.desc, f, sv[f] if f in on else None)


def run2():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for pt, s, p in zip(pts, streams, parts):
        with torch.cuda.stream(s): ops.tower_forward(pt, p)
    for s in streams: cur.wait_stream(s)


def run1():
    ops.tower_forward(pts[0], px)


CONFIGS = [("LDS-staged (round 2)", ()), ("direct-B: all four", FRAGS), ("direct-B: qkv fc1", ("w_qkv_frag", "w_fc1_frag")),
           ("direct-B: qkv fc1 out", ("w_qkv_frag", "w_fc1_frag", "w_o_frag")), ("direct-B: qkv fc1 fc2", ("w_qkv_frag", "w_fc1_frag", "w_fc2_frag")),
           ("direct-B: qkv", ("w_qkv_frag",)), ("direct-B: fc1", ("w_fc1_frag",))]
print("== tower, 40 crops: two streams x 20 | one stream x 40 (ms) ==", flush=True)
for rep in range(2):
    for name, on in CONFIGS:
        set_frags(on)
        ts = []
        for fn in (run2, run1):
            for _ in range(2): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8): fn()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 8)
        print(f"{name:26s}: {ts[0]*1e3:6.2f} ms {40/ts[0]:5.0f} crops/s | {ts[1]*1e3:6.2f} ms {40/ts[1]:5.0f} crops/s", flush=True)
set_frags(FRAGS)
