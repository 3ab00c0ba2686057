#!/usr/bin/env python3
"""Round 3: does wave priority make the direct-B kernel's epilogue (and prologue) overlap with the co-resident workgroup's MFMAs?
Diagnostic bits (slime_gemm_set_db_ablation): 4 = s_setprio 3 from the end of the main loop on, 8 = s_setprio 3 through the
prologue (back to 0 when the main loop starts), 12 = both.  Stand-alone per tower shape and inside the two-stream 40-crop tower,
interleaved rounds, same process."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
SH = {"qkv": (3072, 1024, _lib.EPI_BIAS_T), "out": (1024, 1024, _lib.EPI_BIAS_RESID_F32), "fc1": (4096, 1024, _lib.EPI_BIAS_QUICKGELU_T),
      "fc2": (1024, 4096, _lib.EPI_BIAS_RESID_F32)}
VARIANTS = (0, 4, 8, 12)


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


print("== stand-alone direct-B, TF/s (us), priority bits " + " | ".join(str(v) for v in VARIANTS) + " ==", flush=True)
for M in (11540, 23080):
    for name, (N, K, epi) in SH.items():
        a, w, b = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5), rnd((N,), 3, 1.0, torch.float32)
        wf = ops.pack_b_frag(w)
        out = torch.zeros((M, N), dtype=dt if epi <= _lib.EPI_BIAS_GELU_T else torch.float32, device=dev)
        fl = 2.0 * M * N * K
        for rep in range(2):
            row = []
            for abl in VARIANTS:
                lib.slime_gemm_set_db_ablation(abl)
                row.append(time_ms(lambda: ops.gemm(a, w, b, epi, out=out, w_frag=wf)))
            lib.slime_gemm_set_db_ablation(0)
            print(f"M {M:6d} {name}: " + " | ".join(f"{fl/t/1e9:6.0f} ({t*1e3:5.1f})" for t in row), flush=True)

tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
parts = list(px.chunk(2))


def run2():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for pt, s, p in zip(pts, streams, parts):
        with torch.cuda.stream(s): ops.tower_forward(pt, p)
    for s in streams: cur.wait_stream(s)


def run1():
    ops.tower_forward(pts[0], px)


print("== tower, 40 crops: two streams x 20 | one stream x 40 (ms) ==", flush=True)
for rep in range(4):
    for abl in VARIANTS:
        lib.slime_gemm_set_db_ablation(abl)
        ts = []
        for fn in (run2, run1):
            for _ in range(2): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8): fn()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 8)
        print(f"priority bits {abl:2d}: {ts[0]*1e3:6.2f} ms {40/ts[0]:5.0f} crops/s | {ts[1]*1e3:6.2f} ms {40/ts[1]:5.0f} crops/s", flush=True)
lib.slime_gemm_set_db_ablation(0)
