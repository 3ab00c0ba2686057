#!/usr/bin/env python3
"""Sample rocm-smi clocks/power while (a) the fc1-shaped GEMM, (b) the two-stream tower loops for a few seconds."""
import os, sys, time, subprocess, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); dt = torch.bfloat16
def sample(tag, stop):
    while not stop.is_set():
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
            keep = [l.strip() for l in o.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "junction"))]
            print(tag, " | ".join(keep), flush=True)
        except Exception as e:
            print(tag, "rocm-smi failed", e, flush=True)
        time.sleep(0.5)
def run(tag, fn, secs=4.0):
    stop = threading.Event(); th = threading.Thread(target=sample, args=(tag, stop)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    dtm = (time.time() - t0) / n
    stop.set(); th.join()
    return dtm
print(subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True).stdout[-600:])
M, N, K = 11540, 4096, 1024
a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.03).to(dt); b = torch.zeros(N, device=dev); c = torch.empty(M, N, device=dev, dtype=dt)
t = run("[gemm fc1]", lambda: ops.gemm(a, w, b, 1, out=c)); print(f"gemm fc1: {t*1e6:.1f} us {2*M*N*K/t/1e12:.0f} TF/s")
M, N, K = 16384, 4096, 16384
a = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.01).to(dt); b = torch.zeros(N, device=dev); c = torch.empty(M, N, device=dev, dtype=dt)
t = run("[gemm K=16384]", lambda: ops.gemm(a, w, b, 0, out=c)); print(f"gemm big: {t*1e6:.1f} us {2*M*N*K/t/1e12:.0f} TF/s")

# the two-stream tower itself
from slime_amd import weights as W
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
parts = list(px.chunk(2))
def tower2():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for pt, s, p in zip(pts, streams, parts):
        with torch.cuda.stream(s): ops.tower_forward(pt, p)
    for s in streams: cur.wait_stream(s)
def run2(tag, fn, secs=4.0):
    stop = threading.Event(); th = threading.Thread(target=sample, args=(tag, stop)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(5): fn()
        torch.cuda.synchronize(); n += 5
    dtm = (time.time() - t0) / n
    stop.set(); th.join()
    return dtm
t = run2("[tower 2 streams]", tower2); print(f"tower: {t*1e3:.2f} ms {40/t:.0f} crops/s")
