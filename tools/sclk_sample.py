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
