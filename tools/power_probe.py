#!/usr/bin/env python3
"""Round 4: is the two-stream tower bound by the 1400 W power cap?  The same launch sequence over (a) the synthetic weights / pixels
of the bench and (b) all-zero weights and pixels (identical instruction streams, no operand switching activity), with rocm-smi
power / sclk sampled meanwhile.  If (b) is much faster at lower power, the step's currency is energy, not issue slots."""
import os, sys, time, subprocess, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, weights as W
dev = torch.device("cuda:0"); dt = torch.bfloat16


def sample(tag, stop, acc):
    while not stop.is_set():
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
            sclk = [l for l in o.splitlines() if "sclk" in l]; pw = [l for l in o.splitlines() if "(W)" in l]
            acc.append((sclk[0].split("(")[-1].split(")")[0] if sclk else "?", pw[0].split(":")[-1].strip() if pw else "?"))
        except Exception as e:
            acc.append(("?", repr(e)[:40]))
        time.sleep(0.4)


def tower_pair(tsd, px):
    pts = [ops.pack_tower(W.strip_tower_prefix(tsd), W.CLIP_L_336, dt, dev) for _ in range(2)]
    side = torch.cuda.Stream(); parts = list(px.chunk(2))

    def run2():
        cur = torch.cuda.current_stream(); side.wait_stream(cur)
        with torch.cuda.stream(side): ops.tower_forward(pts[1], parts[1])
        ops.tower_forward(pts[0], parts[0]); cur.wait_stream(side)

    def run1():
        ops.tower_forward(pts[0], px)
    return run2, run1


def measure(tag, fn, secs=4.0):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    acc, stop = [], threading.Event(); th = threading.Thread(target=sample, args=(tag, stop, acc)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(5): fn()
        torch.cuda.synchronize(); n += 5
    ms = (time.time() - t0) / n * 1e3
    stop.set(); th.join()
    print(f"{tag:34s}: {ms:6.2f} ms per 40 crops; rocm-smi (sclk, power): {acc[1:-1][:8]}", flush=True)


tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
zsd = {k: torch.zeros_like(v) for k, v in tsd.items()}
for rnd in range(2):
    r2, r1 = tower_pair(tsd, px)
    measure("bench operands, two streams", r2); measure("bench operands, one stream", r1)
    z2, z1 = tower_pair(zsd, torch.zeros_like(px))
    measure("all-zero operands, two streams", z2); measure("all-zero operands, one stream", z1)
