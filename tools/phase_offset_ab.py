#!/usr/bin/env python3
"""Do the two tower streams interfere because they run the SAME kernel at the same time (two attention launches cannot share a CU: 152 KiB
of LDS each; two fc2 launches both want the matrix pipes)?  Delay the side stream's half batch by a fraction of a layer (torch.cuda._sleep
on that stream: one spinning wave) and time the 40-crop step, rounds interleaved.  A layer of one 20-crop stream is ~370 us."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, weights as W
dev = torch.device("cuda:0"); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
side = torch.cuda.Stream(); parts = list(px.chunk(2))
HZ = 100e6            # s_memtime / _sleep tick on gfx950 (100 MHz constant clock); calibrated below


def run(delay_ticks):
    cur = torch.cuda.current_stream(); side.wait_stream(cur)
    with torch.cuda.stream(side):
        if delay_ticks: torch.cuda._sleep(delay_ticks)
        ops.tower_forward(pts[1], parts[1])
    ops.tower_forward(pts[0], parts[0]); cur.wait_stream(side)


# calibrate _sleep: time 10^6 ticks
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(2_000_000); e1.record(); torch.cuda.synchronize()
us_per_tick = e0.elapsed_time(e1) * 1e3 / 2_000_000
print(f"_sleep: {us_per_tick*1e3:.2f} ns per tick")
delays = [0, 45, 90, 135, 180, 225, 270, 320]
print("delay of the side stream (us): ms per 40 crops (minus the delay itself)")
for rnd in range(3):
    row = []
    for d in delays:
        ticks = int(d / us_per_tick)
        for _ in range(3): run(ticks)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(12): run(ticks)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 12 * 1e3
        row.append(f"{d:3d}: {ms:6.2f} ({ms - d * 1e-3:6.2f})")
    print(" | ".join(row), flush=True)
