#!/usr/bin/env python3
"""Round 6: is a 5- / 9-crop tower pass bound by the boundaries BETWEEN its 117 launches, or by the launches?
Run under `rocprofv3 --kernel-trace --stats`: the script prints the wall time per pass (HIP events around N passes on one stream);
the profiler's per-kernel totals / calls give the sum of the kernels' own durations per pass.  wall - sum = what the boundaries
(dispatch, drain, fill of the next grid) cost -- the most a persistent per-layer kernel with grid-wide phase barriers could recover,
and that only if its barriers were free.  usage: small_pass_gaps.py [crops ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, weights as W
dev = torch.device("cuda:0"); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
pt = ops.pack_tower(tsd, W.CLIP_L_336, dt, dev)
PASSES = 20
for n in [int(a) for a in sys.argv[1:]] or [5, 9]:
    px = W.synthetic_pixels(n, seed=n).to(dev).to(dt)
    for _ in range(3): ops.tower_forward(pt, px, out_dtype=dt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(PASSES): ops.tower_forward(pt, px, out_dtype=dt)
    e1.record(); torch.cuda.synchronize()
    print(f"crops {n}: wall {e0.elapsed_time(e1) / PASSES * 1e3:.1f} us per pass over {PASSES} timed passes (+3 warm-up passes in the trace)", flush=True)
