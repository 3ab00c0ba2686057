#!/usr/bin/env python3
"""One tower pass over n crops on one stream vs split in two halves on two streams, n = 2 ... 48 (product library, bf16, interleaved).
The table behind HipCLIPVisionModel.encode's split policy (slime_amd/model/multimodal_encoder/clip_encoder.py: TWO_STREAM_COUNTS)."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import weights as W
from slime_amd.model.multimodal_encoder.clip_encoder import HipCLIPVisionModel
dev = torch.device("cuda:0"); dt = torch.bfloat16
vm = HipCLIPVisionModel(W.CLIP_L_336); vm.load_state_dict(W.make_tower_state_dict(W.CLIP_L_336, seed=1234)); vm.to(dev).to(dt)
px = W.synthetic_pixels(48, seed=0).to(dev).to(dt)


def t_ms(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3


res = {}
for n in list(range(2, 41)) + [44, 48]:
    x = px[:n].contiguous()
    one, two = [], []
    for rep in range(2):
        vm.force_streams = 1; one.append(t_ms(lambda: vm.encode(x)))
        vm.force_streams = 2; two.append(t_ms(lambda: vm.encode(x)))
    vm.force_streams = 0
    res[n] = (round(min(one), 3), round(min(two), 3))
    print(f"{n:2d} crops: one stream {min(one):6.3f} ms   two streams {min(two):6.3f} ms   -> {'two' if min(two) < 0.985 * min(one) else 'one'}", file=sys.stderr, flush=True)
print(json.dumps({"ms_one_two": res, "two_streams_when_faster_by_1.5pct": [n for n, (a, b) in res.items() if b < 0.985 * a]}))
