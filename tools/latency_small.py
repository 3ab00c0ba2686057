#!/usr/bin/env python3
"""Small-batch latency: BASELINE config 1 (one 336x336 crop, global only) and one SliME image (1+4 crops)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, weights as W
from slime_amd.model.llava_arch import SlimeVisualEncoder, default_slime_config
dev = torch.device("cuda:0"); dt = torch.bfloat16
enc = SlimeVisualEncoder(default_slime_config("synthetic:1234"))
enc.load_visual_state(W.make_tower_state_dict(W.CLIP_L_336, seed=1234), W.make_adapter_state_dict(W.ADAPTER_8B, seed=4321))
enc.to(dev); enc.get_vision_tower().vision_tower.to(dt)
model = enc.get_model(); tower = enc.get_vision_tower()
pg = model.mm_projector.packed(dt); post = model.sampler.post_qformer.packed(576, dt)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
px1 = W.synthetic_pixels(1, seed=1).to(dev).to(dt)
px5 = W.synthetic_pixels(5, seed=2).to(dev).to(dt)
print(f"cfg1: 1 crop, tower only           : {t(lambda: tower(px1)):.2f} ms")
print(f"cfg1: 1 crop, tower + gated project : {t(lambda: ops.adapter_forward(pg, None, tower(px1), 1, 0, 1, 1, False, -1, dt)):.2f} ms")
print(f"one image (1+4 crops), tower + adapter: {t(lambda: ops.adapter_forward(pg, post, tower(px5), 1, 4, 2, 2, True, -1, dt)):.2f} ms")
