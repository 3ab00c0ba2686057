#!/usr/bin/env python3
"""slime_patch_embed_prenorm stand-alone: time per launch at 5 / 20 / 40 crops (bf16 pixels, ViT-L/14-336), split-stream outputs.
usage: python tools/patch_embed_bench.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, weights as W

dev, dt = torch.device("cuda:0"), torch.bfloat16
cfg = W.CLIP_L_336
pt = ops.pack_tower(W.make_tower_state_dict(cfg, seed=1234), cfg, dt, dev, select_layer=0)
T = pt.tensors
for n in (5, 20, 40):
    px = W.synthetic_pixels(n, seed=3).to(dev).to(dt)
    run = lambda: ops.patch_embed_prenorm(px, T["patch_w_frag"], T["cls"], T["pos"], T["pre_ln_w"], T["pre_ln_b"], cfg.layer_norm_eps, dt,
                                          cfg.image_size, cfg.patch_size, pt.desc.kpad, want_h=False, want_lo=True)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    gf = 2.0 * n * 576 * 588 * 1024 / 1e9
    print(f"patch_embed_prenorm {n:3d} crops: {us:7.1f} us per launch (incl. 3 output allocations), {gf / us * 1e-3:6.1f} TF/s on the conv's {gf:.1f} GF; "
          f"round 4's three launches: 84 us at 20 crops")
