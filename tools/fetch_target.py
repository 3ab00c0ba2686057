#!/usr/bin/env python3
"""FETCH_SIZE target: the tower over one 20-crop half batch (two passes) through a product-library variant (VARIANT env; 'product' =
slime_amd/libslime_hip.so).  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`; tools/fetch_summary.py reduces the CSVs."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slime_amd import _lib
name = os.environ.get("VARIANT", "product")
if name != "product":
    _lib.LIB_PATH = os.path.join(ROOT, "slime_amd", "variants", f"libslime_hip_{name}.so")
from slime_amd import ops, weights as W
dev = torch.device("cuda:0"); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
pt = ops.pack_tower(tsd, W.CLIP_L_336, dt, dev)
px = W.synthetic_pixels(int(os.environ.get("CROPS", "20")), seed=0).to(dev).to(dt)
for _ in range(2): ops.tower_forward(pt, px)
torch.cuda.synchronize(); print("fetch target done", name)
