#!/usr/bin/env python3
"""Profiling target (PRODUCT library): the tower's real kernel sequence over one 20-crop half batch -- every hot kernel at its
production launch shape, 46 dispatches each -- plus the fused adapter of the bench step and the prefill attention."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W

dev = torch.device("cuda:0")
lib = _lib.load()
dt = torch.bfloat16
n = int(os.environ.get("CROPS", "20"))
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
pt = ops.pack_tower(tsd, W.CLIP_L_336, dt, dev)
px = W.synthetic_pixels(n, seed=0).to(dev).to(dt)
for _ in range(2):
    feats = ops.tower_forward(pt, px)
torch.cuda.synchronize()
asd = W.make_adapter_state_dict(W.ADAPTER_8B, seed=4321)
pg = ops.pack_gated(W.sub_state(asd, "mm_projector."), W.ADAPTER_8B, dt, dev)
post = ops.pack_resampler(W.sub_state(asd, "sampler.post_qformer."), 1024, 8, 576, dt, dev, W.ADAPTER_8B.ln_eps)
ops.adapter_forward(pg, post, feats, n // 5, 4, 2, 2, True, -1, dt)
B, S, HQ, HKV = 8, 1216, 32, 8
N = (HQ + 2 * HKV) * 128
qkv = (torch.randn(B, S, N, device=dev) * 0.3).to(dt)
o = torch.empty((B, S, HQ * 128), dtype=dt, device=dev)
for _ in range(4):
    _lib.check(lib.slime_prefill_attention(qkv.data_ptr(), S * N, N, qkv.data_ptr() + HQ * 256, S * N, N, qkv.data_ptr() + (HQ + HKV) * 256,
                                           S * N, N, o.data_ptr(), S * HQ * 128, HQ * 128, B, HQ, HKV, 128, S, None, None, _lib.BF16,
                                           torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
# the video-length sequence of config 5
B, S = 1, 9280
qkv = (torch.randn(B, S, N, device=dev) * 0.3).to(dt)
o = torch.empty((B, S, HQ * 128), dtype=dt, device=dev)
for _ in range(3):
    _lib.check(lib.slime_prefill_attention(qkv.data_ptr(), S * N, N, qkv.data_ptr() + HQ * 256, S * N, N, qkv.data_ptr() + (HQ + HKV) * 256,
                                           S * N, N, o.data_ptr(), S * HQ * 128, HQ * 128, B, HQ, HKV, 128, S, None, None, _lib.BF16,
                                           torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print("pmc target done")
