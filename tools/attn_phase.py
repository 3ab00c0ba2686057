#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
B = 16
qkv = torch.randn(B, 577, 3072, device=dev).to(dt); qkv[..., :1024] *= 0.125
q, k, v = qkv[..., :1024], qkv[..., 1024:2048], qkv[..., 2048:]
dbg = torch.zeros(B * 16 * 8 * 4, dtype=torch.int64, device=dev)
lib.slime_attention_set_debug(dbg.data_ptr())
for _ in range(3): ops.attention(q, k, v, 16, 64)
torch.cuda.synchronize(); lib.slime_attention_set_debug(None)
d = dbg.view(B * 16, 8, 4).cpu().double()
stage, comp, fin = d[..., 1] - d[..., 0], d[..., 2] - d[..., 1], d[..., 3] - d[..., 2]
for w in range(8):
    print(f"wave {w}: staging {stage[:, w].mean():8.0f}  compute {comp[:, w].mean():8.0f}  finalize {fin[:, w].mean():7.0f}  total {(d[:, w, 3]-d[:, w, 0]).mean():8.0f}")
