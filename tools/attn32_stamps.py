#!/usr/bin/env python3
"""s_memtime stamps of attn32 (diag variant 4/5): where a workgroup's time goes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
qkv = torch.randn(B, 577, 3072, device=dev).to(dt); qkv[..., :1024] *= 0.125
q, k, v = qkv[..., :1024], qkv[..., 1024:2048], qkv[..., 2048:]
for var, abl in ((4, 0), (6, 0)):
    lib.slime_attention_set_variant(var); lib.slime_attention_set_ablation(abl)
    for _ in range(3): ops.attention(q, k, v, 16, 64)
    nwg = 16 * B * 5
    buf = torch.zeros(nwg * 4 * 32, dtype=torch.int64, device=dev)
    lib.slime_attention_set_debug(buf.data_ptr())
    ops.attention(q, k, v, 16, 64); torch.cuda.synchronize()
    lib.slime_attention_set_debug(None)
    t = buf.view(nwg, 4, 32).cpu().double()
    used = t[:, 0, 0] != 0
    t = t[used]; nwg = int(used.sum())
    span = (t[:, :, 31].max() - t[:, :, 0].min()).item()
    print(f'launch span (first start to last store): {span:.0f} cycles')
    t0 = t[:, :, 0:1]
    rel = t - t0
    def col(i): return rel[:, :, i].mean().item(), rel[:, :, i].min().item(), rel[:, :, i].max().item()
    print(f"variant {var} abl {abl}: {nwg} workgroups; cycles since wave start (mean/min/max over waves)")
    for name, i in (("dma issued", 1), ("granule 0", 3), ("step 0", 4), ("step 17", 21), ("step 18", 22), ("loop+pad", 30), ("stored", 31)):
        m, lo, hi = col(i); print(f"  {name:12s} {m:9.0f} {lo:9.0f} {hi:9.0f}")
    d = (t[:, :, 5:23] - t[:, :, 4:22])
    print("  per-step deltas (mean over waves), steps 1..18:", [int(x) for x in d.mean(dim=(0, 1)).tolist()])
    for wv in range(4):
        dd = d[:, wv, 2:16].mean(dim=1)          # mean step time per (workgroup, this wave), steps 3..16
        vals = sorted(set(int(round(x / 50.0) * 50) for x in dd.tolist()))
        print(f"  wave {wv}: step-time classes (cycles, rounded to 50): {vals[:12]}  mean {dd.mean().item():.0f}")
    print("  first-wave start spread over workgroups (cycles):", int((t[:, 0, 0].max() - t[:, 0, 0].min()).item()))
lib.slime_attention_set_variant(0); lib.slime_attention_set_ablation(0)
