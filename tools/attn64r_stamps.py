#!/usr/bin/env python3
"""Round 3: how much of an attn64r workgroup's life is start-up (first instruction -> first K/V granule ready) at the tower's batch
sizes, as a function of how many K/V granules are requested ahead (rounds 1-2: the whole panel up front)?  One s_memtime record per
workgroup (diagnostic build), kernel times, bit-equality, and the two-stream 40-crop tower with each depth."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
H, S, DH = 16, 577, 64
VARIANTS = ((27, "all 10 granules up front"), (26, "6 ahead"), (24, "4 ahead"), (0, "3 ahead (product)"), (22, "2 ahead"))
for B in (5, 10, 20, 40):
    qkv = (torch.randn(B, S, 3 * H * DH, device=dev) * 0.5).to(dt)
    o = torch.empty((B, S, H * DH), dtype=dt, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    N = 3 * H * DH
    def run():
        _lib.check(lib.slime_attention(qkv.data_ptr(), S * N, N, qkv.data_ptr() + H * DH * 2, S * N, N, qkv.data_ptr() + 2 * H * DH * 2, S * N, N,
                                      o.data_ptr(), S * H * DH, H * DH, B, H, DH, S, S, ops.dtype_code(dt), st))
    ref = None
    for var, name in VARIANTS:
        lib.slime_attention_set_variant(var); lib.slime_attention_set_debug(None)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        t_us = e0.elapsed_time(e1) * 50
        if ref is None: ref = o.clone()
        same = bool(torch.equal(ref, o))
        cnt = torch.zeros(8 * 2 * B * H, dtype=torch.int64, device=dev)
        lib.slime_attention_set_variant(17 if var == 0 else var)   # 17 = the product kernel with records; lib.slime_attention_set_debug(cnt.data_ptr())
        for _ in range(4): run()
        torch.cuda.synchronize()
        cnt.zero_(); torch.cuda.synchronize()
        run(); torch.cuda.synchronize()
        lib.slime_attention_set_variant(0); lib.slime_attention_set_debug(None)
        rec = cnt.cpu().view(-1, 8)
        rec = rec[rec[:, 0] != 0]
        su, p1, life = (rec[:, 1] - rec[:, 0]).float(), (rec[:, 2] - rec[:, 1]).float(), (rec[:, 3] - rec[:, 0]).float()
        print(f"B={B:2d} {name:26s}: kernel {t_us:6.1f} us, bit-equal {same}; {rec.shape[0]} workgroups; wave 0: start-up {su.mean():6.0f} ticks "
              f"(median {su.median():6.0f}, max {su.max():6.0f}) = {100*su.sum()/life.sum():4.1f} % of its life ({life.mean():6.0f}); pass 1 {p1.mean():6.0f}", flush=True)

# the two-stream 40-crop tower with each depth
from slime_amd import weights as W
import time
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
parts = list(px.chunk(2))


def run2():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for pt, s, p in zip(pts, streams, parts):
        with torch.cuda.stream(s): ops.tower_forward(pt, p)
    for s in streams: cur.wait_stream(s)


print("== two-stream tower, 40 crops (ms) ==", flush=True)
for rep in range(3):
    row = []
    for var, name in VARIANTS:
        lib.slime_attention_set_variant(var)
        for _ in range(2): run2()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): run2()
        torch.cuda.synchronize(); row.append((time.perf_counter() - t0) / 8 * 1e3)
    print("   " + " | ".join(f"{name}: {t:6.2f}" for (v, name), t in zip(VARIANTS, row)), flush=True)
lib.slime_attention_set_variant(0)
