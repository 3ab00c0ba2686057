#!/usr/bin/env python3
"""Causal GQA prefill attention (32 query / 8 kv heads, dh 128) at the shapes of bench.py --config 4 (8 sequences x 1216) and
--config 5 (one sequence x 9280): prefill32 (one wave per SIMD, 32x32x16 MFMAs, LDS-DMA ring; diagnostic variant 0 = product
dispatch: persistent over its work items since round 3) against the same kernel with one item per workgroup (variant 2: the
round-2 launch; must be bit-equal) and the eight-wave kernel (variant 1), same call, interleaved."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
HQ, HKV = 32, 8
N = (HQ + 2 * HKV) * 128
NAMES = {0: 'prefill32 persistent', 3: 'prefill32 4 ahead   ', 2: 'prefill32 1 item/wg ', 1: 'eight-wave          '}
for B, S in ((8, 1216), (1, 9280), (4, 4096), (2, 1216), (8, 700)):
    qkv = (torch.randn(B, S, N, device=dev) * 0.5).to(dt); qkv[..., :HQ * 128] *= 0.1
    o = torch.empty((B, S, HQ * 128), dtype=dt, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        _lib.check(lib.slime_prefill_attention(qkv.data_ptr(), S * N, N, qkv.data_ptr() + HQ * 256, S * N, N, qkv.data_ptr() + (HQ + HKV) * 256, S * N, N,
                                              o.data_ptr(), S * HQ * 128, HQ * 128, B, HQ, HKV, 128, S, None, None, ops.dtype_code(dt), st))
    outs = {}
    fl = 4.0 * B * HQ * (S * (S + 1) / 2) * 128
    for rnd in range(2):
        for var in (0, 3, 2, 1):
            lib.slime_prefill_set_variant(var)
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 10 * 1e-3
            outs[var] = o.clone()
            print(f"B={B} S={S} {NAMES[var]}: {t*1e3:7.3f} ms {fl/t/1e12:7.1f} TF/s", flush=True)
    d = (outs[0].float() - outs[1].float()).norm() / outs[1].float().norm()
    print(f"   rel-L2 prefill32 vs eight-wave: {float(d):.2e}; persistent bit-equal to one item per workgroup: {bool(torch.equal(outs[0], outs[2]))}, to 4 steps ahead: {bool(torch.equal(outs[0], outs[3]))}")
lib.slime_prefill_set_variant(0)
