#!/usr/bin/env python3
"""Round 3: where a prefill32 workgroup's time goes (diagnostic build: s_memtime stamps summed over all workgroups), persistent
launch (variant 0) and one item per workgroup (variant 2), at the shapes of bench.py --config 4 / 5."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
HQ, HKV = 32, 8
N = (HQ + 2 * HKV) * 128
NAMES = ("start-up", "item prologue", "main loop", "next item + Q request", "drain/normalise/store", "wait for Q")
for B, S in ((8, 1216), (1, 9280), (8, 700)):
    qkv = (torch.randn(B, S, N, device=dev) * 0.5).to(dt); qkv[..., :HQ * 128] *= 0.1
    o = torch.empty((B, S, HQ * 128), dtype=dt, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        _lib.check(lib.slime_prefill_attention(qkv.data_ptr(), S * N, N, qkv.data_ptr() + HQ * 256, S * N, N, qkv.data_ptr() + (HQ + HKV) * 256, S * N, N,
                                              o.data_ptr(), S * HQ * 128, HQ * 128, B, HQ, HKV, 128, S, None, None, ops.dtype_code(dt), st))
    for var in (0, 2):
        lib.slime_prefill_set_variant(var)
        lib.slime_prefill_set_debug(None)
        for _ in range(3): run()
        torch.cuda.synchronize()
        cnt = torch.zeros(8, dtype=torch.int64, device=dev)
        lib.slime_prefill_set_debug(cnt.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        lib.slime_prefill_set_debug(None)
        c = cnt.cpu().tolist()
        tot = sum(c[:6])
        print(f"B={B} S={S} variant {var}: kernel {e0.elapsed_time(e1)*1e3:.1f} us, {c[7]} items, {c[6]} steps; ticks per workgroup-sum {tot}: "
              + ", ".join(f"{n} {100.0*v/tot:.1f}%" for n, v in zip(NAMES, c[:6]))
              + f"; main loop {c[2]/max(c[6],1):.1f} ticks/step, per item: prologue {c[1]/c[7]:.0f} finish {c[4]/c[7]:.0f} next+Q {c[3]/c[7]:.0f} waitQ {c[5]/c[7]:.0f} ticks", flush=True)
lib.slime_prefill_set_variant(0)
