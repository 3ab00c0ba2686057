#!/usr/bin/env python3
"""where does a tower kernel's time go AT FULL CLOCK?  Every per-layer kernel of the 20-crop half batch
(M = 11540) is timed stand-alone on ALL-ZERO operands (no operand switching: the chip keeps 2.40 GHz, profiles/r04_power_cap.txt)
and on random operands (the power-capped clock), and the zero-operand launch time is split into

  mfma      : FLOPs / 2.04 PF -- the bare MFMA stream at full clock (profiles/r04_power_cap.txt: fc1's MFMAs alone 95.1 us for 193.7 GF)
  tail idle : grid quantisation -- (1 - workgroups / (rounds x resident slots)) of the launch
  epilogue  : launch time minus the same launch with the epilogue compiled out of the stream (diagnostic build, db ablation 2; the
              ping-pong kernel has no such switch: fc2's epilogue is priced on the direct-B kernel forced onto its shape, same code)
  rest      : prologue (first DMA round trip, LayerNorm row table), operand delivery in the main loop, barriers

usage: python tools/full_clock_budget.py   (diagnostic library; ~1 minute)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import _lib, ops
lib = _lib.load_diag()
dev, dt = torch.device("cuda:0"), torch.bfloat16
M, D, F, S, H = 20 * 577, 1024, 4096, 577, 16
MFMA_FULL_CLOCK_PF = 2.04
CUS = torch.cuda.get_device_properties(0).multi_processor_count


def timed(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


def operands(N, K, zero):
    g = torch.Generator().manual_seed(N * 7 + K)
    mk = (lambda *sh, s=1.0: torch.zeros(sh)) if zero else (lambda *sh, s=1.0: torch.randn(sh, generator=g) * s)
    a = mk(M, K).to(dt).to(dev)
    w = mk(N, K, s=K ** -0.5).to(dt).to(dev)
    return a, ops.pack_b_frag(w), mk(N, s=0.02).float().to(dev)


def consumer(N, epi, zero):
    a, wf, bias = operands(N, D, zero)
    xr = a.float().view(M, D // 64, 64)
    stats = torch.stack([xr.sum(-1), (xr * xr).sum(-1)], -1).contiguous()
    colsum = torch.zeros(N, device=dev) if zero else torch.randn(N, device=dev) * 0.1
    return lambda: ops.gemm_ln_consumer(a, stats, None, bias, colsum, 1e-5, epi, w_frag=wf)


def producer(K, zero):
    a, wf, bias = operands(D, K, zero)
    h = torch.zeros(M, D) if zero else torch.randn(M, D)
    hi, lo = ops.resid_split(h.to(dev), dt)                      # the split residual stream: T + one signed byte (ABI 7)
    return lambda: ops.gemm_resid_split(a, None, bias, hi, lo, w_frag=wf)


def attention(zero):
    qkv = (torch.zeros(20, S, 3 * D) if zero else torch.randn(20, S, 3 * D) * 0.5).to(dt).to(dev)
    return lambda: ops.attention(qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:], H, 64)


def quant_idle(tiles, slots):
    rounds = -(-tiles // slots)
    return 1.0 - tiles / (rounds * slots)


rows = []
KERNELS = [("qkv_proj (direct-B, LN-fold)", lambda z: consumer(3 * D, _lib.EPI_BIAS_T, z), 2.0 * M * 3 * D * D, 91 * 12, 2 * CUS, True, None),
           ("attention (attn64r)", attention, 4.0 * 20 * H * S * S * 64, None, None, False, None),
           ("out_proj + split residual (direct-B)", lambda z: producer(D, z), 2.0 * M * D * D, 91 * 4, 2 * CUS, True, None),
           ("fc1 + quick-GELU (direct-B, LN-fold)", lambda z: consumer(F, _lib.EPI_BIAS_QUICKGELU_T, z), 2.0 * M * F * D, 91 * 16, 2 * CUS, True, None),
           ("fc2 + split residual (ping-pong)", lambda z: producer(F, z), 2.0 * M * D * F, 46 * 4, CUS, False, 12)]
print(f"20-crop half batch (M = {M}), stand-alone launches, us per launch; {CUS} CUs; mfma = FLOPs / {MFMA_FULL_CLOCK_PF} PF")
print(f"{'kernel':40s} {'real':>7s} {'zero':>7s} | {'mfma':>6s} {'idle':>6s} {'epilog':>6s} {'rest':>6s} | x46 launches: zero ms, mfma ms")
tot = {"real": 0.0, "zero": 0.0, "mfma": 0.0, "idle": 0.0, "epi": 0.0, "rest": 0.0}
for name, make, flops, tiles, slots, db_abl_ok, force_tile in KERNELS:
    t_real = timed(make(False))
    fz = make(True)
    t_zero = timed(fz)
    mfma = flops / (MFMA_FULL_CLOCK_PF * 1e15) * 1e6
    idle = t_zero * quant_idle(tiles, slots) if tiles else 0.0
    epi = 0.0
    if db_abl_ok or force_tile:
        if force_tile: lib.slime_gemm_force_tile(force_tile)
        try:
            with_e = timed(fz)
            lib.slime_gemm_set_db_ablation(2)
            without = timed(fz)
        finally:
            lib.slime_gemm_set_db_ablation(0)
            lib.slime_gemm_force_tile(0)
        epi = max(with_e - without, 0.0)
    rest = t_zero - mfma - idle - epi
    for k, v in (("real", t_real), ("zero", t_zero), ("mfma", mfma), ("idle", idle), ("epi", epi), ("rest", rest)): tot[k] += v
    print(f"{name:40s} {t_real:7.1f} {t_zero:7.1f} | {mfma:6.1f} {idle:6.1f} {epi:6.1f} {rest:6.1f} | {t_zero * 46 / 1e3:5.2f} {mfma * 46 / 1e3:5.2f}")
print(f"{'per layer half (5 launches)':40s} {tot['real']:7.1f} {tot['zero']:7.1f} | {tot['mfma']:6.1f} {tot['idle']:6.1f} {tot['epi']:6.1f} {tot['rest']:6.1f} |"
      f" {tot['zero'] * 46 / 1e3:5.2f} {tot['mfma'] * 46 / 1e3:5.2f}")
print(f"x 46 (23 layers x 2 half batches), serialised: real {tot['real'] * 46 / 1e3:.2f} ms, zero operands {tot['zero'] * 46 / 1e3:.2f} ms = "
      f"mfma {tot['mfma'] * 46 / 1e3:.2f} + tail idle {tot['idle'] * 46 / 1e3:.2f} + epilogues {tot['epi'] * 46 / 1e3:.2f} + rest {tot['rest'] * 46 / 1e3:.2f}")
print("(two streams overlap the tail idle and part of the epilogues of one half batch with the other's kernels: the two-stream tower is shorter than this sum)")
