#!/usr/bin/env python3
"""Round 4: is the two-stream tower bound by the 1400 W power cap?  The same launch sequence over (a) the synthetic weights / pixels
of the bench and (b) all-zero weights and pixels (identical instruction streams, no operand switching activity), with rocm-smi
power / sclk sampled meanwhile.  If (b) is much faster at lower power, the step's currency is energy, not issue slots."""
import os, sys, time, subprocess, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, weights as W
dev = torch.device("cuda:0"); dt = torch.bfloat16


def sample(tag, stop, acc):
    while not stop.is_set():
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
            sclk = [l for l in o.splitlines() if "sclk" in l]; pw = [l for l in o.splitlines() if "(W)" in l]
            acc.append((sclk[0].split("(")[-1].split(")")[0] if sclk else "?", pw[0].split(":")[-1].strip() if pw else "?"))
        except Exception as e:
            acc.append(("?", repr(e)[:40]))
        time.sleep(0.4)


def tower_pair(tsd, px):
    pts = [ops.pack_tower(W.strip_tower_prefix(tsd), W.CLIP_L_336, dt, dev) for _ in range(2)]
    side = torch.cuda.Stream(); parts = list(px.chunk(2))

    def run2():
        cur = torch.cuda.current_stream(); side.wait_stream(cur)
        with torch.cuda.stream(side): ops.tower_forward(pts[1], parts[1])
        ops.tower_forward(pts[0], parts[0]); cur.wait_stream(side)

    def run1():
        ops.tower_forward(pts[0], px)
    return run2, run1


def measure(tag, fn, secs=4.0):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    acc, stop = [], threading.Event(); th = threading.Thread(target=sample, args=(tag, stop, acc)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(5): fn()
        torch.cuda.synchronize(); n += 5
    ms = (time.time() - t0) / n * 1e3
    stop.set(); th.join()
    print(f"{tag:34s}: {ms:6.2f} ms per 40 crops; rocm-smi (sclk, power): {acc[1:-1][:8]}", flush=True)


if "--hbm" in sys.argv:
    # What does a byte cost?  Streaming copies (torch's copy kernel: probe only) over buffers that live in HBM (2 x 2 GiB), in the
    # 256 MiB memory-side cache (2 x 48 MiB) and a read-only reduction, random and zero contents: GB/s, W, and (W - idle) / (GB/s) = mJ per GB.
    def watts(fn, secs=3.0, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        acc, stop = [], threading.Event(); th = threading.Thread(target=sample, args=("", stop, acc)); th.start()
        t0 = time.time(); n = 0
        while time.time() - t0 < secs:
            for _ in range(reps): fn()
            torch.cuda.synchronize(); n += reps
        dt_ = (time.time() - t0) / n
        stop.set(); th.join()
        ws = [float(p_) for _, p_ in acc[1:-1] if p_.replace(".", "").isdigit()]
        return dt_, sum(ws) / max(len(ws), 1), (acc[2][0] if len(acc) > 2 else "?")
    acc0 = []; stop0 = threading.Event(); th0 = threading.Thread(target=sample, args=("", stop0, acc0)); th0.start(); time.sleep(2.5); stop0.set(); th0.join()
    idle = [float(p_) for _, p_ in acc0 if p_.replace(".", "").isdigit()]; idle = sum(idle) / max(len(idle), 1)
    print(f"idle: {idle:.0f} W", flush=True)
    for tag, nbytes in (("HBM (2 GiB -> 2 GiB)", 2 << 30), ("memory-side cache (48 MiB -> 48 MiB)", 48 << 20), ("L2-sized (2 MiB -> 2 MiB per call, 16 calls)", 2 << 20)):
        for kind in ("random", "zero"):
            src = (torch.randint(0, 2 ** 31 - 1, (nbytes // 4,), dtype=torch.int32, device=dev) if kind == "random" else torch.zeros(nbytes // 4, dtype=torch.int32, device=dev))
            dst = torch.empty_like(src)
            reps = 4 if nbytes > (1 << 30) else 200
            fn = (lambda: dst.copy_(src))
            t, w, sclk = watts(fn, reps=reps)
            gbs = 2 * nbytes / t / 1e9
            print(f"copy  {tag:46s} {kind:6s}: {gbs:7.0f} GB/s (read + write), {w:6.0f} W, sclk {sclk}: {(w - idle) / gbs * 1e3:6.1f} mJ per GB moved", flush=True)
            fn = (lambda: src.sum())
            t, w, sclk = watts(fn, reps=reps)
            gbs = nbytes / t / 1e9
            print(f"read  {tag:46s} {kind:6s}: {gbs:7.0f} GB/s,                {w:6.0f} W, sclk {sclk}: {(w - idle) / gbs * 1e3:6.1f} mJ per GB read", flush=True)
            del src, dst
    sys.exit(0)

if "--mfma" in sys.argv:
    # the bare MFMA stream of profiles/r04_ps_ablation.txt (gemm_ps32_kernel with every memory instruction removed: diagnostic
    # ablation 134; fc1 shape, M = 23080) and the complete persistent kernel, on random and on zero operands: is 1.62 PF a POWER limit?
    import ctypes as C
    from slime_amd import _lib
    lib = _lib.load_diag(); E = _lib
    M, N, K = 23080, 4096, 1024
    for tag, zero in (("random operands", False), ("zero operands", True)):
        g0 = torch.Generator().manual_seed(1)
        x = (torch.zeros(M, K) if zero else torch.randn(M, K, generator=g0)).to(dt).to(dev)
        w = (torch.zeros(N, K) if zero else torch.randn(N, K, generator=g0) * K ** -0.5).to(dt).to(dev)
        stats = torch.stack([x.float().view(M, K // 64, 64).sum(-1), (x.float() ** 2).view(M, K // 64, 64).sum(-1)], -1).contiguous()
        bias = torch.zeros(N, device=dev); colsum = w.float().sum(-1).contiguous(); wf = ops.pack_b_frag(w)
        out = torch.empty((M, N), dtype=dt, device=dev)
        g = E.GemmArgs(A=x.data_ptr(), lda=K, B=w.data_ptr(), bias=bias.data_ptr(), C=out.data_ptr(), ldc=N, M=M, N=N, K=K, dtype=ops.dtype_code(dt),
                       epilogue=E.EPI_BIAS_QUICKGELU_T, ln_stats=stats.data_ptr(), ln_groups=16, ln_colsum=colsum.data_ptr(), ln_eps=1e-5, B_frag=wf.data_ptr())
        st = ops._stream()
        ladder = ((134, "bare MFMA stream"), (70, "+ activation-fragment reads (LDS)"), (54, "+ weight requests (L2 -> VGPR)"), (38, "+ LDS-DMA instead (L2 -> LDS)"),
                  (22, "+ both = complete main loop"), (18, "+ counted waits"), (0, "complete persistent kernel"), (-1, "shipped direct-B kernel"))
        for abl, name in (ladder if "--ladder" in sys.argv else (ladder[0], ladder[-2], ladder[-1])):
            lib.slime_gemm_force_tile(12 if abl < 0 else 17); lib.slime_gemm_set_db_ablation(max(abl, 0))
            fn = lambda: lib.slime_gemm_ex(C.byref(g), st)
            for _ in range(5): fn()
            torch.cuda.synchronize()
            acc, stop = [], threading.Event(); th = threading.Thread(target=sample, args=("", stop, acc)); th.start()
            t0 = time.time(); n = 0
            while time.time() - t0 < 3.0:
                for _ in range(50): fn()
                torch.cuda.synchronize(); n += 50
            us = (time.time() - t0) / n * 1e6
            stop.set(); th.join()
            ws = [float(p_) for _, p_ in acc[1:-1] if p_.replace(".", "").isdigit()]
            w_avg = sum(ws) / max(len(ws), 1)
            print(f"{name:36s} {tag:16s}: {us:6.1f} us = {2.0*M*N*K/us/1e6:6.0f} TF/s; {w_avg:6.0f} W -> {w_avg*us*1e-3:6.1f} mJ per launch; sclk {acc[2][0] if len(acc) > 2 else '?'}", flush=True)
        lib.slime_gemm_force_tile(0); lib.slime_gemm_set_db_ablation(0)
    sys.exit(0)

acc0 = []; stop0 = threading.Event(); th0 = threading.Thread(target=sample, args=("", stop0, acc0)); th0.start(); time.sleep(2.0); stop0.set(); th0.join()
print("idle (context created, nothing running): rocm-smi (sclk, W):", acc0[:4], flush=True)
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
zsd = {k: torch.zeros_like(v) for k, v in tsd.items()}
for rnd in range(2):
    r2, r1 = tower_pair(tsd, px)
    measure("bench operands, two streams", r2); measure("bench operands, one stream", r1)
    z2, z1 = tower_pair(zsd, torch.zeros_like(px))
    measure("all-zero operands, two streams", z2); measure("all-zero operands, one stream", z1)
