#!/usr/bin/env python3
"""Round 4: the persistent direct-B GEMM (tile 16, gemm_ps.inc: the previous tile's epilogue rides in the next tile's MFMA stream)
against the direct-B kernel (tile 12) -- bit-equality on the tower's LayerNorm-fold shapes (incl. guard rows past a ragged M),
stand-alone timing, and the tower with per-shape overrides.  Diagnostic build, same process, interleaved rounds.

    python tools/ps_gemm_check.py [check] [time] [tower]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); lib = _lib.load_diag()
E = _lib
what = set(sys.argv[1:]) or {"check", "time", "tower"}
PS = 17                                   # 17 = persistent kernel on 32x32x16 MFMAs (gemm_ps32.inc), 16 = its 16x16x32 predecessor


def rnd(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


def time_ms(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def consumer_case(M, N, dt, seed=0):
    """x16 + producer statistics as the tower has them, folded weights; returns a closure (tile) -> output incl. guard rows."""
    K = 1024
    x = rnd((M, K), seed + 1, 1.0) + rnd((M, 1), seed + 2, 0.5)              # rows with a mean
    x16 = x.to(dt)
    xf = x.float()                                                           # statistics of the fp32 rows (as the producer has them)
    stats = torch.stack([xf.view(M, K // 64, 64).sum(-1), (xf * xf).view(M, K // 64, 64).sum(-1)], dim=-1).contiguous()
    w = rnd((N, K), seed + 3, K ** -0.5).to(dt)
    bias = rnd((N,), seed + 4, 0.5)
    colsum = w.float().sum(-1).contiguous()
    wf = ops.pack_b_frag(w)
    GUARD = 96

    def run(tile, epi, abl=0):
        lib.slime_gemm_force_tile(tile)
        lib.slime_gemm_set_db_ablation(abl)
        buf = torch.full((M + GUARD, N), 7.0, dtype=dt, device=dev)
        g = E.GemmArgs(A=x16.data_ptr(), lda=K, B=w.data_ptr(), bias=bias.data_ptr(), C=buf.data_ptr(), ldc=N, M=M, N=N, K=K,
                       dtype=ops.dtype_code(dt), epilogue=epi, ln_stats=stats.data_ptr(), ln_groups=K // 64,
                       ln_colsum=colsum.data_ptr(), ln_eps=1e-5, B_frag=wf.data_ptr())
        import ctypes as C
        E.check(lib.slime_gemm_ex(C.byref(g), ops._stream()), "slime_gemm_ex")
        torch.cuda.synchronize()
        lib.slime_gemm_force_tile(0); lib.slime_gemm_set_db_ablation(0)
        return buf
    return run, (x16, stats, w, bias, colsum, wf)


def tile_report(out, ref, M, N, cus=256, group_m=8):
    """Which 128 x 256 tiles differ, by the persistent kernel's walk: virtual block vb -> (workgroup, body index)."""
    tm, tn = (M + 127) // 128, N // 256
    nblk = tm * tn
    o, r = out[:M].float(), ref[:M].float()
    bad = torch.zeros((tm, tn), dtype=torch.bool)
    frac = torch.zeros((tm, tn))
    for i in range(tm):
        d = (o[i * 128:(i + 1) * 128] != r[i * 128:(i + 1) * 128]).view(-1, tn, 256).float().mean((0, 2)).cpu()
        frac[i] = d; bad[i] = d > 0
    q, rr = nblk >> 3, nblk & 7
    by_body, by_last, by_xcd = {}, {}, {}
    for vb in range(nblk):
        xcd = vb & 7
        pid = (xcd * (q + 1) if xcd < rr else rr * (q + 1) + (xcd - rr) * q) + (vb >> 3)
        in_group = group_m * tn
        first_m = (pid // in_group) * group_m
        gsz = min(tm - first_m, group_m)
        i, j = first_m + (pid % in_group) % gsz, (pid % in_group) // gsz
        body, wg = vb // cus, vb % cus
        n_wg = (nblk - wg + cus - 1) // cus                                  # tiles of this workgroup
        b = bool(bad[i, j])
        for dct, key in ((by_body, body), (by_last, n_wg - 1 - body), (by_xcd, xcd)):
            t = dct.setdefault(key, [0, 0]); t[0] += 1; t[1] += b
    print("   tiles bad/total by body index:", {k: f"{v[1]}/{v[0]}" for k, v in sorted(by_body.items())})
    print("   ... by bodies left after it  :", {k: f"{v[1]}/{v[0]}" for k, v in sorted(by_last.items())})
    print("   ... by XCD                   :", {k: f"{v[1]}/{v[0]}" for k, v in sorted(by_xcd.items())})
    print("   mean mismatch fraction inside bad tiles %.3f; row-in-tile histogram of bad rows (16 bins of 8): %s" % (
        float(frac[bad].mean()) if bad.any() else 0.0,
        torch.bincount(((o != r).any(1).nonzero().flatten() % 128) // 8, minlength=16).tolist()), flush=True)


if "check" in what:
    print("== bit-equality persistent (tile 16) vs direct-B (tile 12), LayerNorm-fold consumers ==", flush=True)
    bad = 0
    for dt in (torch.bfloat16, torch.float16):
        for M in (11540, 23080, 8192 + 20):
            for N, epi, nm in ((3072, E.EPI_BIAS_T, "qkv"), (4096, E.EPI_BIAS_QUICKGELU_T, "fc1")):
                run, _ = consumer_case(M, N, dt, seed=M % 97)
                ref = run(12, epi)
                for rep in range(2):
                    out = run(PS, epi)
                    eq = torch.equal(out, ref)
                    msg = f"{str(dt)[6:]:9s} M {M:6d} {nm}: equal={eq}"
                    if not eq:
                        bad += 1
                        d = (out.float() - ref.float())
                        nz = (d != 0) | (out.float().isnan() != ref.float().isnan())
                        rows = nz.any(1).nonzero().flatten(); cols = nz.any(0).nonzero().flatten()
                        msg += (f"  mismatched elements {int(nz.sum())} rows {int(rows.numel())} [{int(rows.min())}..{int(rows.max())}] cols {int(cols.numel())} "
                                f"[{int(cols.min())}..{int(cols.max())}] max|d| {float(d.abs().nan_to_num(1e9).max()):.3g} guard rows touched "
                                f"{bool((out[M:] != 7.0).any())}  row%128 hist {torch.bincount(rows % 128, minlength=128).nonzero().flatten()[:12].tolist()} "
                                f"col%64 hist {torch.bincount(cols % 64, minlength=64).nonzero().flatten()[:12].tolist()}")
                    print(msg, flush=True)
                    if not eq and rep == 0:
                        tile_report(out, ref, M, N)
    run, _ = consumer_case(11540, 4096, torch.bfloat16, seed=5)
    ref = run(12, E.EPI_BIAS_QUICKGELU_T)
    for abl, nm in ((17, "every counted wait drained"), (0, "shipped waits")):
        out = run(PS, E.EPI_BIAS_QUICKGELU_T, abl)
        print(f"fc1 M 11540 bf16, persistent with {nm}: equal={torch.equal(out, ref)}", flush=True)
        if not torch.equal(out, ref): tile_report(out, ref, 11540, 4096)
    print("CHECK", "FAILED" if bad else "ok", flush=True)

if "time" in what:
    print("== stand-alone, hot operands, TF/s (us): direct-B (tile 12) | persistent 16x16x32 (tile 16) | persistent 32x32x16 (tile 17) [fc1: tile 17 with | no epilogue slots | + no vmcnt waits | + no weight requests | + no LDS-DMA (weights kept) | + neither | + no fragment reads | tile 16 without epilogue slots]; three interleaved rounds ==", flush=True)
    dt = torch.bfloat16
    for M in (11540, 23080):
        for N, epi, nm in ((3072, E.EPI_BIAS_T, "qkv"), (4096, E.EPI_BIAS_QUICKGELU_T, "fc1")):
            _, (x16, stats, w, bias, colsum, wf) = consumer_case(M, N, dt)
            out = torch.empty((M, N), dtype=dt, device=dev)
            import ctypes as C
            g = E.GemmArgs(A=x16.data_ptr(), lda=1024, B=w.data_ptr(), bias=bias.data_ptr(), C=out.data_ptr(), ldc=N, M=M, N=N, K=1024,
                           dtype=ops.dtype_code(dt), epilogue=epi, ln_stats=stats.data_ptr(), ln_groups=16,
                           ln_colsum=colsum.data_ptr(), ln_eps=1e-5, B_frag=wf.data_ptr())
            st = ops._stream()
            fn = lambda: lib.slime_gemm_ex(C.byref(g), st)
            fl = 2.0 * M * N * 1024
            for rep in range(3):
                row = []
                for tile, abl in ((12, 0), (16, 0), (PS, 0)) + (((PS, 18), (PS, 22), (PS, 38), (PS, 54), (PS, 70), (PS, 134), (16, 18)) if nm == "fc1" else ()):
                    lib.slime_gemm_force_tile(tile); lib.slime_gemm_set_db_ablation(abl)
                    row.append(time_ms(fn))
                lib.slime_gemm_force_tile(0); lib.slime_gemm_set_db_ablation(0)
                print(f"M {M:6d} {nm}: " + " | ".join(f"{fl/t/1e9:6.0f} ({t*1e3:5.1f})" for t in row), flush=True)

if "tower" in what:
    dt = torch.bfloat16
    tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
    px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
    pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    parts = list(px.chunk(2))

    def run2():
        cur = torch.cuda.current_stream()
        for s in streams: s.wait_stream(cur)
        outs = []
        for pt, s, p in zip(pts, streams, parts):
            with torch.cuda.stream(s): outs.append(ops.tower_forward(pt, p))
        for s in streams: cur.wait_stream(s)
        return torch.cat(outs)

    def run1():
        return ops.tower_forward(pts[0], px)

    def rules(on):
        lib.slime_gemm_set_shape_tile(0, 0, 0)
        for (N, K) in on: lib.slime_gemm_set_shape_tile(N, K, PS)

    CONFIGS = [("shipped dispatch", ()), ("persistent qkv + fc1", ((3072, 1024), (4096, 1024))), ("persistent fc1", ((4096, 1024),)),
               ("persistent qkv", ((3072, 1024),))]
    rules(()); ref2 = run2(); ref1 = run1(); torch.cuda.synchronize()
    print("== tower, 40 crops: two streams x 20 | one stream x 40 (ms); outputs bit-equal to the shipped dispatch? ==", flush=True)
    for rep in range(3):
        for name, on in CONFIGS:
            rules(on)
            ts = []
            eqs = (torch.equal(run2(), ref2), torch.equal(run1(), ref1))
            for fn in (run2, run1):
                for _ in range(2): fn()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(8): fn()
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 8)
            print(f"{name:26s}: {ts[0]*1e3:6.2f} ms {40/ts[0]:5.0f} crops/s | {ts[1]*1e3:6.2f} ms {40/ts[1]:5.0f} crops/s   equal {eqs}", flush=True)
    rules(())
