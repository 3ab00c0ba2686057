"""GPU parity tests of every HIP kernel behind the C ABI (run on the MI355X box: pytest -m gpu).

Each kernel is compared with a plain fp32 torch restatement of the same op evaluated on the SAME
16-bit-rounded operands, so the tolerances below bound only accumulation order (fp32 outputs) or one
final 16-bit rounding (T outputs):
    fp32 outputs : rel-L2 <= 2e-5
    bf16 outputs : rel-L2 <= 4e-3   (bf16 unit roundoff 2^-9 = 1.95e-3 per element)
    fp16 outputs : rel-L2 <= 6e-4   (fp16 unit roundoff 2^-12 = 4.9e-4)
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL_F32 = 2e-5
TOL_T = {torch.bfloat16: 4e-3, torch.float16: 6e-4}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from slime_amd import _lib
    _lib.load()                      # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


def _rand(shape, dtype, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile,sched", [(1, 1), (1, 0), (3, 1), (3, 0), (0, 1), (4, 1), (5, 1), (7, 1), (9, 1), (10, 1), (11, 1), (15, 1), (18, 1)])
@pytest.mark.parametrize("M,N,K", [(1731, 1024, 1024), (300, 768, 640), (77, 256, 64), (2308, 512, 128)])
def test_gemm_epilogues(dev, dtype, tile, sched, M, N, K):
    """Every tile / kernel variant (forced through the DIAGNOSTIC build's hook) x every epilogue vs fp32 torch."""
    from slime_amd import _lib
    with _lib.diag() as lib:
        lib.slime_gemm_force_tile(tile)
        lib.slime_gemm_set_sched(sched)
        try:
            _check_all_epilogues(dev, dtype, M, N, K)
        finally:
            lib.slime_gemm_force_tile(0)
            lib.slime_gemm_set_sched(1)


def _check_all_epilogues(dev, dtype, M, N, K):
    from slime_amd import ops, _lib
    a = _rand((M, K), dtype, dev, 1)
    w = _rand((N, K), dtype, dev, 2, K ** -0.5)     # asymmetric, random: catches row/col swaps
    bias = _rand((N,), torch.float32, dev, 3)
    ref = a.float() @ w.float().t() + bias
    out = ops.gemm(a, w, bias, _lib.EPI_BIAS_F32)
    assert rel_l2(out, ref) < TOL_F32
    out = ops.gemm(a, w, None, _lib.EPI_BIAS_F32)
    assert rel_l2(out, ref - bias) < TOL_F32
    out = ops.gemm(a, w, bias, _lib.EPI_BIAS_T)
    assert out.dtype == dtype and rel_l2(out.float(), ref) < TOL_T[dtype]
    out = ops.gemm(a, w, bias, _lib.EPI_BIAS_QUICKGELU_T)
    assert rel_l2(out.float(), ref * torch.sigmoid(1.702 * ref)) < TOL_T[dtype]
    out = ops.gemm(a, w, bias, _lib.EPI_BIAS_GELU_T)
    assert rel_l2(out.float(), F.gelu(ref)) < TOL_T[dtype]
    h = _rand((M, N), torch.float32, dev, 4)
    h0 = h.clone()
    ops.gemm(a, w, bias, _lib.EPI_BIAS_RESID_F32, out=h)
    assert rel_l2(h, h0 + ref) < TOL_F32
    # 16-bit residual stream (Llama decoder layer): T(acc + bias + resid), also in place (out aliases resid)
    r = _rand((M, N), dtype, dev, 5, 2.0)
    out = ops.gemm(a, w, bias, _lib.EPI_BIAS_RESID_T, resid=r)
    assert out.dtype == dtype and rel_l2(out.float(), ref + r.float()) < TOL_T[dtype]
    r2 = r.clone()
    ops.gemm(a, w, bias, _lib.EPI_BIAS_RESID_T, out=r2, resid=r2)
    assert torch.equal(r2, out)


# The tower's production launch shapes (20-crop half batch M = 11540; the stacked adapter MLP M = 13824), AUTO dispatch of the
# PRODUCT library, every epilogue, against fp32 torch on the same rounded operands.  K >= 2048 reaches the KTAG = 1
# instantiations: gemm_pp_kernel<T, EPI, 1, 0, 4> (fc2 + residual: N = 1024, K = 4096, sub-round grid -> ping-pong) and
# gemm_w4_kernel<T, EPI, 1, MI> (N = 4096: multi-round grid -> four-wave stream kernel, hand-fenced epilogue).
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(11540, 1024, 4096), (11540, 4096, 1024), (11540, 3072, 1024), (11540, 1024, 1024),
                                   (13824, 4096, 4096), (13824, 4096, 1024), (11540, 1024, 2048), (23080, 4096, 2048)])
def test_gemm_production_shapes(dev, dtype, M, N, K):
    _check_all_epilogues(dev, dtype, M, N, K)


# ---- direct-B kernel (round 3): the static operand in MFMA-fragment order, 128 x 256 tiles, two workgroups per CU ----
def _frag_order_reference(w):
    """slime_gemm_pack_b's documented layout (include/slime_hip.h) as a torch permutation:
    out[t][s][f][lane][e] = w[64 t + 32 (f >> 1) + 8 ((lane & 15) >> 2) + 4 (f & 1) + (lane & 3)][32 s + 8 (lane >> 4) + e]."""
    N, K = w.shape
    v = w.view(N // 64, 2, 4, 2, 4, K // 32, 4, 8)          # t, p, g, h, q, s, lq, e   (n = 64t + 32p + 8g + 4h + q)
    return v.permute(0, 5, 1, 3, 6, 2, 4, 7).contiguous().view(N, K)    # t, s, p, h (f = 2p + h), lq, g, q (lane = 16 lq + 4g + q), e


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,K", [(256, 64), (1024, 1024), (3072, 1024), (1024, 4096)])
def test_gemm_pack_b_layout(dev, dtype, N, K):
    """The fragment-order copy is a pure permutation of 16-byte chunks: bit-exact against the documented index map."""
    from slime_amd import ops
    w = _rand((N, K), dtype, dev, 11)
    wf = ops.pack_b_frag(w)
    assert wf is not None and wf.shape == w.shape and wf.dtype == w.dtype
    assert torch.equal(wf, _frag_order_reference(w))
    stacked = torch.stack([w, w.flip(0)])                  # per-layer stacks are packed slice by slice
    wf2 = ops.pack_b_frag(stacked)
    assert torch.equal(wf2[0], wf) and torch.equal(wf2[1], _frag_order_reference(w.flip(0).contiguous()))
    # ABI 5: every [N % 64, K % 64] operand can be run from its fragment image (the library is asked, not a rule restated here)
    lib = ops._lib.load()
    assert lib.slime_gemm_b_frag_usable(384, 128) == 1 and ops.pack_b_frag(_rand((384, 128), dtype, dev, 12)) is not None
    assert lib.slime_gemm_b_frag_usable(96, 128) == 0 and ops.pack_b_frag(_rand((96, 128), dtype, dev, 12)) is None
    assert lib.slime_gemm_b_frag_usable(128, 96) == 0


def _check_direct_b(dev, dtype, M, N, K, forced):
    """gemm_db_kernel against (a) fp32 torch on the same rounded operands and (b) the LDS-staged kernels BIT FOR BIT (same
    k order per accumulator, same epilogue code), every epilogue incl. both sides of the LayerNorm fold."""
    from slime_amd import ops, _lib
    a = _rand((M, K), dtype, dev, 1)
    w = _rand((N, K), dtype, dev, 2, K ** -0.5)
    bias = _rand((N,), torch.float32, dev, 3)
    wf = ops.pack_b_frag(w)
    assert wf is not None
    if not forced:
        # the dispatch rule (gemm.hip auto_tile): fragment-order B -> direct-B kernel, except K > 2048 grids between half a round and
        # one round of 256-row tiles (fc2 at a 20-crop half batch: ping-pong kernel)
        n256 = ((M + 255) // 256) * (N // 256)
        pp_band = K > 2048 and 128 <= n256 < 256
        assert ("gemm_db_kernel" in ops.gemm_kernel_name(M, N, K, dtype, _lib.EPI_BIAS_T, True)) == (not pp_band)
        assert "gemm_db_kernel" not in ops.gemm_kernel_name(M, N, K, dtype, _lib.EPI_BIAS_T, False)
    ref = a.float() @ w.float().t() + bias
    for epi, fn, tol in ((_lib.EPI_BIAS_F32, lambda r: r, TOL_F32), (_lib.EPI_BIAS_T, lambda r: r, TOL_T[dtype]),
                         (_lib.EPI_BIAS_QUICKGELU_T, lambda r: r * torch.sigmoid(1.702 * r), TOL_T[dtype]),
                         (_lib.EPI_BIAS_GELU_T, F.gelu, TOL_T[dtype])):
        got = ops.gemm(a, w, bias, epi, w_frag=wf)
        assert rel_l2(got.float(), fn(ref)) < tol, epi
        if not forced:
            assert torch.equal(got, ops.gemm(a, w, bias, epi)), f"epilogue {epi}: differs from the LDS-staged kernel"
    got = ops.gemm(a, w, None, _lib.EPI_BIAS_F32, w_frag=wf)
    assert rel_l2(got, ref - bias) < TOL_F32
    r = _rand((M, N), dtype, dev, 5, 2.0)
    got = ops.gemm(a, w, bias, _lib.EPI_BIAS_RESID_T, w_frag=wf, resid=r)
    assert rel_l2(got.float(), ref + r.float()) < TOL_T[dtype]
    if not forced:
        assert torch.equal(got, ops.gemm(a, w, bias, _lib.EPI_BIAS_RESID_T, resid=r))
    r2 = r.clone()
    ops.gemm(a, w, bias, _lib.EPI_BIAS_RESID_T, out=r2, w_frag=wf, resid=r2)
    assert torch.equal(r2, got)
    # residual update, plain and as LayerNorm-fold producer
    h0 = _rand((M, N), torch.float32, dev, 4, 2.0) + 0.3
    h1, h2, h3 = h0.clone(), h0.clone(), h0.clone()
    ops.gemm(a, w, bias, _lib.EPI_BIAS_RESID_F32, out=h1, w_frag=wf)
    assert rel_l2(h1, h0 + ref) < TOL_F32
    x16, stats = ops.gemm_ln_producer(a, w, bias, h2, w_frag=wf)
    assert torch.equal(h2, h1) and torch.equal(x16, h2.to(dtype))
    xr = h2.view(M, N // 64, 64)
    assert rel_l2(stats[..., 0], xr.sum(-1)) < 1e-5 and rel_l2(stats[..., 1], (xr * xr).sum(-1)) < 1e-5
    if not forced:
        x16b, statsb = ops.gemm_ln_producer(a, w, bias, h3)
        assert torch.equal(h3, h2) and torch.equal(x16b, x16) and torch.equal(statsb, stats)
    # LayerNorm-fold consumer (needs K = the normalised width, 64-column groups)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, K, generator=g) * 1.5 + 0.8
    x[:, 7] += 40.0
    x16 = x.to(dev).to(dtype)
    xr = x16.float().view(M, K // 64, 64)
    st = torch.stack([xr.sum(-1), (xr * xr).sum(-1)], -1).contiguous()
    colsum = w.double().sum(1).float()
    for code in (_lib.EPI_BIAS_T, _lib.EPI_BIAS_QUICKGELU_T):
        got = ops.gemm_ln_consumer(x16, st, w, bias, colsum, 1e-5, code, w_frag=wf)
        r = F.layer_norm(x16.double(), (K,), None, None, 1e-5) @ w.double().t() + bias.double()
        if code == _lib.EPI_BIAS_QUICKGELU_T:
            r = r * torch.sigmoid(1.702 * r)
        assert rel_l2(got.float(), r) < TOL_T[dtype]
        if not forced:
            assert torch.equal(got, ops.gemm_ln_consumer(x16, st, w, bias, colsum, 1e-5, code))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(11540, 3072, 1024), (11540, 4096, 1024), (11540, 1024, 4096), (11540, 1024, 1024), (13824, 4096, 4096),
                                   (5193, 3072, 1024), (23080, 1024, 1024), (5770, 1024, 4096)])
def test_gemm_direct_b_production_shapes(dev, dtype, M, N, K):
    """The tower's launch shapes through the PRODUCT library's auto dispatch with B_frag set (20-crop half batch, the stacked
    adapter MLP, a 9-crop rank shard, a 40-crop batch): gemm_db_kernel, ragged last row tile (11540 = 90 x 128 + 20)."""
    _check_direct_b(dev, dtype, M, N, K, forced=False)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(1731, 1024, 1024), (300, 768, 640), (77, 256, 64), (2308, 512, 128), (128, 256, 192), (129, 256, 128),
                                   (11540, 1024, 4096), (2885, 1024, 4096)])
@pytest.mark.parametrize("tile", [12, 13, 19])
def test_gemm_direct_b_small_and_edge_shapes(dev, dtype, M, N, K, tile):
    """The same kernel forced (diagnostic build) onto shapes the dispatch would not give it: one k-tile (K = 64: prologue +
    last-tile body only), two and three k-tiles (every tile-body variant), M < 128, M = 128 exactly, one row over, and the
    sub-round K = 4096 grids (fc2 at a 20-crop and a 5-crop batch) that the auto rule leaves to the LDS-staged kernels.
    tile 12 = 128-row, 13 = 64-row, 19 = 96-row (round 6) workgroup tiles; M = 77 / 128 / 129 / 300 also walk the 96-row tile's ragged
    last row tile (the forced tile takes the 96-row form for the tower's epilogues, the 128-row one for the others)."""
    from slime_amd import _lib
    with _lib.diag() as lib:
        lib.slime_gemm_force_tile(tile)
        try:
            _check_direct_b(dev, dtype, M, N, K, forced=True)
        finally:
            lib.slime_gemm_force_tile(0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", [16, 17])
@pytest.mark.parametrize("M,N,epi_name", [(11540, 4096, "quick_gelu"), (11540, 3072, "bias"), (8212, 4096, "quick_gelu"), (23080, 3072, "bias")])
def test_gemm_persistent_equals_direct_b(dev, dtype, tile, M, N, epi_name):
    """Round 4's measured alternatives (diagnostic build; profiles/r04_ps_ablation.txt): the persistent direct-B kernels whose
    epilogue rides in the next tile's MFMA stream -- tile 16 on 16x16x32 MFMAs (bf16 only: fp16 falls through to tile 17), tile 17
    on 32x32x16 -- reproduce gemm_db_kernel BIT FOR BIT on the LayerNorm-fold consumers (q/k/v, fc1) at the tower's launch shapes:
    ragged last row tile, 3-12 tiles per workgroup, every workgroup's last tile through the stand-alone drain; rows past M of
    the output buffer stay untouched (buffer stores bounded at M rows).  Also pins tools/mfma_shape_equal.hip's finding that the
    two MFMA shapes accumulate to the same fp32 bits."""
    import ctypes as C
    from slime_amd import ops, _lib
    K = 1024
    epi = _lib.EPI_BIAS_QUICKGELU_T if epi_name == "quick_gelu" else _lib.EPI_BIAS_T
    x = _rand((M, K), torch.float32, dev, 1) + _rand((M, 1), torch.float32, dev, 2, 0.5)
    x16 = x.to(dtype)
    stats = torch.stack([x.view(M, K // 64, 64).sum(-1), (x * x).view(M, K // 64, 64).sum(-1)], dim=-1).contiguous()
    w = _rand((N, K), dtype, dev, 3, K ** -0.5)
    bias = _rand((N,), torch.float32, dev, 4, 0.5)
    colsum = w.float().sum(-1).contiguous()
    wf = ops.pack_b_frag(w)
    with _lib.diag() as lib:
        outs = {}
        try:
            for t in (12, tile):
                buf = torch.full((M + 64, N), 7.0, dtype=dtype, device=dev)
                g = _lib.GemmArgs(A=x16.data_ptr(), lda=K, B=w.data_ptr(), bias=bias.data_ptr(), C=buf.data_ptr(), ldc=N, M=M, N=N, K=K,
                                  dtype=ops.dtype_code(dtype), epilogue=epi, ln_stats=stats.data_ptr(), ln_groups=K // 64,
                                  ln_colsum=colsum.data_ptr(), ln_eps=1e-5, B_frag=wf.data_ptr())
                lib.slime_gemm_force_tile(t)
                _lib.check(lib.slime_gemm_ex(C.byref(g), ops._stream()), "slime_gemm_ex")
                torch.cuda.synchronize()
                outs[t] = buf
        finally:
            lib.slime_gemm_force_tile(0)
    assert bool((outs[tile][M:] == 7.0).all()), "rows past M were written"
    assert torch.equal(outs[tile], outs[12])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(577, 512, 128), (64, 256, 1024), (1153, 4096, 1024), (4608, 4096, 1024)])
def test_gemm_gelu_mix_epilogue(dev, dtype, M, N, K):
    """SLIME_EPI_BIAS_GELU_MIX_T (round 4): one direct-B launch computes erf-GELU(A W^T + b) and erf-GELU(A2 W^T + b) for a token and
    stores T(g0 a0 + g1 a1) -- the GatedBlock's two experts' hidden rows mixed in fp32, rounded once.  Against float64 arithmetic on
    the same rounded operands; ragged token counts (577, 1153), one tile (64), the adapter's shape (4608 x 4096 x 1024); guard rows
    past M untouched; and the result equals the two-launch form (BIAS_GELU_T twice + slime_gate_premix) up to ITS extra rounding."""
    import ctypes as C
    from slime_amd import ops, _lib
    lib = _lib.load()
    code = ops.dtype_code(dtype)
    a = _rand((M, K), dtype, dev, 41)
    a2 = _rand((M, K), dtype, dev, 42)
    w = _rand((N, K), dtype, dev, 43, K ** -0.5)
    bias = _rand((N,), torch.float32, dev, 44, 0.5)
    x = _rand((M, 256), torch.float32, dev, 45)
    wg = _rand((256, 2), torch.float32, dev, 46, 0.2)
    gates = torch.empty((M, 2), dtype=torch.float32, device=dev)
    _lib.check(lib.slime_gate_weights(x.data_ptr(), 256, wg.data_ptr(), gates.data_ptr(), M, ops._stream()), "slime_gate_weights")
    p = torch.softmax(x.double().cpu() @ wg.double().cpu(), 1)
    gref = p / (p.sum(1, keepdim=True) + 1e-6)
    assert float((gates.double().cpu() - gref).abs().max()) < 2e-5
    wf = ops.pack_b_frag(w)
    buf = torch.full((M + 70, N), 7.0, dtype=dtype, device=dev)
    g = _lib.GemmArgs(A=a.data_ptr(), lda=K, B=w.data_ptr(), bias=bias.data_ptr(), C=buf.data_ptr(), ldc=N, M=M, N=N, K=K, dtype=code,
                      epilogue=_lib.EPI_BIAS_GELU_MIX_T, B_frag=wf.data_ptr(), A2=a2.data_ptr(), mix_gates=gates.data_ptr())
    _lib.check(lib.slime_gemm_ex(C.byref(g), ops._stream()), "slime_gemm_ex")
    torch.cuda.synchronize()
    assert bool((buf[M:] == 7.0).all()), "rows past M were written"
    wd, bd, gd = w.double().cpu(), bias.double().cpu(), gates.double().cpu()
    h0 = torch.nn.functional.gelu(a.double().cpu() @ wd.T + bd)
    h1 = torch.nn.functional.gelu(a2.double().cpu() @ wd.T + bd)
    want = gd[:, :1] * h0 + gd[:, 1:] * h1
    got = buf[:M].double().cpu()
    half = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    mag = (gd[:, :1] * h0).abs() + (gd[:, 1:] * h1).abs()
    excess = (got - want).abs() - (1.02 * half * want.abs() + 2e-5 * (mag + 1.0) + 1e-7)      # + fp32 accumulation of K products, erff
    i = int(excess.argmax())
    assert float(excess.max()) <= 0, (float(excess.max()), float(got.flatten()[i]), float(want.flatten()[i]))
    # the two-launch form: hidden rows rounded per expert, then mixed and rounded again
    mid = torch.empty((2 * M, N), dtype=dtype, device=dev)
    for src, dst in ((a, mid[:M]), (a2, mid[M:])):
        g2 = _lib.GemmArgs(A=src.data_ptr(), lda=K, B=w.data_ptr(), bias=bias.data_ptr(), C=dst.data_ptr(), ldc=N, M=M, N=N, K=K, dtype=code,
                           epilogue=_lib.EPI_BIAS_GELU_T, B_frag=wf.data_ptr())
        _lib.check(lib.slime_gemm_ex(C.byref(g2), ops._stream()), "slime_gemm_ex")
    _lib.check(lib.slime_gate_premix(x.data_ptr(), 256, wg.data_ptr(), mid[:M].data_ptr(), mid[M:].data_ptr(), mid[M:].data_ptr(), code, M, N,
                                     ops._stream()), "slime_gate_premix")
    assert rel_l2(mid[M:].float().cpu(), got.float()) < (4e-3 if dtype == torch.bfloat16 else 5e-4)
    assert rel_l2(got.float(), want.float()) < (2.5e-3 if dtype == torch.bfloat16 else 3.2e-4)
    # without the fragment-order weights there is no kernel for this epilogue: loud
    g.B_frag = None
    assert lib.slime_gemm_ex(C.byref(g), ops._stream()) != 0


def test_gemm_direct_b_determinism_under_load(dev):
    """Counted waits: a load that is waited for too early shows up as run-to-run differences, not as a large error.  The same
    GEMM 20 times while a second stream keeps the memory system busy: every result identical."""
    from slime_amd import ops, _lib
    dt = torch.bfloat16
    a = _rand((11540, 1024), dt, dev, 1)
    w = _rand((4096, 1024), dt, dev, 2, 1024 ** -0.5)
    bias = _rand((4096,), torch.float32, dev, 3)
    wf = ops.pack_b_frag(w)
    first = ops.gemm(a, w, bias, _lib.EPI_BIAS_QUICKGELU_T, w_frag=wf)
    noise = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream()
    for i in range(20):
        with torch.cuda.stream(side):
            noise.add_(1)
        got = ops.gemm(a, w, bias, _lib.EPI_BIAS_QUICKGELU_T, w_frag=wf)
        assert torch.equal(got, first), i
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(1731, 1024, 1024), (300, 256, 128), (11540, 1024, 4096), (11540, 1024, 1024)])
def test_gemm_layernorm_fold_producer(dev, dtype, M, N, K):
    """SLIME_EPI_BIAS_RESID_F32_LN: the residual update of BIAS_RESID_F32 bit for bit, plus x16 = T(h) exactly and the
    (sum, sum of squares) of the rounded rows per 64-column group (every kernel family via the auto dispatch: 128x128
    lock-step, ping-pong, and -- at M = 11540, K = 4096 / 1024 -- the production launch shapes of fc2 / out_proj)."""
    from slime_amd import ops, _lib
    a = _rand((M, K), dtype, dev, 1)
    w = _rand((N, K), dtype, dev, 2, K ** -0.5)
    bias = _rand((N,), torch.float32, dev, 3)
    h = _rand((M, N), torch.float32, dev, 4, 2.0) + 0.3
    h_plain = h.clone()
    ops.gemm(a, w, bias, _lib.EPI_BIAS_RESID_F32, out=h_plain)
    x16, stats = ops.gemm_ln_producer(a, w, bias, h)
    assert torch.equal(h, h_plain)
    assert torch.equal(x16, h.to(dtype))
    xr = h.view(M, N // 64, 64)
    assert rel_l2(stats[..., 0], xr.sum(-1)) < 1e-5 and rel_l2(stats[..., 1], (xr * xr).sum(-1)) < 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("epi", ["bias", "quick_gelu"])
@pytest.mark.parametrize("M,N,K", [(1731, 1024, 1024), (300, 256, 128), (577, 768, 256), (11540, 3072, 1024), (11540, 4096, 1024)])
def test_gemm_layernorm_fold_consumer(dev, dtype, epi, M, N, K):
    """LayerNorm folded into the consuming GEMM vs the unfused fp32 computation on the same rounded operands:
    epi(LayerNorm(x16; gamma, beta) @ W^T + b) with W' = T(W diag(gamma)), b' = b + W beta, colsum = row sums of W'.
    Rows carry a mean offset and a few large channels (the case the fold's  acc - mu * colsum  cancellation must survive)."""
    from slime_amd import ops, _lib
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, K, generator=g) * 1.5 + 0.8
    x[:, 7] += 40.0
    x[:, K // 2] -= 25.0
    h = x.to(dev)
    x16 = h.to(dtype)
    xr = x16.float().view(M, K // 64, 64)
    stats = torch.stack([xr.sum(-1), (xr * xr).sum(-1)], -1).contiguous()
    gamma = (torch.randn(K, generator=g) * 0.2 + 1).to(dev)
    beta = (torch.randn(K, generator=g) * 0.2).to(dev)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    Wf = (W.double() * gamma.double()[None]).float().to(dtype)
    bf = (b.double() + W.double() @ beta.double()).float()
    colsum = Wf.double().sum(1).float()
    code = _lib.EPI_BIAS_T if epi == "bias" else _lib.EPI_BIAS_QUICKGELU_T
    out = ops.gemm_ln_consumer(x16, stats, Wf, bf, colsum, 1e-5, code)
    xn = F.layer_norm(x16.double(), (K,), None, None, 1e-5)
    ref = xn @ Wf.double().t() + bf.double()
    if epi == "quick_gelu":
        ref = ref * torch.sigmoid(1.702 * ref)
    assert out.dtype == dtype and rel_l2(out.float(), ref) < TOL_T[dtype]
    # ... and it equals (to rounding of the A operand) the unfused kernel sequence it replaces
    _, xn16, _ = ops.layernorm(x16.float(), gamma, beta, 1e-5, dtype)
    unf = ops.gemm(xn16, W.to(dtype), b, code)
    assert rel_l2(out.float(), unf.float()) < 3 * TOL_T[dtype]


def test_gemm_strided_and_identity(dev):
    """lda/ldc > width (the q/k/v thirds of a packed buffer) and an A = I transpose check."""
    from slime_amd import ops, _lib
    dt = torch.bfloat16
    big = _rand((500, 3 * 256), dt, dev, 5)
    a = big[:, 256:512]                                   # lda = 768
    w = _rand((384, 256), dt, dev, 6, 0.1)
    outbuf = torch.zeros((500, 1024), dtype=torch.float32, device=dev)
    out = outbuf[:, 128:512]                              # ldc = 1024
    lib = _lib.load()
    _lib.check(lib.slime_gemm(a.data_ptr(), a.stride(0), w.data_ptr(), None, out.data_ptr(), out.stride(0), 500, 384,
                              256, _lib.BF16, _lib.EPI_BIAS_F32, torch.cuda.current_stream().cuda_stream))
    assert rel_l2(out.cpu(), (a.float() @ w.float().t()).cpu()) < TOL_F32
    assert float(outbuf[:, :128].abs().max()) == 0 and float(outbuf[:, 512:].abs().max()) == 0
    eye = torch.eye(256, dtype=dt, device=dev)
    out = ops.gemm(eye, w, None, _lib.EPI_BIAS_F32)       # I @ w.T == w.T exactly
    assert torch.equal(out.cpu(), w.float().t().cpu())


def test_gemm_rejects_bad_shapes(dev):
    from slime_amd import ops, _lib
    a = _rand((64, 96), torch.bfloat16, dev, 1)
    w = _rand((128, 96), torch.bfloat16, dev, 2)
    with pytest.raises(_lib.SlimeHipError, match="multiple of 64"):
        ops.gemm(a, w, None, _lib.EPI_BIAS_F32)
    a = _rand((64, 64), torch.bfloat16, dev, 1)
    w = _rand((100, 64), torch.bfloat16, dev, 2)
    with pytest.raises(_lib.SlimeHipError, match="multiple of 128"):
        ops.gemm(a, w, None, _lib.EPI_BIAS_F32)


# ------------------------------------------------------------------------------------ row kernels
@pytest.mark.parametrize("D", [128, 256, 1024])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_layernorm(dev, D, dtype):
    from slime_amd import ops
    rows = 577 * 2 + 3
    x = _rand((rows, D), torch.float32, dev, 7, 3.0) + 0.7
    w = _rand((D,), torch.float32, dev, 8) * 0.1 + 1
    b = _rand((D,), torch.float32, dev, 9) * 0.1
    add = _rand((577, D), torch.float32, dev, 10)
    ref = F.layer_norm(x, (D,), w, b, 1e-5)
    o32, ot, ot2 = ops.layernorm(x, w, b, 1e-5, dtype, want_f32=True, want_t=True, add=add)
    assert rel_l2(o32.cpu(), ref.cpu()) < TOL_F32
    assert rel_l2(ot.float().cpu(), ref.cpu()) < TOL_T[dtype]
    idx = torch.arange(rows, device=dev) % 577
    assert rel_l2(ot2.float().cpu(), (ref + add[idx]).cpu()) < TOL_T[dtype]
    # cast-only mode
    _, ot, _ = ops.layernorm(x, None, None, 0.0, dtype, normalize=False)
    assert torch.equal(ot.cpu(), x.to(dtype).cpu())


def _patch_embed_reference(sd, cfg, px, dt):
    """The oracle's embeddings (oracle.clip_embeddings = HF CLIPVisionEmbeddings) + pre_layrnorm on the SAME rounded operands the
    kernel multiplies: pixels and conv weight rounded to T, everything else fp32."""
    import oracle.slime_oracle as O
    from slime_amd import weights as W
    tsd = {k: v.clone() for k, v in W.strip_tower_prefix(sd).items()}
    tsd["embeddings.patch_embedding.weight"] = tsd["embeddings.patch_embedding.weight"].to(dt).float()
    x = O.clip_embeddings(tsd, px.float().cpu().to(dt).float(), cfg.patch_size)
    return F.layer_norm(x, (cfg.hidden_size,), tsd["pre_layrnorm.weight"], tsd["pre_layrnorm.bias"], cfg.layer_norm_eps)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("geom", ["vit_l_20_crops", "tiny_3_crops"])
def test_patch_embed_prenorm_vs_oracle(dev, dt, geom):
    """slime_patch_embed_prenorm (round 5: im2col + conv GEMM + class token + position table + pre-LayerNorm in ONE launch) against
    the oracle's CLIPVisionEmbeddings + pre_layrnorm -- at the bench step's launch shape (20 crops of ViT-L/14-336: 240 workgroups of
    48 rows x 1024 columns) and at the tiny test geometry (hidden 128: two waves per workgroup), fp32 and T pixels, with every output:
    fp32 rows, T(h), the split stream's lower half, partial sums of the ROUNDED rows."""
    from slime_amd import ops, weights as W, _lib
    cfg = W.CLIP_L_336 if geom.startswith("vit_l") else W.TINY
    n = 20 if geom.startswith("vit_l") else 3
    sd = W.make_tower_state_dict(cfg, seed=77)
    pt = ops.pack_tower(sd, cfg, dt, dev, select_layer=0)              # layers_run = 0: embeddings only
    T = pt.tensors
    px = W.synthetic_pixels(n, seed=5).to(dev)
    D, P, S = cfg.hidden_size, cfg.num_patches, cfg.seq_len
    kpad = pt.desc.kpad
    ref = _patch_embed_reference(sd, cfg, px, dt).reshape(n * S, D)
    h, x16, lo, stats = ops.patch_embed_prenorm(px, T["patch_w_frag"], T["cls"], T["pos"], T["pre_ln_w"], T["pre_ln_b"], cfg.layer_norm_eps,
                                                dt, cfg.image_size, cfg.patch_size, kpad, want_lo=True)
    assert rel_l2(h.cpu(), ref) < TOL_F32
    per_row = ((h.cpu() - ref).norm(dim=1) / ref.norm(dim=1)).max()
    assert float(per_row) < 5 * TOL_F32, float(per_row)               # no stray row (class-token rows, workgroup seams, last patch row)
    assert torch.equal(x16, h.to(dt))                                  # the GEMM operand is T(h) exactly
    hi_ref, lo_ref = ops.resid_split(h, dt)
    assert lo.dtype == torch.int8 and torch.equal(lo, lo_ref)          # the split stream's lower part is the byte of the host restatement, exactly
    bits = 16 if dt == torch.bfloat16 else 19
    assert float(((ops.resid_join(x16, lo) - h).abs() - h.abs() * 2.0 ** -bits).max()) <= (0.0 if dt == torch.bfloat16 else 2.0 ** -25)
    xr = x16.float().view(n * S, D // 64, 64)
    assert rel_l2(stats[..., 0], xr.sum(-1)) < 1e-5 and rel_l2(stats[..., 1], (xr * xr).sum(-1)) < 1e-5
    # pixels already in T: the same bits (fp32 pixels are rounded to T on the way in, clip_encoder.py:55)
    h2, x2, _, st2 = ops.patch_embed_prenorm(px.to(dt), T["patch_w_frag"], T["cls"], T["pos"], T["pre_ln_w"], T["pre_ln_b"],
                                             cfg.layer_norm_eps, dt, cfg.image_size, cfg.patch_size, kpad)
    assert torch.equal(h2, h) and torch.equal(x2, x16) and torch.equal(st2, stats)
    # batch invariance: a crop's rows do not depend on its position in the batch or on the batch size
    h3, _, _, _ = ops.patch_embed_prenorm(px[n - 1:], T["patch_w_frag"], T["cls"], T["pos"], T["pre_ln_w"], T["pre_ln_b"], cfg.layer_norm_eps,
                                          dt, cfg.image_size, cfg.patch_size, kpad)
    assert torch.equal(h3, h[(n - 1) * S:])
    # the tower driver's entry 0 (hidden_states[0]) is this kernel's output
    st = ops.tower_hidden_states(pt, px)
    assert st.shape[0] == 1 and rel_l2(st[0].reshape(n * S, D).cpu(), ref) < (2.0 ** -15 if dt == torch.bfloat16 else 2e-5) + TOL_F32
    lib = _lib.load()
    with pytest.raises(_lib.SlimeHipError, match="lo8"):               # the lower part comes with its upper part
        _lib.check(lib.slime_patch_embed_prenorm(px.data_ptr(), _lib.F32, T["patch_w_frag"].data_ptr(), T["cls"].data_ptr(), T["pos"].data_ptr(),
                                                 T["pre_ln_w"].data_ptr(), T["pre_ln_b"].data_ptr(), 1e-5, h.data_ptr(), None, lo.data_ptr(), None,
                                                 ops.dtype_code(dt), n, cfg.image_size, cfg.patch_size, kpad, D, 0))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,tile", [(11540, 1024, 4096, 0), (11540, 1024, 1024, 0), (2885, 1024, 4096, 0), (577, 1024, 1024, 0),
                                        (1731, 1024, 1024, 4), (300, 768, 640, 3), (2308, 512, 128, 15), (1731, 1024, 1024, 12),
                                        (1731, 1024, 1024, 11), (1731, 1024, 1024, 1), (1731, 1024, 1024, 18), (577, 1024, 4096, 18),
                                        (1731, 1024, 1024, 19), (5193, 1024, 4096, 19), (11540, 1024, 1024, 19), (100, 1024, 1024, 19)])
def test_gemm_split_residual_epilogue(dev, dtype, M, N, K, tile):
    """SLIME_EPI_BIAS_RESID_SPLIT_LN (round 5; ABI 7: the lower part is one signed byte): the residual update on the split stream.
    Against fp32 torch on the same operands: hi' = T(c) EXACTLY for c = the kernel's own fp32 result (checked through the fp32
    epilogue, which computes the same c when fed h = join(hi, lo8)), lo8' = the host restatement's byte exactly, join(hi', lo8')
    within 2^-16 (bf16) / 2^-19 (fp16) of c, partial sums of c; and bit-equal across kernel families (auto dispatch with / without
    the fragment image, forced tiles)."""
    from slime_amd import ops, _lib
    a = _rand((M, K), dtype, dev, 1)
    w = _rand((N, K), dtype, dev, 2, K ** -0.5)
    bias = _rand((N,), torch.float32, dev, 3)
    wf = ops.pack_b_frag(w)
    h0 = _rand((M, N), torch.float32, dev, 4, 2.0) + 0.3
    h0[:, 5] *= 50.0                                                   # an outlier channel
    hi0, lo0 = ops.resid_split(h0, dtype)
    hsum = ops.resid_join(hi0, lo0)                                    # what the stream holds

    def run(with_frag, with_w=True):
        hi, lo = hi0.clone(), lo0.clone()
        st = ops.gemm_resid_split(a, w if with_w else None, bias, hi, lo, w_frag=wf if with_frag else None)
        return hi, lo, st
    with _lib.diag() as lib:
        lib.slime_gemm_force_tile(tile)
        try:
            hi, lo, st = run(True)
            c = hsum.clone()
            ops.gemm(a, w, bias, _lib.EPI_BIAS_RESID_F32, out=c, w_frag=wf)          # the same fp32 c, through the fp32 epilogue
        finally:
            lib.slime_gemm_force_tile(0)
    ref = hsum.double() + a.double() @ w.double().t() + bias.double()
    assert rel_l2(c, ref) < TOL_F32
    assert torch.equal(hi, c.to(dtype)), "upper half must be T(c)"
    assert lo.dtype == torch.int8 and torch.equal(lo, ops.resid_split(c, dtype)[1]), "lower part must be the byte of the host restatement"
    # what the pair keeps of c: 16 (bf16) / 19 (fp16) significant bits; below fp16's normal range (|c| < 2^-14) what hi keeps: 2^-25
    bits = 16 if dtype == torch.bfloat16 else 19
    err = (ops.resid_join(hi, lo) - c).abs() - c.abs() * 2.0 ** -bits
    assert float(err.max()) <= (0.0 if dtype == torch.bfloat16 else 2.0 ** -25), float(err.max())
    cr = c.view(M, N // 64, 64)
    assert rel_l2(st[..., 0], cr.sum(-1)) < 1e-5 and rel_l2(st[..., 1], (cr * cr).sum(-1)) < 1e-5
    if tile == 0:
        # product dispatch: with the fragment image (direct-B / ping-pong), without it (LDS-staged), and from the image ALONE
        for args in ((False, True), (True, False)):
            hi2, lo2, st2 = run(*args)
            assert torch.equal(hi2, hi) and torch.equal(lo2, lo) and torch.equal(st2, st), args


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,tile", [(11540, 1024, 4096, 0), (11540, 4096, 1024, 0), (2885, 1024, 4096, 0), (577, 3072, 1024, 0), (577, 1024, 4096, 0),
                                        (4608, 1024, 1024, 0), (300, 768, 640, 0), (77, 256, 64, 0), (2308, 512, 128, 0), (1731, 1024, 1024, 4),
                                        (1731, 1024, 1024, 9), (1731, 1024, 1024, 3), (1731, 1024, 1024, 15), (1731, 1024, 1024, 1),
                                        (1731, 1024, 1024, 11), (1731, 1024, 1024, 5), (1731, 1024, 1024, 18),
                                        (300, 768, 640, 18)])
def test_gemm_from_fragment_image_alone(dev, dtype, M, N, K, tile):
    """ABI 5: with B = NULL every kernel the dispatch can reach takes the static operand from the fragment-order
    image -- the LDS-staged kernels DMA the same 16-byte chunks from permuted addresses -- and the result is BIT-IDENTICAL to the
    row-major path: auto dispatch at the tower's shapes (fc2's ping-pong kernel, the small-grid 128 x 128 kernels, N % 256 != 0) and
    forced tiles (4 / 9 ping-pong, 3 / 15 / 18 / 1 lock-step; 11 / 5 cannot read the image and are re-routed to kernels that can -- as is 7,
    the 32x32x16 variant, whose own epilogue arithmetic differs from the others' in the last bit and is therefore not compared here)."""
    from slime_amd import ops, _lib
    a = _rand((M, K), dtype, dev, 1)
    w = _rand((N, K), dtype, dev, 2, K ** -0.5)
    bias = _rand((N,), torch.float32, dev, 3)
    wf = ops.pack_b_frag(w)
    assert wf is not None
    h0 = _rand((M, N), torch.float32, dev, 4, 2.0)
    with _lib.diag() as lib:
        lib.slime_gemm_force_tile(tile)
        try:
            for epi in (_lib.EPI_BIAS_F32, _lib.EPI_BIAS_T, _lib.EPI_BIAS_QUICKGELU_T):
                want = ops.gemm(a, w, bias, epi)
                assert torch.equal(ops.gemm(a, None, bias, epi, w_frag=wf), want), epi
            h1, h2 = h0.clone(), h0.clone()
            x1, s1 = ops.gemm_ln_producer(a, w, bias, h1)
            x2, s2 = ops.gemm_ln_producer(a, None, bias, h2, w_frag=wf)
            assert torch.equal(h1, h2) and torch.equal(x1, x2) and torch.equal(s1, s2)
        finally:
            lib.slime_gemm_force_tile(0)
    with pytest.raises(ValueError, match="static operand"):
        ops.gemm(a, None, bias, _lib.EPI_BIAS_F32, out=torch.empty((M, N), dtype=torch.float32, device=dev))
    g = _lib.GemmArgs(A=a.data_ptr(), lda=K, bias=bias.data_ptr(), C=h0.data_ptr(), ldc=N, M=M, N=N, K=K, dtype=ops.dtype_code(dtype),
                      epilogue=_lib.EPI_BIAS_F32)
    import ctypes
    assert _lib.load().slime_gemm_ex(ctypes.byref(g), 0) == -1 and b"null pointer" in _lib.load().slime_last_error()     # C ABI: B and B_frag both NULL


def test_96_row_direct_b_tile_is_a_bit_identical_alternative(dev):
    """Round 6: the 96 x 256 direct-B tile (diagnostic tile 19) was measured and NOT adopted (profiles/r06_db96_*_ab.txt): the product
    dispatch keeps 128-row tiles at every shape the rule had fired on; forced, the tile is bit-invisible -- the split-residual update
    of 9 crops' rows on 96-row tiles equals the same rows of a 12-crop launch on 128-row tiles (both planes + LayerNorm partial sums),
    and a LayerNorm-fold consumer launch (q/k/v shape) equals the product's."""
    from slime_amd import ops, _lib
    dt = torch.bfloat16
    name = lambda M, N, K, epi: ops.gemm_kernel_name(M, N, K, dt, epi, True)
    split = _lib.EPI_BIAS_RESID_SPLIT_LN
    for crops in (8, 9, 10, 11, 12, 13):
        assert name(577 * crops, 1024, 4096, split) == "gemm_db_kernel<BF16, 8, 1, 8>", crops
    assert name(2885, 3072, 1024, _lib.EPI_BIAS_T).endswith(", 8>") and name(11540, 1024, 1024, split).endswith(", 8>")
    assert "gemm_pp_kernel" in name(11540, 1024, 4096, split)
    M9, M12 = 9 * 577, 12 * 577
    a2 = _rand((M12, 4096), dt, dev, 4)
    w2 = _rand((1024, 4096), dt, dev, 5, 1 / 64)
    b2 = _rand((1024,), torch.float32, dev, 6)
    wf2 = ops.pack_b_frag(w2)
    h0 = _rand((M12, 1024), torch.float32, dev, 7, 2.0)
    hi0, lo0 = ops.resid_split(h0, dt)
    hi_a, lo_a = hi0.clone(), lo0.clone()
    st_a = ops.gemm_resid_split(a2, None, b2, hi_a, lo_a, w_frag=wf2)                      # product dispatch: 128-row tiles
    a3 = _rand((5 * 577, 1024), dt, dev, 8)
    w3 = _rand((3072, 1024), dt, dev, 9, 1 / 32)
    b3 = _rand((3072,), torch.float32, dev, 10)
    wf3 = ops.pack_b_frag(w3)
    want3 = ops.gemm(a3, None, b3, _lib.EPI_BIAS_T, w_frag=wf3)
    with _lib.diag() as lib:
        lib.slime_gemm_force_tile(19)
        try:
            hi_b, lo_b = hi0[:M9].clone(), lo0[:M9].clone()
            st_b = ops.gemm_resid_split(a2[:M9].contiguous(), None, b2, hi_b, lo_b, w_frag=wf2)
            got3 = ops.gemm(a3, None, b3, _lib.EPI_BIAS_T, w_frag=wf3)
        finally:
            lib.slime_gemm_force_tile(0)
    assert torch.equal(hi_a[:M9], hi_b) and torch.equal(lo_a[:M9], lo_b) and torch.equal(st_a[:M9], st_b)
    assert torch.equal(got3, want3)


def test_small_grid_dispatch_uses_64_row_tiles(dev):
    """auto_tile (round 5): grids that leave half the CUs without a 128 x 128 workgroup -- one crop's GEMMs -- take the 64 x 64 ring
    tile; from one workgroup per two CUs on, the 128 x 128 kernels as before; and the tile is bit-invisible (one crop inside a batch of
    three equals the crop alone: test_tower_batch_invariance covers the tower, this the GEMM)."""
    from slime_amd import ops, _lib
    dt = torch.bfloat16
    assert "64, 64, 4, 1" in ops.gemm_kernel_name(577, 1024, 4096, dt, _lib.EPI_BIAS_F32)
    assert "64, 64, 4, 1" in ops.gemm_kernel_name(577, 3072, 1024, dt, _lib.EPI_BIAS_T)
    assert "128, 128, 2, 2" in ops.gemm_kernel_name(2885, 1024, 1024, dt, _lib.EPI_BIAS_F32)
    a = _rand((3 * 577, 1024), dt, dev, 1)
    w = _rand((1024, 1024), dt, dev, 2, 1 / 32)
    bias = _rand((1024,), torch.float32, dev, 3)
    wf = ops.pack_b_frag(w)
    big = ops.gemm(a, None, bias, _lib.EPI_BIAS_QUICKGELU_T, w_frag=wf)
    one = ops.gemm(a[577:1154].contiguous(), None, bias, _lib.EPI_BIAS_QUICKGELU_T, w_frag=wf)
    assert torch.equal(big[577:1154], one)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,tile", [(9216, 4096, 4096, 0), (300, 768, 640, 0), (577, 1024, 1024, 0), (1731, 1024, 1024, 4), (1731, 1024, 1024, 3),
                                        (1731, 1024, 1024, 12), (1731, 1024, 1024, 11), (1731, 1024, 1024, 15), (577, 1024, 1024, 18)])
def test_gemm_row_map_scatter(dev, dtype, M, N, K, tile):
    """slime_gemm_args.row_map (round 5): output row r is stored at C row row_map[r] -- a scatter into a larger buffer, bit-identical to
    the plain result row by row, untouched rows stay untouched; plain T / fp32 epilogues on every kernel family (the adapter's
    projection[2] writes the token buffer this way: shape 9216 x 4096 x 4096)."""
    from slime_amd import ops, _lib
    a = _rand((M, K), dtype, dev, 1)
    w = _rand((N, K), dtype, dev, 2, K ** -0.5)
    bias = _rand((N,), torch.float32, dev, 3)
    wf = ops.pack_b_frag(w)
    g = torch.Generator().manual_seed(7)
    rows_out = M + 37
    rmap = torch.randperm(rows_out, generator=g)[:M].to(torch.int32).to(dev)
    with _lib.diag() as lib:
        lib.slime_gemm_force_tile(tile)
        try:
            for epi, odt in ((_lib.EPI_BIAS_T, dtype), (_lib.EPI_BIAS_GELU_T, dtype), (_lib.EPI_BIAS_F32, torch.float32)):
                plain = ops.gemm(a, w, bias, epi, w_frag=wf)
                out = torch.full((rows_out, N), 7.0, dtype=odt, device=dev)
                ops.gemm(a, None, bias, epi, out=out, w_frag=wf, row_map=rmap)
                assert torch.equal(out[rmap.long()], plain), epi
                untouched = torch.ones(rows_out, dtype=torch.bool, device=dev)
                untouched[rmap.long()] = False
                assert bool((out[untouched] == 7.0).all()), epi
        finally:
            lib.slime_gemm_force_tile(0)
    x16 = _rand((M, K), dtype, dev, 5)
    st = torch.zeros((M, K // 64, 2), dtype=torch.float32, device=dev)
    gargs = _lib.GemmArgs(A=x16.data_ptr(), lda=K, B_frag=wf.data_ptr(), bias=bias.data_ptr(), C=plain.data_ptr(), ldc=N, M=M, N=N, K=K,
                          dtype=ops.dtype_code(dtype), epilogue=_lib.EPI_BIAS_T, ln_stats=st.data_ptr(), ln_groups=K // 64,
                          ln_colsum=bias.data_ptr(), ln_eps=1e-5, row_map=rmap.data_ptr())
    import ctypes
    assert _lib.load().slime_gemm_ex(ctypes.byref(gargs), 0) == -1 and b"row_map" in _lib.load().slime_last_error()


def test_gather_rows_split(dev):
    """slime_gather_rows_split: the tower's final feature_select / cast from the split residual stream (T + one signed byte, ABI 7)."""
    from slime_amd import _lib, ops
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    n, S, D = 3, 577, 1024
    for dt, code in ((torch.bfloat16, _lib.BF16), (torch.float16, _lib.F16)):
        h = _rand((n * S, D), torch.float32, dev, 21, 3.0)
        hi, lo = ops.resid_split(h, dt)
        want = ops.resid_join(hi, lo).view(n, S, D)
        for odt, ocode in ((torch.float32, _lib.F32), (torch.bfloat16, _lib.BF16), (torch.float16, _lib.F16)):
            for off, rows in ((1, S - 1), (0, S)):
                out = torch.empty((n, rows, D), dtype=odt, device=dev)
                _lib.check(lib.slime_gather_rows_split(hi.data_ptr(), lo.data_ptr(), code, S, off, out.data_ptr(), ocode, n, rows, D, st))
                assert torch.equal(out, want[:, off:off + rows].to(odt))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gate_premix_hidden_rows(dev, dt):
    """slime_gate_premix (round 4): out = T(g0 a0 + g1 a1) on the MLP's hidden rows with slime_gate_mix's gates -- and, since
    projection[2] is linear with g0 + g1 = 1 / (1 + 1e-6), projection[2](out) equals the gate mix of the two expert OUTPUTS
    (projector/builder.py:190-206) to fp32 accuracy when the operands are the same.  In place over a1 as the adapter uses it."""
    from slime_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    for rows, D, H in ((576, 128, 256), (1153, 1024, 4096)):
        x = _rand((rows, D), torch.float32, dev, 31)
        wg = _rand((D, 2), torch.float32, dev, 32, 0.2)
        a0 = _rand((rows, H), torch.float32, dev, 33).to(dt)
        a1 = _rand((rows, H), torch.float32, dev, 34).to(dt)
        # reference gates in float64 on the host (torch's fp32 matmul on the device is not the yardstick for a max-norm bound)
        p = torch.softmax(x.double().cpu() @ wg.double().cpu(), 1)
        gts = p / (p.sum(1, keepdim=True) + 1e-6)
        want = (gts[:, :1] * a0.double().cpu() + gts[:, 1:] * a1.double().cpu())[:-1]
        out = a1.clone()                                                       # in place over a1
        guard = out.clone()
        _lib.check(lib.slime_gate_premix(x.data_ptr(), D, wg.data_ptr(), a0.data_ptr(), out.data_ptr(), out.data_ptr(),
                                         _lib.BF16 if dt == torch.bfloat16 else _lib.F16, rows - 1, H, st))
        assert torch.equal(out[-1], guard[-1])                                 # rows past `rows` untouched
        got = out[:-1].double().cpu()
        # one rounding to T of an fp32 value computed from the same operands: half a spacing of T (relative 2^-8 bf16, 2^-11 fp16) of
        # the result, plus the fp32 gate arithmetic (logit sums over D terms, expf: ~1e-5 relative) on each PRODUCT -- the two may cancel
        half = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        mag = (gts[:, :1] * a0.double().cpu()).abs()[:-1] + (gts[:, 1:] * a1.double().cpu()).abs()[:-1]
        excess = (got - want).abs() - (1.02 * half * want.abs() + 3e-5 * mag + 1e-7)
        i = int(excess.argmax())
        assert float(excess.max()) <= 0, (rows, D, H, float(excess.max()), float(got.flatten()[i]), float(want.flatten()[i]), gts[i // H].tolist())
        assert rel_l2(got.float(), want.float()) < (3e-3 if dt == torch.bfloat16 else 4e-4)
        # linearity: W2 (g0 a0 + g1 a1) + b2 == g0 (W2 a0 + b2) + g1 (W2 a1 + b2) up to 1e-6 |b2| (float64 reference arithmetic)
        W2 = _rand((64, H), torch.float32, dev, 35, H ** -0.5).double().cpu(); b2 = _rand((64,), torch.float32, dev, 36).double().cpu()
        a0d, a1d = a0.double().cpu()[:-1], a1.double().cpu()[:-1]
        mix_out = gts[:-1, :1] * (a0d @ W2.T + b2) + gts[:-1, 1:] * (a1d @ W2.T + b2)
        assert rel_l2((want @ W2.T + b2).float(), mix_out.float()) < 5e-6


def test_gate_mix_gather_merge(dev):
    from slime_amd import ops, _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    rows, D, H = 576, 128, 256
    x = _rand((rows, D), torch.float32, dev, 17)
    wg = _rand((D, 2), torch.float32, dev, 18, 0.2)
    e0 = _rand((rows, H), torch.float32, dev, 19)
    e1 = _rand((rows, H), torch.float32, dev, 20)
    out = torch.empty_like(e0)
    _lib.check(lib.slime_gate_mix(x.data_ptr(), D, wg.data_ptr(), e0.data_ptr(), e1.data_ptr(), out.data_ptr(), rows, H, st))
    p = torch.softmax(x @ wg, 1)
    gts = p / (p.sum(1, keepdim=True) + 1e-6)
    assert rel_l2(out.cpu(), (e0 * gts[:, :1] + e1 * gts[:, 1:]).cpu()) < TOL_F32
    # gather: drop the class token, cast
    h = _rand((3, 577, D), torch.float32, dev, 21)
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        o = torch.empty((3, 576, D), dtype=dt, device=dev)
        ops.gather_rows(h, o, 577, 1, 3, 576)
        assert torch.equal(o.cpu(), h[:, 1:].to(dt).cpu())
    # spatial merge (llava_arch.py:235-244) for a 2 x 3 grid of 12 x 12 tokens
    nw, nh, g = 2, 3, 12
    loc = _rand((nw * nh, g * g, H), torch.float32, dev, 22)
    o = torch.zeros((5 + nw * nh * g * g, H), dtype=torch.float32, device=dev)
    ops.merge_rows(loc, o, 5, nw, nh, g, True)
    ref = loc.view(nh, nw, g, g, H).permute(0, 2, 1, 3, 4).reshape(-1, H)
    assert torch.equal(o[5:].cpu(), ref.cpu()) and float(o[:5].abs().max()) == 0
    ops.merge_rows(loc, o, 5, nw, nh, g, False)
    assert torch.equal(o[5:].cpu(), loc.view(-1, H).cpu())


def test_tile_normalize(dev):
    from slime_amd import ops
    g = torch.Generator().manual_seed(23)
    canvas = torch.randint(0, 256, (672, 1008, 3), generator=g, dtype=torch.uint8)
    mean, std = [0.48145466, 0.4578275, 0.40821073], [0.26862954, 0.26130258, 0.27577711]
    out = ops.tile_normalize(canvas.to(dev), 336, mean, std, torch.float32)
    x = canvas.float() * (1.0 / 255.0)
    x = (x - torch.tensor(mean)) / torch.tensor(std)
    ref = x.view(2, 336, 3, 336, 3).permute(0, 2, 4, 1, 3).reshape(6, 3, 336, 336)
    assert float((out.cpu() - ref).abs().max()) < 2e-6
    outh = ops.tile_normalize(canvas.to(dev), 336, mean, std, torch.bfloat16)
    assert rel_l2(outh.float().cpu(), ref) < TOL_T[torch.bfloat16]


# -------------------------------------------------------------------------------------- attention
LOG2E = 1.4426950408889634


def _attn_ref(q, k, v, heads, dh):
    B, nq = k.shape[0], q.shape[1]
    qf = q.float().expand(B, -1, -1).reshape(B, nq, heads, dh).transpose(1, 2)
    kf = k.float().reshape(B, -1, heads, dh).transpose(1, 2)
    vf = v.float().reshape(B, -1, heads, dh).transpose(1, 2)
    att = torch.softmax(qf @ kf.transpose(-1, -2) / LOG2E, -1)   # q carries dh^-0.5 * log2(e): logits in log2 units
    return (att @ vf).transpose(1, 2).reshape(B, nq, heads * dh)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,heads,dh,nq,nkv,shared_q", [
    (3, 2, 64, 577, 577, False),     # CLIP self-attention geometry (ragged: 577 = 18*32 + 1)
    (2, 16, 64, 577, 577, False),    # 16 heads
    (1, 1, 64, 40, 100, False),      # short, ragged kv
    (2, 2, 64, 700, 1300, False),    # kv longer than one LDS chunk (608)
    (4, 2, 128, 144, 576, True),     # post_qformer: 144 shared queries x 576 keys
    (2, 8, 128, 576, 576, True),     # GatedBlock.attn: 576 queries (q-split over two workgroups)
    (1, 1, 128, 33, 64, False),
])
def test_attention(dev, dtype, B, heads, dh, nq, nkv, shared_q):
    from slime_amd import ops
    E = heads * dh
    q = _rand((1 if shared_q else B, nq, E), dtype, dev, 30, dh ** -0.5 * LOG2E)
    k = _rand((B, nkv, E), dtype, dev, 31)
    v = _rand((B, nkv, E), dtype, dev, 32)
    out = ops.attention(q, k, v, heads, dh)
    ref = _attn_ref(q, k, v, heads, dh)
    assert rel_l2(out.float().cpu(), ref.cpu()) < TOL_T[dtype] * 1.5   # + P rounded to T before PV


@pytest.mark.parametrize("B,heads,nq,nkv,gain", [
    (2, 2, 577, 577, 1.0),      # CLIP geometry: 19 query blocks -> 5,5,5,4 per wave; ragged last key step (one live key)
    (3, 4, 577, 577, 12.0),     # logits with std 12: the speculative softmax is thrown away and recomputed many times
    (1, 1, 577, 321, 1.0),      # shortest panel the kernel takes (11 key steps, ragged)
    (2, 3, 100, 400, 1.0),      # 4 blocks -> one per wave (the NB = 1 pipeline), half-dead last step
    (2, 2, 33, 608, 2.0),       # 2 blocks: two idle waves that only take part in the barriers; full panel, no ragged step
    (1, 2, 640, 576, 1.0),      # 20 blocks -> 5 per wave, even number of key steps
    (2, 2, 300, 577, 6.0),      # 10 blocks -> 3,3,2,2
    (17, 16, 577, 577, 1.0),    # 272 (crop, head) items: one round of 256 uncut + 16 cut by query blocks
])
def test_attention32_launch_forms(dev, B, heads, nq, nkv, gain):
    """attn32 (one wave per SIMD, 32x32x16 MFMAs, hand-placed softmax stream), the measured alternative of
    the CLIP attention kernel in the DIAGNOSTIC build: 4 = tail cutting, 5 = every item cut in two, 6 = uncut.  Same bar as the
    product kernel (0 = 7 = attn64r), plus a spiked key (the runaway path late in the sweep).  Recorded here because it decides
    what can ship: the cut forms do NOT reproduce the uncut form bit for bit on large batches (round 3: isolated rows whose
    reference maximum moves, tools/attn32_cut_invariance.py), so a batch-size dependent cut would break the tower's shard
    invariance."""
    from slime_amd import ops, _lib
    E = heads * 64
    qkv = _rand((B, max(nq, nkv), 3 * E), torch.bfloat16, dev, 50)
    qkv[..., :E] *= gain * 0.125 * LOG2E
    if nkv > 500 and nq > 17:
        qkv[0, 500, E:2 * E] = (qkv[0, 17, :E].float() * 60.0 / gain).to(torch.bfloat16)
    q, k, v = qkv[:, :nq, :E], qkv[:, :nkv, E:2 * E], qkv[:, :nkv, 2 * E:]
    ref = _attn_ref(q, k, v, heads, 64)
    outs = {}
    with _lib.diag() as lib:
        try:
            for variant in (0, 4, 5, 6, 7):
                lib.slime_attention_set_variant(variant)
                outs[variant] = ops.attention(q, k, v, heads, 64)
                torch.cuda.synchronize()
        finally:
            lib.slime_attention_set_variant(0)
    for variant, out in outs.items():
        assert torch.isfinite(out.float()).all(), variant
        assert rel_l2(out.float().cpu(), ref.cpu()) < 6e-3, variant
    assert torch.equal(outs[7], outs[0]) and torch.equal(outs[0], ops.attention(q, k, v, heads, 64))       # the product kernel


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("gain", [1.0, 6.0, 12.0])
def test_attention_launch_form_invariance(dev, dtype, gain):
    """The launcher gives a CLIP (crop, head) one workgroup or two depending on the batch size (rounds of CUs x cost), which changes
    which query sub-blocks share a wave.  The lazily refreshed softmax reference used to be refreshed per WAVE (any sub-block of the
    wave running away moved all of them): with logits large enough to trigger it (real checkpoints have such heads) the SAME crop
    came out different in the last bits at different batch sizes -- found in round 4 by the ring alternative below, which groups
    sub-blocks differently again.  Now per sub-block: 5 crops alone (two workgroups per item), the same 5 inside 10 (one) and
    inside 20 (two) are bit-identical at every logit scale."""
    from slime_amd import ops
    heads, E = 16, 1024
    qkv = _rand((20, 577, 3 * E), dtype, dev, 52)
    qkv[..., :E] *= gain * 0.125 * LOG2E
    outs = {}
    for n in (5, 10, 20):
        q, k, v = qkv[:n, :, :E], qkv[:n, :, E:2 * E], qkv[:n, :, 2 * E:]
        outs[n] = ops.attention(q, k, v, heads, 64)
    assert torch.equal(outs[5], outs[10][:5]) and torch.equal(outs[5], outs[20][:5]) and torch.equal(outs[10], outs[20][:10])
    assert rel_l2(outs[5].float().cpu(), _attn_ref(qkv[:5, :, :E], qkv[:5, :, E:2 * E], qkv[:5, :, 2 * E:], heads, 64).cpu()) < 6e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,heads,nq,nkv,gain", [
    (2, 16, 577, 577, 1.0),     # CLIP geometry: 37 query sub-blocks -> four workgroups of 10 / 10 / 10 / 7 per (crop, head)
    (3, 4, 577, 577, 12.0),     # logits with std 12: the reference maximum moves often
    (1, 1, 577, 321, 1.0),      # shortest panel (11 steps: the ring wraps once), one item = a padded group of eight
    (2, 3, 100, 400, 1.0),      # 7 sub-blocks: one workgroup, waves with 2 / 2 / 2 / 1
    (2, 2, 33, 608, 2.0),       # 3 sub-blocks: an idle wave that only streams and takes part in the barriers; full panel
    (5, 16, 577, 577, 1.0),     # 80 items
])
def test_attention_ring_alternative(dev, dtype, B, heads, nq, nkv, gain):
    """Round 4's measured alternative (diagnostic variant 30, attn64g_kernel): K/V through a 2 x 32 KiB ring, four waves, two
    workgroups per CU.  Same arithmetic in the same order per query sub-block: BIT-EQUAL to the product kernel on every shape
    (and 6-12 % faster stand-alone at 20-40 crops; in the two-stream tower it is 1.3 % slower, profiles/r04_attention_ring.txt)."""
    from slime_amd import ops, _lib
    E = heads * 64
    qkv = _rand((B, max(nq, nkv), 3 * E), dtype, dev, 51)
    qkv[..., :E] *= gain * 0.125 * LOG2E
    q, k, v = qkv[:, :nq, :E], qkv[:, :nkv, E:2 * E], qkv[:, :nkv, 2 * E:]
    want = ops.attention(q, k, v, heads, 64)
    with _lib.diag() as lib:
        try:
            lib.slime_attention_set_variant(30)
            got = ops.attention(q, k, v, heads, 64)
            torch.cuda.synchronize()
        finally:
            lib.slime_attention_set_variant(0)
    assert torch.equal(got, want)
    assert rel_l2(got.float().cpu(), _attn_ref(q, k, v, heads, 64).cpu()) < 6e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("gain", [4.0, 12.0])
def test_attention_large_logits(dev, dtype, gain):
    """Logits with std 4 / 12: the reference max of the lazy-rescale softmax is refreshed many times,
    and scores sit up to 8 log2-units above it in between (p up to 256 before normalisation)."""
    from slime_amd import ops
    B, S, heads, dh = 3, 577, 4, 64
    E = heads * dh
    q = _rand((B, S, E), dtype, dev, 40, gain * dh ** -0.5 * LOG2E)
    k = _rand((B, S, E), dtype, dev, 41)
    v = _rand((B, S, E), dtype, dev, 42)
    out = ops.attention(q, k, v, heads, dh)
    ref = _attn_ref(q, k, v, heads, dh)
    assert torch.isfinite(out.float()).all()
    assert rel_l2(out.float().cpu(), ref.cpu()) < TOL_T[dtype] * 1.5


def test_attention_packed_qkv_and_spike(dev):
    """q/k/v as thirds of one packed [B,S,3E] buffer (the tower's layout), and a forced running-max
    jump: one key row spiked against one query so the online-softmax rescale branch is exercised
    late in the kv sweep (cdna guide rule 26)."""
    from slime_amd import ops
    B, S, heads, dh = 2, 577, 2, 64
    E = heads * dh
    qkv = _rand((B, S, 3 * E), torch.bfloat16, dev, 33)
    qkv[..., :E] *= dh ** -0.5 * LOG2E
    qkv[0, 500, E:2 * E] = qkv[0, 17, :E] * 60.0            # k[500] aligned with q[17]: huge logit at kv=500
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    out = ops.attention(q, k, v, heads, dh)
    ref = _attn_ref(q, k, v, heads, dh)
    assert torch.isfinite(out.float()).all()
    assert rel_l2(out.float().cpu(), ref.cpu()) < 6e-3
    assert rel_l2(out[0, 17].float().cpu(), ref[0, 17].cpu()) < 6e-3
