"""GPU parity tests of SURVEY.md section 8 row f-2 (BASELINE configs 4 / 5): the visual-token splice (bit-exact) and the Llama-3
prefill attention (RoPE + causal GQA + projections) against oracle/prefill_oracle.py and the reference-generated vectors of
tests/golden/prefill.npz.  Tolerances as everywhere (rel-L2 vs fp32): per stage fp16 3e-3, bf16 1.5e-2."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
import prefill_fixture as F

pytestmark = pytest.mark.gpu
TOL = {torch.float16: 3e-3, torch.bfloat16: 1.5e-2}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from slime_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------------------------ splice
@pytest.mark.parametrize("name", ["A", "B", "C"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_splice_kernel_bit_exact_vs_reference(dev, name, dtype):
    """One launch reproduces the reference's new_input_embeds bit for bit (same-dtype rows are raw copies; the reference's
    fp32 vectors rounded to a 16-bit table / feature dtype are still exact copies of the rounded inputs)."""
    from slime_amd import ops
    from slime_amd.model.llava_arch import splice_plan
    c = F.splice_case(F.load(), name)
    am = None if c["attention_mask"] is None else c["attention_mask"].numpy()
    lb = None if c["labels"] is None else c["labels"].numpy()
    src, _, _, _ = splice_plan(c["input_ids"].numpy(), am, lb, [f.shape[0] for f in c["feats"]], c["max_length"], c["padding_side"])
    table = c["table"].to(dtype).to(dev)
    allf = torch.cat(c["feats"], 0).to(dtype).to(dev)
    out = ops.splice_rows(table, allf, torch.from_numpy(src).reshape(-1).to(dev), dtype).view(*src.shape, -1)
    s = torch.from_numpy(src)
    want = torch.zeros(src.shape + (table.shape[1],), dtype=dtype)
    want[s >= 0] = table.cpu()[s[s >= 0]]
    want[s <= -2] = allf.cpu()[-2 - s[s <= -2]]
    assert torch.equal(out.cpu(), want)
    if dtype == torch.float32:
        assert torch.equal(out.cpu(), c["out_embeds"])
    # mixed dtypes: fp32 features into a bf16 table's output (cast path)
    mixed = ops.splice_rows(c["table"].to(torch.bfloat16).to(dev), torch.cat(c["feats"], 0).to(dev),
                            torch.from_numpy(src).reshape(-1).to(dev), torch.bfloat16).view(*src.shape, -1)
    assert torch.equal(mixed.cpu(), c["out_embeds"].to(torch.bfloat16))


def test_prepare_inputs_labels_for_multimodal_end_to_end(dev):
    """The reference-shaped method on the tiny encoder: encode_images (HIP tower + adapter + router) feeds the splice;
    checked against the oracle's splice of the very same visual tokens, and the None-passing contract of :456-470."""
    import torch.nn as nn
    from oracle import prefill_oracle as P
    from test_gpu_modules import _tiny_encoder
    embed = nn.Embedding(2048, 256)
    embed.weight.data.copy_(torch.randn(2048, 256, generator=torch.Generator().manual_seed(3)) * 0.5)
    enc, _, _ = _tiny_encoder(dev, torch.bfloat16, embed=embed)
    from slime_amd import weights as W
    px = [W.synthetic_pixels(3, seed=61).to(dev), W.synthetic_pixels(3, seed=62).to(dev)]
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1, 2000, (2, 12), generator=g)
    ids[0, 3] = -200
    ids[1, 0] = -200
    am = torch.ones_like(ids)
    am[1, 9:] = 0
    lab = ids.clone()
    ids_d, am_d, lab_d = ids.to(dev), am.to(dev), lab.to(dev)
    sizes = [(336, 336), (336, 336)]
    feats, _ = enc.encode_images(torch.cat(px, 0), ids_d, [3, 3], am_d, None, sizes, labels=lab_d)
    out = enc.prepare_inputs_labels_for_multimodal(ids_d, None, am_d, None, lab_d, px, image_sizes=sizes)
    none_ids, pos, mask, pkv, emb, labels = out
    assert none_ids is None and pos is None and pkv is None
    want, wl, wm, _ = P.splice(embed.weight.detach().cpu(), [f[0].float().cpu() for f in feats], ids, am, lab)
    assert emb.shape == want.shape and torch.equal(emb.float().cpu(), want)
    assert torch.equal(labels.cpu(), wl) and torch.equal(mask.cpu(), wm)
    # text-only step (decode: one token) and image-free call return the inputs untouched
    one = enc.prepare_inputs_labels_for_multimodal(ids_d[:, :1], None, am_d[:, :1], None, None, px)
    assert one[0] is not None and one[4] is None
    # a token id beyond the embedding table raises like nn.Embedding does in the reference (llava_arch.py:373), instead of an
    # out-of-bounds device read
    bad = ids.clone()
    bad[0, 5] = 10 ** 6
    with pytest.raises(IndexError, match="index out of range"):
        enc.prepare_inputs_labels_for_multimodal(bad.to(dev), None, am_d, None, lab_d, px, image_sizes=sizes)
    neg = ids.clone()
    neg[0, 5] = -7                                  # a negative id other than IMAGE_TOKEN_INDEX would be read as an image-feature row
    with pytest.raises(IndexError, match="index out of range"):
        enc.prepare_inputs_labels_for_multimodal(neg.to(dev), None, am_d, None, lab_d, px, image_sizes=sizes)


# ------------------------------------------------------------------------------------------------ RoPE / attention
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_rope_kernel_vs_oracle(dev, dtype):
    from slime_amd import ops, _lib
    from oracle import prefill_oracle as P
    lib = _lib.load()
    B, S, HQ, HKV = 2, 77, 8, 2
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(B, S, (HQ + 2 * HKV) * 128, generator=g).to(dtype)
    pos = torch.randint(0, 8000, (B, S), generator=g)
    inv = ops.llama_inv_freq(128, 500000.0)
    cos, sin = P.rope_tables(pos, 128, 500000.0)
    x = qkv.float().view(B, S, HQ + 2 * HKV, 128).transpose(1, 2)
    scale = 128 ** -0.5 * ops.LOG2E
    want = x.clone()
    want[:, :HQ + HKV] = P.apply_rope(x[:, :HQ + HKV], cos, sin)
    want[:, :HQ] *= scale
    d = qkv.to(dev).clone()
    pos_d, inv_d = pos.to(torch.int32).to(dev), inv.to(dev)             # keep the device copies alive across the launch
    _lib.check(lib.slime_rope(d.data_ptr(), d.shape[-1], pos_d.data_ptr(), B * S, HQ + HKV, HQ, 128, inv_d.data_ptr(), scale,
                              ops.dtype_code(dtype), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = d.float().cpu().view(B, S, HQ + 2 * HKV, 128).transpose(1, 2)
    assert rel_l2(got[:, :HQ + HKV], want[:, :HQ + HKV]) < {torch.bfloat16: 4e-3, torch.float16: 6e-4}[dtype]
    assert torch.equal(got[:, HQ + HKV:], x[:, HQ + HKV:])                       # v untouched


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("mode", ["nopad", "right", "left"])
def test_llama_attention_vs_hf_golden_and_oracle(dev, dtype, mode):
    """slime_llama_attn_forward (projection GEMM + RoPE + causal GQA + o_proj) vs the HF LlamaAttention vectors and, on
    the full tensor, vs the oracle; rows at padded positions are exactly zero (pad_input)."""
    from slime_amd.model.language_model import HipLlamaAttention
    from oracle import prefill_oracle as P
    g = F.load()
    (D, HQ, HKV, S, B), w, hidden = F.llama_inputs(g)
    m = HipLlamaAttention(D, HQ, HKV, 128, float(g["llama_theta"][0]), compute_dtype=dtype)
    m.load_state_dict({f"{k}.weight": v for k, v in w.items()})
    m.to(dev)
    mask = torch.from_numpy(g[f"llama_{mode}_mask"])
    pos = torch.from_numpy(g[f"llama_{mode}_pos"])
    out, _, _ = m(hidden.to(dev), attention_mask=None if mode == "nopad" else mask.to(dev), position_ids=pos.to(dev))
    assert out.dtype == torch.float32 and out.shape == (B, S, D)
    out = out.cpu()
    assert rel_l2(out[:, ::3, ::7], g[f"llama_{mode}_out"]) < TOL[dtype]
    ref = P.llama_attention_forward(hidden, w["q_proj"], w["k_proj"], w["v_proj"], w["o_proj"], HQ, HKV, pos,
                                    None if mode == "nopad" else mask, float(g["llama_theta"][0]))
    assert rel_l2(out, ref) < TOL[dtype]
    worst = ((out - ref).norm(dim=-1) / ref.norm(dim=-1).clamp_min(1e-6))[mask.bool()].max()
    assert float(worst) < 6 * TOL[dtype], "no single token row far off (catches a wrong row hiding in the aggregate)"
    if mode != "nopad":
        assert float(out[mask == 0].abs().max()) == 0.0


def test_prefill_attention_causality_and_group_mapping(dev):
    """Size-independent properties at the Llama-3-8B shape (32 query / 8 kv heads, S = 1300): (i) causal -- changing tokens
    after position t leaves rows <= t bit-identical; (ii) GQA -- the 4 query heads of a kv group given IDENTICAL queries
    produce identical outputs, and a different kv head's keys do not leak in; (iii) a sequence's result does not depend on
    what shares the batch with it."""
    from slime_amd import ops
    B, S, HQ, HKV = 2, 1300, 32, 8
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(8)
    N = (HQ + 2 * HKV) * 128
    qkv = (torch.randn(B, S, N, generator=g) * 0.5).to(dt).to(dev)
    qkv[..., :HQ * 128] *= 0.1

    def run(x, start=None, length=None):
        lib = ops._lib.load()
        o = torch.empty((x.shape[0], S, HQ * 128), dtype=dt, device=dev)
        ops._lib.check(lib.slime_prefill_attention(x.data_ptr(), S * N, N, x.data_ptr() + HQ * 256, S * N, N,
                                                  x.data_ptr() + (HQ + HKV) * 256, S * N, N, o.data_ptr(), S * HQ * 128, HQ * 128,
                                                  x.shape[0], HQ, HKV, 128, S, None if start is None else start.data_ptr(),
                                                  None if length is None else length.data_ptr(), ops.dtype_code(dt),
                                                  torch.cuda.current_stream().cuda_stream))
        return o

    base = run(qkv)
    assert torch.isfinite(base.float()).all()
    t = 700
    mod = qkv.clone()
    mod[:, t + 1:] = torch.randn(B, S - t - 1, N, generator=g).to(dt).to(dev)
    assert torch.equal(run(mod)[:, :t + 1], base[:, :t + 1])                                   # (i)
    same = qkv.clone()
    same[..., :HQ * 128] = same[..., :128].repeat(1, 1, HQ)                                      # every q head = head 0's q
    o = run(same).view(B, S, HQ, 128)
    for h in range(1, 4):
        assert torch.equal(o[:, :, h], o[:, :, 0])                                               # (ii) same kv group
    assert not torch.equal(o[:, :, 4], o[:, :, 0])                                               #      next group: other keys
    assert torch.equal(run(qkv[1:2].contiguous())[0], base[1])                                   # (iii)
    # token ranges: left-padded sequence == the same tokens run without padding (RoPE already in the inputs)
    start = torch.tensor([0, 300], dtype=torch.int32, device=dev)
    length = torch.tensor([S, S - 300], dtype=torch.int32, device=dev)
    ranged = run(qkv, start, length)
    assert float(ranged[1, :300].float().abs().max()) == 0.0
    alone = run(torch.cat([qkv[1:2, 300:], qkv[1:2, :300]], 1).contiguous(), torch.tensor([0], dtype=torch.int32, device=dev),
                torch.tensor([S - 300], dtype=torch.int32, device=dev))
    assert rel_l2(ranged[1, 300:].float(), alone[0, :S - 300].float()) < 4e-3      # same math; chunk boundaries differ -> rounding only


def test_prefill_attention_video_length(dev):
    """BASELINE configs[4] (bench.py --config 5): ONE sequence of 64 text + 8 x 1152 visual positions = 9280.  Causal GQA
    attention at that length against plain fp32 torch (8 query / 2 kv heads keep the reference's 9280 x 9280 score matrices small)."""
    from slime_amd import ops
    B, S, HQ, HKV = 1, 9280, 8, 2
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(21)
    N = (HQ + 2 * HKV) * 128
    qkv = (torch.randn(B, S, N, generator=g) * 0.5).to(dt).to(dev)
    qkv[..., :HQ * 128] *= 128 ** -0.5 * 1.4426950408889634 * 2.0                                # logits with std ~1 (log2 units)
    lib = ops._lib.load()
    o = torch.empty((B, S, HQ * 128), dtype=dt, device=dev)
    ops._lib.check(lib.slime_prefill_attention(qkv.data_ptr(), S * N, N, qkv.data_ptr() + HQ * 256, S * N, N,
                                              qkv.data_ptr() + (HQ + HKV) * 256, S * N, N, o.data_ptr(), S * HQ * 128, HQ * 128,
                                              B, HQ, HKV, 128, S, None, None, ops.dtype_code(dt), torch.cuda.current_stream().cuda_stream))
    q = qkv[0, :, :HQ * 128].float().view(S, HQ, 128).transpose(0, 1) / 1.4426950408889634      # [HQ, S, 128], natural-log units
    k = qkv[0, :, HQ * 128:(HQ + HKV) * 128].float().view(S, HKV, 128).transpose(0, 1)
    v = qkv[0, :, (HQ + HKV) * 128:].float().view(S, HKV, 128).transpose(0, 1)
    causal = torch.ones(S, S, dtype=torch.bool, device=dev).tril()
    worst = 0.0
    for h in range(HQ):
        sc = (q[h] @ k[h // (HQ // HKV)].t()).masked_fill(~causal, float("-inf"))
        ref = torch.softmax(sc, -1) @ v[h // (HQ // HKV)]
        got = o[0, :, h * 128:(h + 1) * 128].float()
        worst = max(worst, float((got - ref).norm() / ref.norm()))
        rows = ((got - ref).norm(dim=-1) / ref.norm(dim=-1).clamp_min(1e-6))
        assert float(rows.max()) < 5e-2, "no single position far off"
    assert worst < 6e-3


def test_prefill_attention_huge_logits(dev):
    """Language-model logits are not bounded like CLIP's: after a few dozen random-weight layers they reach 1e5..1e6 (log2 units),
    where the reference maximum no longer fits a single 16-bit MFMA operand.  One-hot-sharp softmax rows must come out finite and
    equal to fp64 torch."""
    from slime_amd import ops
    B, S, HQ, HKV = 2, 600, 8, 2
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(33)
    N = (HQ + 2 * HKV) * 128
    qkv = (torch.randn(B, S, N, generator=g) * 0.5).to(dt).to(dev)
    qkv[..., :HQ * 128] *= 3000.0                                                                  # logits with std ~ 1.7e4
    lib = ops._lib.load()
    o = torch.empty((B, S, HQ * 128), dtype=dt, device=dev)
    ops._lib.check(lib.slime_prefill_attention(qkv.data_ptr(), S * N, N, qkv.data_ptr() + HQ * 256, S * N, N,
                                              qkv.data_ptr() + (HQ + HKV) * 256, S * N, N, o.data_ptr(), S * HQ * 128, HQ * 128,
                                              B, HQ, HKV, 128, S, None, None, ops.dtype_code(dt), torch.cuda.current_stream().cuda_stream))
    assert torch.isfinite(o.float()).all()
    q = qkv[..., :HQ * 128].double().view(B, S, HQ, 128).transpose(1, 2) / 1.4426950408889634
    k = qkv[..., HQ * 128:(HQ + HKV) * 128].double().view(B, S, HKV, 128).transpose(1, 2).repeat_interleave(HQ // HKV, 1)
    v = qkv[..., (HQ + HKV) * 128:].double().view(B, S, HKV, 128).transpose(1, 2).repeat_interleave(HQ // HKV, 1)
    causal = torch.ones(S, S, dtype=torch.bool, device=dev).tril()
    ref = (torch.softmax((q @ k.transpose(-1, -2)).masked_fill(~causal, float("-inf")), -1) @ v).transpose(1, 2).reshape(B, S, HQ * 128)
    err = float((o.double() - ref).norm() / ref.norm())
    assert err < 6e-3, err


def test_prefill_attention_is_deterministic(dev):
    """The product kernel for the Llama-3 geometry pads its MFMA latencies and counts its waits by hand: a margin that is too thin
    shows as run-to-run differences.  Same inputs, 6 launches while another stream keeps the GPU busy: bit-identical."""
    from slime_amd import ops
    B, S, HQ, HKV = 3, 1500, 32, 8
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(44)
    N = (HQ + 2 * HKV) * 128
    qkv = (torch.randn(B, S, N, generator=g) * 0.5).to(dt).to(dev)
    qkv[..., :HQ * 128] *= 0.2
    lib = ops._lib.load()
    start = torch.tensor([0, 37, 0], dtype=torch.int32, device=dev)
    length = torch.tensor([S, S - 37, S - 200], dtype=torch.int32, device=dev)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev)
    outs = []
    for i in range(6):
        if i % 2:
            with torch.cuda.stream(side):
                for _ in range(4):
                    a @ a                                                                          # noise on the other CUs
        o = torch.empty((B, S, HQ * 128), dtype=dt, device=dev)
        ops._lib.check(lib.slime_prefill_attention(qkv.data_ptr(), S * N, N, qkv.data_ptr() + HQ * 256, S * N, N,
                                                  qkv.data_ptr() + (HQ + HKV) * 256, S * N, N, o.data_ptr(), S * HQ * 128, HQ * 128,
                                                  B, HQ, HKV, 128, S, start.data_ptr(), length.data_ptr(), ops.dtype_code(dt),
                                                  torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        outs.append(o)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    assert torch.isfinite(outs[0].float()).all()


@pytest.mark.parametrize("B,S", [(12, 1500), (3, 4200), (40, 200)])
def test_prefill32_persistent_launch_equals_one_item_per_workgroup(dev, B, S):
    """Round 3: prefill32 walks several work items (sequence, kv head, 64 query rows) per workgroup -- the K/V ring, its look-ahead
    DMAs and the Q request run across the seam between items.  An item's arithmetic must not notice: the persistent launch is
    BIT-equal to the diagnostic launch with one item per workgroup (the round-2 form), with token ranges that start off a step
    boundary, end early (items with no live row get zeros and are skipped by the ring) or cover a single token, and with an odd
    number of key steps in many items."""
    from slime_amd import ops
    HQ, HKV = 32, 8
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(77 + B)
    N = (HQ + 2 * HKV) * 128
    qkv = (torch.randn(B, S, N, generator=g) * 0.5).to(dt).to(dev)
    qkv[..., :HQ * 128] *= 0.2
    lib = ops._lib.load_diag()
    st0 = [0, 37, 0, 64, S // 3, S - 1, 5, 0, 31, 33, 96, 1]
    ln0 = [S, S - 37, S - 200 if S > 200 else S, 100, S // 2, 1, 70, 64, 1, S, 97, S - 1]
    start = torch.tensor([st0[i % 12] for i in range(B)], dtype=torch.int32, device=dev)
    length = torch.tensor([max(1, min(ln0[i % 12], S - st0[i % 12])) for i in range(B)], dtype=torch.int32, device=dev)
    outs = {}
    try:
        for ranges in (True, False):
            for var in (0, 2):
                lib.slime_prefill_set_variant(var)
                o = torch.full((B, S, HQ * 128), 7.0, dtype=dt, device=dev)
                ops._lib.check(lib.slime_prefill_attention(qkv.data_ptr(), S * N, N, qkv.data_ptr() + HQ * 256, S * N, N,
                                                          qkv.data_ptr() + (HQ + HKV) * 256, S * N, N, o.data_ptr(), S * HQ * 128, HQ * 128,
                                                          B, HQ, HKV, 128, S, start.data_ptr() if ranges else None,
                                                          length.data_ptr() if ranges else None, ops.dtype_code(dt),
                                                          torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize()
                outs[(ranges, var)] = o
            assert torch.equal(outs[(ranges, 0)], outs[(ranges, 2)])
            assert torch.isfinite(outs[(ranges, 0)].float()).all()
    finally:
        lib.slime_prefill_set_variant(0)
    # padded rows are zeros, the rest is attention (spot check of one ranged sequence against fp32 torch)
    o = outs[(True, 0)]
    b = 1 % B
    lo, hi = int(start[b]), int(start[b]) + int(length[b])
    assert float(o[b, :lo].float().abs().max() if lo else 0) == 0.0 and float(o[b, hi:].float().abs().max() if hi < S else 0) == 0.0
    hsel = 5
    q = qkv[b, lo:hi, hsel * 128:(hsel + 1) * 128].float() / 1.4426950408889634
    k = qkv[b, lo:hi, (HQ + hsel // 4) * 128:(HQ + hsel // 4 + 1) * 128].float()
    v = qkv[b, lo:hi, (HQ + HKV + hsel // 4) * 128:(HQ + HKV + hsel // 4 + 1) * 128].float()
    causal = torch.ones(hi - lo, hi - lo, dtype=torch.bool, device=dev).tril()
    ref = torch.softmax((q @ k.T).masked_fill(~causal, float("-inf")), -1) @ v
    got = o[b, lo:hi, hsel * 128:(hsel + 1) * 128].float()
    assert float((got - ref).norm() / ref.norm()) < 6e-3


@pytest.mark.parametrize("S", [1, 5, 33, 64, 65, 127, 193])
def test_prefill_attention_short_sequences(dev, S):
    """Sequences shorter than a workgroup's 64 query rows / not a multiple of the 32-key step, with and without a token range."""
    from slime_amd import ops
    B, HQ, HKV = 2, 8, 2
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(50 + S)
    N = (HQ + 2 * HKV) * 128
    qkv = (torch.randn(B, S, N, generator=g) * 0.5).to(dt).to(dev)
    qkv[..., :HQ * 128] *= 0.2
    lib = ops._lib.load()
    start = torch.tensor([0, min(2, S - 1)], dtype=torch.int32, device=dev)
    length = torch.tensor([S, max(1, S - 3)], dtype=torch.int32, device=dev)
    for ranges in (False, True):
        o = torch.full((B, S, HQ * 128), 7.0, dtype=dt, device=dev)
        ops._lib.check(lib.slime_prefill_attention(qkv.data_ptr(), S * N, N, qkv.data_ptr() + HQ * 256, S * N, N,
                                                  qkv.data_ptr() + (HQ + HKV) * 256, S * N, N, o.data_ptr(), S * HQ * 128, HQ * 128,
                                                  B, HQ, HKV, 128, S, start.data_ptr() if ranges else None,
                                                  length.data_ptr() if ranges else None, ops.dtype_code(dt),
                                                  torch.cuda.current_stream().cuda_stream))
        assert torch.isfinite(o.float()).all()
        for b in range(B):
            lo = int(start[b]) if ranges else 0
            hi = min(S, lo + int(length[b])) if ranges else S
            q = qkv[b, lo:hi, :HQ * 128].float().view(hi - lo, HQ, 128).transpose(0, 1) / 1.4426950408889634
            k = qkv[b, lo:hi, HQ * 128:(HQ + HKV) * 128].float().view(hi - lo, HKV, 128).transpose(0, 1).repeat_interleave(HQ // HKV, 0)
            v = qkv[b, lo:hi, (HQ + HKV) * 128:].float().view(hi - lo, HKV, 128).transpose(0, 1).repeat_interleave(HQ // HKV, 0)
            causal = torch.ones(hi - lo, hi - lo, dtype=torch.bool, device=dev).tril()
            ref = (torch.softmax((q @ k.transpose(-1, -2)).masked_fill(~causal, float("-inf")), -1) @ v).transpose(0, 1).reshape(hi - lo, HQ * 128)
            got = o[b, lo:hi].float()
            assert float((got - ref).norm() / ref.norm()) < 6e-3
            if ranges:
                assert float(o[b, :lo].abs().max() if lo else 0) == 0.0 and float(o[b, hi:].abs().max() if hi < S else 0) == 0.0


# ------------------------------------------------------------------------------------------------ round 3
@pytest.fixture(scope="module")
def llama8b_case():
    """One attention sub-layer at Llama-3-8B dims (D = 4096, 32 q / 8 kv heads, dh 128), B = 2, S = 1216 (BASELINE config 4's
    sequence length), right padding on the second sequence; the fp32 oracle result (a few seconds of CPU)."""
    from oracle import prefill_oracle as P
    D, HQ, HKV, B, S = 4096, 32, 8, 2, 1216
    g = torch.Generator().manual_seed(41)
    w = [torch.randn(n, k, generator=g) * k ** -0.5 for n, k in ((HQ * 128, D), (HKV * 128, D), (HKV * 128, D), (D, HQ * 128))]
    hidden = torch.randn(B, S, D, generator=g)
    mask = torch.ones(B, S, dtype=torch.int64)
    mask[1, 900:] = 0
    pos = torch.arange(S)[None].expand(B, S).contiguous()
    ref = P.llama_attention_forward(hidden, w[0], w[1], w[2], w[3], HQ, HKV, pos, mask)
    return (D, HQ, HKV, B, S), w, hidden, mask, pos, ref


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_llama_attention_real_size_vs_oracle(dev, llama8b_case, dtype):
    """slime_llama_attn_forward at the shape bench.py --config 4 / 5 times -- the fused q/k/v GEMM
    (N = 6144, K = 4096) -> RoPE -> prefill32 / eight-wave attention -> o_proj chain, right padding -- against
    oracle/prefill_oracle.llama_attention_forward (llama_flash_attn_monkey_patch.py:16-93) on the full tensor."""
    from slime_amd import ops
    (D, HQ, HKV, B, S), w, hidden, mask, pos, ref = llama8b_case
    pa = ops.pack_llama_attention(w[0], w[1], w[2], w[3], HQ, HKV, dtype, dev)
    assert pa.tensors["w_qkv_frag"] is not None and pa.tensors["w_o_frag"] is not None
    out = ops.llama_attention_forward(pa, hidden.to(dev), pos.to(dev), mask.to(dev), torch.float32).cpu()
    assert rel_l2(out, ref) < TOL[dtype]
    worst = ((out - ref).norm(dim=-1) / ref.norm(dim=-1).clamp_min(1e-6))[mask.bool()].max()
    assert float(worst) < 6 * TOL[dtype]
    assert float(out[mask == 0].abs().max()) == 0.0
    # the fused-residual form (decoder layer's residual add in o_proj's epilogue, 16-bit stream as in HF):
    # out = T(resid + attn) with ONE rounding -- equal to rounding (resid + fp32 attention output) once
    hid_t = hidden.to(dev).to(dtype)
    resid = (torch.randn(B, S, D, generator=torch.Generator().manual_seed(3)) * 2).to(dtype).to(dev)
    out32 = ops.llama_attention_forward(pa, hid_t, pos.to(dev), mask.to(dev), torch.float32)
    fused = ops.llama_attention_forward_resid(pa, hid_t, resid, pos.to(dev), mask.to(dev))
    assert fused.dtype == dtype and torch.equal(fused, (resid.float() + out32).to(dtype))
    inplace = resid.clone()
    ops.llama_attention_forward_resid(pa, hid_t, inplace, pos.to(dev), mask.to(dev), out=inplace)      # out aliases resid
    assert torch.equal(inplace, fused)


@pytest.mark.parametrize("form", ["row", "vector", "none"])
def test_llama_attention_position_ids_broadcast(dev, form):
    """HF hands the attention position_ids of shape [1, S] (or None) for a batch of B sequences and lets
    cos[position_ids] broadcast; the kernels index one entry per row.  B = 3 with a [1, S] row, a 1-D vector and None must
    equal the explicit [B, S] call bit for bit -- through HipLlamaAttention and through the monkey-patched HF forward."""
    from slime_amd.model.language_model import HipLlamaAttention
    D, HQ, HKV, B, S = 1024, 8, 2, 3, 70
    g = torch.Generator().manual_seed(5)
    m = HipLlamaAttention(D, HQ, HKV, 128, 500000.0, compute_dtype=torch.bfloat16)
    with torch.no_grad():
        for p_ in m.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * p_.shape[1] ** -0.5)
    m.to(dev)
    hidden = torch.randn(B, S, D, generator=g).to(dev)
    full = torch.arange(S, device=dev)[None].expand(B, S).contiguous()
    want, _, _ = m(hidden, position_ids=full)
    pid = {"row": torch.arange(S, device=dev)[None], "vector": torch.arange(S, device=dev), "none": None}[form]
    got, _, _ = m(hidden, position_ids=pid)
    assert torch.equal(got, want)
    with pytest.raises(ValueError, match="broadcast"):
        m(hidden, position_ids=torch.arange(S, device=dev)[None].expand(2, S))
    # the patched HF module (llama_flash_attn_monkey_patch.py:105-115 counterpart)
    from transformers.models.llama import modeling_llama as M
    from transformers import LlamaConfig
    from slime_amd.model.language_model.llama_attention import replace_llama_attn_with_hip_attn, restore_llama_attn, _self_attn_return_arity
    orig = M.LlamaAttention.forward
    try:
        replace_llama_attn_with_hip_attn()
        assert len(M.LlamaAttention.forward(M.LlamaAttention(LlamaConfig(hidden_size=D, num_attention_heads=HQ, num_key_value_heads=HKV,
                   head_dim=128, intermediate_size=256, num_hidden_layers=1, vocab_size=64), layer_idx=0).to(dev), hidden)) == _self_attn_return_arity(M)
        cfg = LlamaConfig(hidden_size=D, num_attention_heads=HQ, num_key_value_heads=HKV, head_dim=128, intermediate_size=256,
                          num_hidden_layers=1, rope_theta=500000.0, vocab_size=64)
        att = M.LlamaAttention(cfg, layer_idx=0)
        att.load_state_dict({k: v for k, v in m.state_dict().items()}, strict=False)
        att.to(dev)
        got2 = att(hidden, attention_mask=None, position_ids=pid)[0]
        assert torch.equal(got2, want)
        # stale-cache guard: an in-place weight edit after the first forward must be seen
        with torch.no_grad():
            att.o_proj.weight.mul_(2.0)
        got3 = att(hidden, attention_mask=None, position_ids=pid)[0]
        assert rel_l2(got3.float().cpu(), (2.0 * want).float().cpu()) < 2e-2 and not torch.equal(got3, got2)
        with pytest.raises(NotImplementedError, match="use_cache=False"):
            att(hidden, attention_mask=None, position_ids=pid, use_cache=True)
    finally:
        restore_llama_attn()
        assert M.LlamaAttention.forward is orig


@pytest.mark.parametrize("padding", ["right", "left", "none"])
def test_patched_attention_inside_hf_llama_model(dev, padding):
    """VERDICT r3 missing #4: ``replace_llama_attn_with_hip_attn`` (counterpart of llama_flash_attn_monkey_patch.py:105-115) inside a
    REAL HF decoder stack of the installed transformers -- a 2-layer random-init ``LlamaModel(inputs_embeds=..., attention_mask=...)``
    with the patch (bf16, on the GPU) against the stock fp32 CPU forward (eager attention, HF's own 4-D causal + padding mask).
    Exercises what broke under transformers >= 4.48: the 2-value return ``LlamaDecoderLayer`` unpacks, ``position_embeddings`` /
    ``position_ids`` keywords, and the model-level mask builder handing the [B, S] key-padding mask through un-expanded."""
    from transformers.models.llama import modeling_llama as M
    from transformers import LlamaConfig
    from slime_amd.model.language_model.llama_attention import replace_llama_attn_with_hip_attn, restore_llama_attn
    torch.manual_seed(3)
    cfg = LlamaConfig(hidden_size=512, num_attention_heads=4, num_key_value_heads=2, head_dim=128, intermediate_size=1024,
                      num_hidden_layers=2, rope_theta=500000.0, vocab_size=128, attn_implementation="eager")
    ref_model = M.LlamaModel(cfg).eval()
    B, S = 3, 83
    emb = torch.randn(B, S, 512) * 0.5
    lens = [83, 61, 17]
    mask = torch.zeros(B, S, dtype=torch.long)
    for b, n in enumerate(lens):
        if padding == "left":
            mask[b, S - n:] = 1
        else:
            mask[b, :n] = 1
    if padding == "none":
        mask = None
    with torch.no_grad():
        ref = ref_model(inputs_embeds=emb, attention_mask=mask, use_cache=False).last_hidden_state
    import copy
    gpu_model = copy.deepcopy(ref_model).to(dev).to(torch.bfloat16)
    orig = M.LlamaAttention.forward
    try:
        replace_llama_attn_with_hip_attn()
        with torch.no_grad():
            got = gpu_model(inputs_embeds=emb.to(dev).to(torch.bfloat16), attention_mask=None if mask is None else mask.to(dev),
                            use_cache=False).last_hidden_state
    finally:
        restore_llama_attn()
    assert M.LlamaAttention.forward is orig
    got = got.float().cpu()
    valid = torch.ones(B, S, dtype=torch.bool) if mask is None else mask.bool()
    err = rel_l2(got[valid], ref[valid])
    print(f"patched LlamaModel ({padding} padding) vs stock fp32 CPU forward: rel-L2 {err:.3e}")
    assert torch.isfinite(got).all() and err < 2.5e-2                    # bf16 stack (HF's own RMSNorm / MLP in bf16 included)


def test_splice_out_of_range_sources(dev):
    """a token id >= vocabulary size raises on the host (nn.Embedding's behaviour, llava_arch.py:373); the kernel never
    dereferences a source row outside its tensor (direct C-ABI callers get a zero row)."""
    from slime_amd import ops
    table = torch.arange(5 * 64, dtype=torch.float32).view(5, 64).to(dev)
    feats = -torch.arange(3 * 64, dtype=torch.float32).view(3, 64).to(dev)
    src = torch.tensor([0, 4, 5, 1000000, -2, -4, -5, -1000, -1], dtype=torch.int64, device=dev)
    out = ops.splice_rows(table, feats, src, torch.float32).cpu()
    assert torch.equal(out[0], table[0].cpu()) and torch.equal(out[1], table[4].cpu())
    assert torch.equal(out[4], feats[0].cpu()) and torch.equal(out[5], feats[2].cpu())
    for r in (2, 3, 6, 7, 8):
        assert float(out[r].abs().max()) == 0.0


def test_token_ranges_cache_is_per_tensor_object(dev):
    """The (start, length) cache of ops.token_ranges must follow the mask OBJECT, not its address: the caching allocator hands
    a freed mask's address to the next one (this bit the left-padding golden test when the cache was keyed on data_ptr)."""
    from slime_amd import ops
    m1 = torch.tensor([[1, 1, 1, 0], [1, 1, 0, 0]], device=dev)
    s1, l1 = ops.token_ranges(m1)
    assert s1.tolist() == [0, 0] and l1.tolist() == [3, 2]
    assert ops.token_ranges(m1)[0] is s1                               # same object, unmodified: cached
    del m1
    m2 = torch.tensor([[0, 1, 1, 1], [0, 0, 1, 1]], device=dev)          # very likely the same address
    s2, l2 = ops.token_ranges(m2)
    assert s2.tolist() == [1, 2] and l2.tolist() == [3, 2]
    m2[0, 0] = 1                                                         # in-place edit: version bump
    s3, l3 = ops.token_ranges(m2)
    assert s3.tolist() == [0, 2] and l3.tolist() == [4, 2]
