"""GPU tests of the reference-shaped plugin API (CLIPVisionTower, build_vision_projector,
build_vision_sampler, encode_images) against the CPU oracle.  Per-stage tolerances are test_gpu_path.py's
(fp16 1.2e-3, bf16 8e-3, rel-L2 vs the fp32 oracle; round 6: they were 3e-3 / 1.5e-2 here); end-of-chain tensors
(pixels rounded to T -> tower -> post_qformer -> MLP) are allowed 1.5x that."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn
from PIL import Image

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = {torch.float16: 1.2e-3, torch.bfloat16: 8e-3}
PIN = "[(336, 672), (672, 336), (672, 672), (1008, 336), (336, 1008)]"


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from slime_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _tiny_encoder(dev, dtype, embed=None, **cfg_over):
    """SlimeVisualEncoder at the tiny fixture geometry (tower 128 wide / 3 layers, adapter 128 -> 256)."""
    from slime_amd import weights as W
    from slime_amd.model.llava_arch import SlimeVisualEncoder, default_slime_config
    from slime_amd.model.multimodal_encoder.clip_encoder import HipCLIPVisionModel
    from slime_amd.image_processor import ClipImageProcessor
    cfg = default_slime_config("synthetic:1", hidden_size=256, mm_hidden_size=128, **cfg_over)
    enc = SlimeVisualEncoder(cfg, embed_tokens=embed)
    vt = enc.get_vision_tower()
    vt.vision_tower = HipCLIPVisionModel(W.TINY)              # tiny stand-in for the 'synthetic:' CLIP-L
    vt.image_processor = ClipImageProcessor()
    vt.is_loaded = True
    tsd = W.make_tower_state_dict(W.TINY, seed=11)
    asd = W.make_adapter_state_dict(W.ADAPTER_TINY, seed=12)
    enc.load_visual_state(tsd, asd)
    enc.to(dev)
    vt.vision_tower.to(dtype)
    return enc, W.strip_tower_prefix(tsd), asd


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_vision_tower_module(dev, dtype):
    from slime_amd import weights as W
    from oracle import slime_oracle as O
    enc, tsd, _ = _tiny_encoder(dev, dtype)
    tower = enc.get_vision_tower()
    assert tower.dtype == dtype and tower.device.type == "cuda" and tower.hidden_size == 128
    px = W.synthetic_pixels(17, seed=5)
    ref = O.tower_forward(tsd, W.TINY, px)
    out = tower(px.to(dev).to(dtype))                          # >= 8 crops: two-stream path (9 + 8)
    assert out.dtype == dtype and out.shape == (17, 576, 128)
    assert rel_l2(out.float().cpu(), ref) < TOL[dtype] * 1.5
    tower.vision_tower.two_streams = False
    one = tower(px.to(dev).to(dtype))
    assert torch.equal(one.cpu(), out.cpu())                   # stream split is bit-invisible
    tower.vision_tower.two_streams = True
    lst = tower([p for p in px[:3].to(dev)])                   # list branch: fp32 crops -> fp32 features
    assert isinstance(lst, list) and lst[0].shape == (1, 576, 128) and lst[0].dtype == torch.float32
    assert rel_l2(torch.cat(lst).cpu(), ref[:3]) < TOL[dtype]
    tower.select_feature = "cls_patch"
    assert tower(px[:2].to(dev)).shape == (2, 577, 128)
    tower.select_feature = "patch"
    assert tower.dummy_feature.shape == (1, 128)
    hs = tower.vision_tower(px[:1].to(dev), output_hidden_states=True).hidden_states
    assert len(hs) == W.TINY.num_hidden_layers + 1
    assert torch.equal(tower.feature_select(SimpleNamespace(hidden_states=hs)).cpu(), tower(px[:1].to(dev)).cpu())
    # ONE pass produces all L+1 states (slime_vit_forward_states): each against the oracle's hidden_states list, and the
    # last one equals the separate full-depth run bit for bit
    ref_hs = O.clip_hidden_states(tsd, W.TINY, px[:1])
    assert all(h.shape == (1, 577, 128) and h.dtype == torch.float32 for h in hs)
    for i, h in enumerate(hs):
        assert rel_l2(h.cpu(), ref_hs[i]) < TOL[dtype] * 1.5, i
    last = tower.vision_tower(px[:1].to(dev)).last_hidden_state
    assert torch.equal(last, hs[-1])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_projector_and_sampler_modules(dev, dtype):
    from slime_amd import weights as W
    from oracle import slime_oracle as O
    enc, tsd, asd = _tiny_encoder(dev, dtype)
    proj_sd, post_sd = W.sub_state(asd, "mm_projector."), W.sub_state(asd, "sampler.post_qformer.")
    feats = O.tower_forward(tsd, W.TINY, W.synthetic_pixels(3, seed=22))
    x = feats.to(dev).to(dtype)
    m = enc.get_model()
    g = m.mm_projector(x[0])                                   # [576, D] -> gated path, squeezed back
    assert g.shape == (576, 256) and g.dtype == dtype
    assert rel_l2(g.float().cpu(), O.gated_block_forward(proj_sd, x[0].float().cpu(), 1)) < TOL[dtype]
    gb = m.mm_projector(x[:2])                                 # [N,576,D] batched gated path
    assert rel_l2(gb.float().cpu(), O.gated_block_forward(proj_sd, x[:2].float().cpu(), 1)) < TOL[dtype]
    comp = m.sampler.post_qformer(x[1:])
    assert comp.shape == (2, 144, 128)
    ref_comp = O.resampler_forward(post_sd, x[1:].float().cpu(), 1)
    assert rel_l2(comp.float().cpu(), ref_comp) < TOL[dtype]
    loc = m.mm_projector(comp)                                 # early-return branch: plain MLP
    assert loc.shape == (2, 144, 256)
    assert rel_l2(loc.float().cpu(), O.mlp_projector(proj_sd, comp.float().cpu())) < TOL[dtype]
    m.mm_projector.learnable_gated = 1
    e1 = m.mm_projector(x[0])
    assert rel_l2(e1.float().cpu(), O.gated_block_forward(proj_sd, x[0].float().cpu(), 1, learnable_gated=1)) < TOL[dtype]
    m.mm_projector.learnable_gated = -1


def test_mlp2x_and_linear_projectors(dev):
    from slime_amd.model.multimodal_projector.builder import build_vision_projector
    import torch.nn.functional as F
    cfg = SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=128, hidden_size=256)
    p = build_vision_projector(cfg).to(dev)
    assert set(p.state_dict()) == {"0.weight", "0.bias", "2.weight", "2.bias"}
    x = torch.randn(5, 144, 128, device=dev)
    ref = F.linear(F.gelu(F.linear(x, p[0].weight, p[0].bias)), p[2].weight, p[2].bias)
    assert rel_l2(p(x).cpu(), ref.cpu()) < 1.5e-2
    lin = build_vision_projector(SimpleNamespace(mm_projector_type="linear", mm_hidden_size=128, hidden_size=256)).to(dev)
    assert rel_l2(lin(x).cpu(), F.linear(x, lin.weight, lin.bias).cpu()) < 1.5e-2


def test_router_kernels(dev):
    from slime_amd import ops
    from oracle import slime_oracle as O
    g = torch.Generator().manual_seed(3)
    for T, L, H, masked in ((288, 9, 256, True), (1008, 300, 4096, True), (576, 40, 256, False), (37, 5, 128, True)):
        img = torch.randn(T, H, generator=g)
        txt = torch.randn(L, H, generator=g)
        mask = (torch.rand(L, generator=g) > 0.3) if masked else None
        ref = O.router_cosine_scores(img, txt, mask)
        sc = ops.router_scores(img.to(dev), txt.to(dev), None if mask is None else mask.to(dev))
        assert rel_l2(sc.cpu(), ref) < 2e-5
        for topp, temp in ((0.95, 1.0), (0.5, 0.3), (1.0, 1.0), (0.001, 1.0)):
            keep, cnt, probs = ops.router_select(sc, topp, temp, want_probs=True)
            n = int(cnt.item())
            # selection logic must be EXACT given the device's own probabilities
            p = probs.cpu()
            sp, si = torch.sort(p, descending=True, stable=True)
            k = int((torch.cumsum(sp, 0) <= topp).sum())
            exp = si[: k + 1] if k < T else torch.arange(T)
            assert n == exp.numel()
            assert torch.equal(keep[:n].cpu().long(), exp.sort()[0])
            assert rel_l2(p, torch.softmax(sc.cpu() / temp, 0)) < 1e-5
        # end to end against the oracle's selection on its own scores (non-borderline data)
        kept = ops.router_topp(img.to(dev), txt.to(dev), None if mask is None else mask.to(dev), 0.95, 1.0)
        ref_keep = O.router_select(ref, 0.95, 1.0)
        assert abs(kept.numel() - ref_keep.numel()) <= 1


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("sizes", [[(672, 672)], [(336, 336), (672, 672), (1344, 1344)]])
def test_encode_images_vs_oracle(dev, dtype, sizes):
    """Full sampler branch of encode_images (tower -> adapter -> merge -> router -> concat) on a batch
    with ragged crop counts (2, 4, 6 local crops), every image against the oracle."""
    from slime_amd import weights as W
    from slime_amd.constants import IMAGE_TOKEN_INDEX
    from oracle import slime_oracle as O
    torch.manual_seed(0)
    embed = nn.Embedding(2000, 256).to(dev)
    enc, tsd, asd = _tiny_encoder(dev, dtype, embed=embed)
    counts = [1 + O.anyres_grid_shape(s)[0] * O.anyres_grid_shape(s)[1] for s in sizes]
    px = [W.synthetic_pixels(c, seed=50 + i) for i, c in enumerate(counts)]
    images = torch.cat(px, 0).to(dev).to(dtype)
    L = 12
    ids = torch.randint(3, 1900, (len(sizes), L), device=dev)
    ids[:, 4] = IMAGE_TOKEN_INDEX
    am = torch.ones_like(ids)
    am[:, -2:] = 0
    feats, ss = enc.encode_images(images, input_ids=ids, split_sizes=counts, attention_mask=am, image_sizes=sizes)
    assert ss == counts and len(feats) == len(sizes)
    text, tmask = enc.get_pure_text_embedding(ids, am)
    sep = embed(torch.tensor(enc.config.seperator, device=dev))
    pairs = enc.encode_visual(images, counts, sizes, merge="spatial")           # device tokens before the router
    for i, s in enumerate(sizes):
        ref = O.encode_image(tsd, asd, W.TINY, W.ADAPTER_TINY, px[i], s, text[i].float().cpu(), tmask[i].cpu(),
                             separator=sep.float().cpu())
        out = feats[i]
        assert out.dim() == 3 and out.shape[0] == 1 and out.dtype == dtype
        out = out[0].float().cpu()
        assert rel_l2(out[:576], ref["global"]) < TOL[dtype] * 1.5
        assert torch.allclose(out[576], sep.float().cpu().to(dtype).float())
        # router: the oracle's rule applied to the DEVICE's merged tokens must select the same set, up to
        # near-ties at the top-p cut (the scores agree to ~1e-6, the selection is discontinuous)
        merged_dev = pairs[i][1].cpu()
        assert rel_l2(merged_dev, ref["merged"]) < TOL[dtype] * 1.5
        keep_ref = O.router_select(O.router_cosine_scores(merged_dev, text[i].float().cpu(), tmask[i].cpu()), 0.95, 1.0)
        rows = out[577:]
        assert abs(rows.shape[0] - keep_ref.numel()) <= 2
        cand = merged_dev.to(dtype).float()
        # every emitted row is one of the merged rows, in ascending order
        idx = torch.cdist(rows, cand).argmin(1)                   # cdist is only used to identify the rows
        assert torch.equal(rows, cand[idx]) and bool((idx[1:] > idx[:-1]).all())
        sym = set(idx.tolist()) ^ set(keep_ref.tolist())
        assert len(sym) <= 4, sorted(sym)
    # router-free form used by bench.py
    for i, s in enumerate(sizes):
        ref = O.encode_image(tsd, asd, W.TINY, W.ADAPTER_TINY, px[i], s)
        assert rel_l2(pairs[i][1].cpu(), ref["merged"]) < TOL[dtype] * 1.5


def test_gpu_slicer_matches_pil_path(dev):
    from slime_amd import mm_utils as M
    from slime_amd.image_processor import ClipImageProcessor
    proc = ClipImageProcessor()
    for i, (w, h) in enumerate([(672, 672), (500, 900), (300, 200), (1344, 1344)]):
        arr = np.random.default_rng(7 + i).integers(0, 256, (h, w, 3), dtype=np.uint8)
        img = Image.fromarray(arr, "RGB")
        ref = M.process_anyres_image(img, proc, PIN)
        out = M.process_anyres_image_gpu(img, proc, PIN, dev)
        assert out.shape == ref.shape
        assert torch.equal(out.cpu(), ref), (w, h)            # same fp32 arithmetic order as the HF processor


def test_encode_images_flags_and_mask(dev):
    """images_mask (collator-padded crops, train.py:903-926 / llava_arch.py:228-231), 'flat' merge,
    use_global_only / use_local_only, and the branches without a sampler."""
    from slime_amd import weights as W
    from slime_amd.constants import IMAGE_TOKEN_INDEX
    from oracle import slime_oracle as O
    dtype = torch.bfloat16
    torch.manual_seed(1)
    embed = nn.Embedding(2000, 256).to(dev)
    ids = torch.randint(3, 1900, (2, 10), device=dev)
    ids[:, 2] = IMAGE_TOKEN_INDEX
    am = torch.ones_like(ids)
    # two images padded to 1 + 4 crops; image 0 really has 2 local crops, image 1 has 4
    px = [W.synthetic_pixels(5, seed=70 + i) for i in range(2)]
    images = torch.cat(px, 0).to(dev).to(dtype)
    mask = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]], device=dev)
    sizes = [(336, 336), (672, 672)]
    enc, tsd, asd = _tiny_encoder(dev, dtype, embed=embed)
    feats, _ = enc.encode_images(images, input_ids=ids, split_sizes=[5, 5], attention_mask=am, images_mask=mask,
                                 image_sizes=sizes)
    text, tmask = enc.get_pure_text_embedding(ids, am)
    for i, n_real in enumerate((2, 4)):
        ref = O.encode_image(tsd, asd, W.TINY, W.ADAPTER_TINY, px[i][: 1 + n_real], sizes[i], text[i].float().cpu(),
                             tmask[i].cpu())
        out = feats[i][0].float().cpu()
        assert rel_l2(out[:576], ref["global"]) < TOL[dtype] * 1.5
        assert abs((out.shape[0] - 577) - ref["router_keep"].numel()) <= 2
    # flat merge keeps crop-major order
    enc_f, _, _ = _tiny_encoder(dev, dtype, embed=embed, mm_patch_merge_type="flat")
    pairs = enc_f.encode_visual(images[:5], [5], [(672, 672)], merge="flat")
    ref = O.encode_image(tsd, asd, W.TINY, W.ADAPTER_TINY, px[0], (672, 672), merge="flat")
    assert rel_l2(pairs[0][1].cpu(), ref["merged"]) < TOL[dtype] * 1.5
    # use_global_only / use_local_only
    enc_g, _, _ = _tiny_encoder(dev, dtype, embed=embed, use_global_only=True)
    fg, _ = enc_g.encode_images(images[:5], input_ids=ids[:1], split_sizes=[5], attention_mask=am[:1], image_sizes=[(672, 672)])
    assert fg[0].shape == (1, 576, 256)
    enc_l, _, _ = _tiny_encoder(dev, dtype, embed=embed, use_local_only=True)
    fl, _ = enc_l.encode_images(images[:5], input_ids=ids[:1], split_sizes=[5], attention_mask=am[:1], image_sizes=[(672, 672)])
    assert fl[0].shape[0] == 1 and fl[0].shape[2] == 256 and 0 < fl[0].shape[1] <= 576
    # no sampler: plain batch through the gated projector, and the split_sizes list branch
    enc_n, _, _ = _tiny_encoder(dev, dtype, embed=embed, mm_resampler_type=None)
    assert not enc_n.get_model().has_sampler
    plain, _ = enc_n.encode_images(images[:3])
    ref_plain = O.gated_block_forward(W.sub_state(asd, "mm_projector."), O.tower_forward(tsd, W.TINY, px[0][:3]), 1)
    assert plain.shape == (3, 576, 256) and rel_l2(plain.float().cpu(), ref_plain) < TOL[dtype] * 1.5
    lst, ss = enc_n.encode_images(images[:5], split_sizes=[2, 3])
    assert ss == [2, 3] and [t.shape[0] for t in lst] == [2, 3]
    assert rel_l2(torch.cat(lst).float().cpu()[:3], ref_plain) < TOL[dtype] * 1.5


def test_tower_list_edge_cases(dev):
    from slime_amd import weights as W
    enc, _, _ = _tiny_encoder(dev, torch.bfloat16)
    tower = enc.get_vision_tower()
    assert tower([]) == []
    one = tower([W.synthetic_pixels(1, seed=1)[0].to(dev)])
    assert len(one) == 1 and one[0].shape == (1, 576, 128)
    with pytest.raises(ValueError, match="doesn't match model"):
        tower(torch.zeros(1, 3, 224, 224, device=dev))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("gated", [-1, 0, 1])
def test_fused_adapter_equals_module_sequence(dev, dtype, gated):
    """slime_adapter_forward (stacked MLP, batched gate mix + merge, one C call) against the per-module launch
    sequence on the same 16-bit tower features: same kernels on the same rows -> equal to rounding noise
    (bit-equal unless the GEMM tile choice differs with M), plus the untouched padding rows of a wider buffer."""
    from slime_amd import ops, weights as W
    enc, _, _ = _tiny_encoder(dev, dtype, mm_learnable_gated=gated)
    model = enc.get_model()
    model.mm_projector.learnable_gated = gated
    B, n_local, nw, nh = 3, 6, 2, 3
    feats = torch.randn(B * (1 + n_local), 576, 128, device=dev).to(dtype)
    f32 = feats.float()
    g_idx = torch.arange(0, B * (1 + n_local), 1 + n_local, device=dev)
    l_idx = torch.tensor([i for i in range(B * (1 + n_local)) if i % (1 + n_local)], device=dev)
    pg = model.mm_projector.packed(dtype)                      # explicit operand dtype for both sides
    post = model.sampler.post_qformer.packed(576, dtype)
    glob = ops.gated_forward(pg, f32.index_select(0, g_idx), gated)
    comp = ops.resampler_forward(post, f32.index_select(0, l_idx))
    loc = ops.mlp_forward(pg.mlp, comp.reshape(-1, 128)).reshape(B * n_local, -1, 256)
    g = model.sampler.grid_size
    ref = torch.empty((B, 576 + n_local * g * g, 256), dtype=torch.float32, device=dev)
    for i in range(B):
        ref[i, :576] = glob[i]
        ops.merge_rows(loc[i * n_local:(i + 1) * n_local].contiguous(), ref[i], 576, nw, nh, g, True)
    out = ops.adapter_forward(pg, post, feats, B, n_local, nw, nh, True, gated, torch.float32)
    assert out.shape == ref.shape
    assert rel_l2(out.cpu(), ref.cpu()) < 1e-6, float((out - ref).abs().max())
    # 16-bit output into a wider token buffer: rows beyond the image's tokens stay untouched
    wide = torch.full((B, ref.shape[1] + 5, 256), 3.0, dtype=dtype, device=dev)
    ops.adapter_forward(pg, post, feats, B, n_local, nw, nh, True, gated, out=wide)
    assert torch.equal(wide[:, :ref.shape[1]], ref.to(dtype)) or rel_l2(wide[:, :ref.shape[1]].float().cpu(), ref.cpu()) < 2e-3
    assert bool((wide[:, ref.shape[1]:] == 3.0).all())
    # flat order and a global-only batch
    flat = ops.adapter_forward(pg, post, feats, B, n_local, n_local, 1, False, gated, torch.float32)
    assert rel_l2(flat[:, 576:].cpu(), loc.reshape(B, n_local * g * g, 256).cpu()) < 1e-6
    only_g = ops.adapter_forward(pg, None, feats[:4], 4, 0, 1, 1, False, gated, torch.float32)
    ref_g = ops.gated_forward(pg, f32[:4].contiguous(), gated)
    assert only_g.shape == (4, 576, 256) and rel_l2(only_g.cpu(), ref_g.cpu()) < 1e-6


def test_fused_adapter_full_dims_and_errors(dev):
    """SliME-8B adapter dims (1024 -> 4096, 8 x 128 heads), 2 images x (1+4): fused == module sequence; bad grids raise."""
    from slime_amd import ops, weights as W
    from slime_amd._lib import SlimeHipError
    dtype = torch.bfloat16
    asd = W.make_adapter_state_dict(W.ADAPTER_8B, seed=5)
    pg = ops.pack_gated(W.sub_state(asd, "mm_projector."), W.ADAPTER_8B, dtype, dev)
    post = ops.pack_resampler(W.sub_state(asd, "sampler.post_qformer."), 1024, 8, 576, dtype, dev, W.ADAPTER_8B.ln_eps)
    B, n = 2, 4
    feats = torch.randn(B * (1 + n), 576, 1024, device=dev).to(dtype)
    f32 = feats.float()
    out = ops.adapter_forward(pg, post, feats, B, n, 2, 2, True, -1, torch.float32)
    glob = ops.gated_forward(pg, f32[0::5].contiguous())
    loc_idx = [i for i in range(B * 5) if i % 5]
    comp = ops.resampler_forward(post, f32[loc_idx].contiguous())
    loc = ops.mlp_forward(pg.mlp, comp.reshape(-1, 1024)).reshape(B * n, 144, 4096)
    assert rel_l2(out[:, :576].cpu(), glob.cpu()) < 1e-6
    ref_l = torch.empty((B, n * 144, 4096), dtype=torch.float32, device=dev)
    for i in range(B):
        ops.merge_rows(loc[i * n:(i + 1) * n].contiguous(), ref_l[i], 0, 2, 2, 12, True)
    assert rel_l2(out[:, 576:].cpu(), ref_l.cpu()) < 1e-6
    with pytest.raises(SlimeHipError, match="grid"):
        ops.adapter_forward(pg, post, feats, B, n, 3, 2, True, -1, torch.float32)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fused_adapter_bench_shape_vs_oracle(dev, dtype):
    """the bench step's own adapter launch -- SliME-8B dims, 8 images x (1 + 4) crops, 13 824 stacked MLP rows
    (gemm_db_kernel<.., 3, 1, 8> / <.., 2, 0, 8>), 2 x 2 merge -- against the fp32 CPU oracle as a whole: gated global rows and
    merged local rows of every image.  Input = 40 crops of tower-like features (unit-variance tokens with a per-crop offset);
    stated bounds: fp16 1.2e-3, bf16 8e-3 (measured 5-7e-4 / 4-6e-3)."""
    from slime_amd import ops, weights as W
    from oracle import slime_oracle as O
    A = W.ADAPTER_8B
    asd = W.make_adapter_state_dict(A, seed=4321)
    proj_sd, post_sd = W.sub_state(asd, "mm_projector."), W.sub_state(asd, "sampler.post_qformer.")
    pg = ops.pack_gated(proj_sd, A, dtype, dev)
    post = ops.pack_resampler(post_sd, 1024, 8, 576, dtype, dev, A.ln_eps)
    B, n = 8, 4
    g = torch.Generator().manual_seed(31)
    feats32 = torch.randn(B * (1 + n), 576, 1024, generator=g) + 0.3 * torch.randn(B * (1 + n), 1, 1024, generator=g)
    feats = feats32.to(dtype)                                            # what the tower hands over in the bench dtype
    out = ops.adapter_forward(pg, post, feats.to(dev), B, n, 2, 2, True, -1, torch.float32).cpu()
    assert out.shape == (B, 576 + n * 144, 4096)
    bound = {torch.float16: 1.2e-3, torch.bfloat16: 8e-3}[dtype]
    ref_in = feats.float()                                               # the oracle sees the same (rounded) features
    worst = 0.0
    for i in range(B):
        f = ref_in[i * (1 + n):(i + 1) * (1 + n)]
        glob = O.gated_block_forward(proj_sd, f[0], A.num_heads)
        comp = O.resampler_forward(post_sd, f[1:], A.num_heads, A.ln_eps)
        merged = O.spatial_merge(O.mlp_projector(proj_sd, comp), 2, 2, 12)
        eg, el = rel_l2(out[i, :576], glob), rel_l2(out[i, 576:], merged)
        worst = max(worst, eg, el)
        assert eg < bound and el < bound, (i, dtype, eg, el)
    print(f"fused adapter, bench shape, {dtype}: worst rel-L2 over 8 images {worst:.3e}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_cfg3_layout_17_crops_vs_oracle(dev, dtype):
    """BASELINE config 3's layout (1 global + 16 local crops on an explicit 4 x 4 grid -- synthetic: the reference slicer
    never emits more than 7 crops), 2 images, tiny geometry: tower + fused adapter against the oracle's stages."""
    from slime_amd import ops, weights as W
    from oracle import slime_oracle as O
    enc, tsd, asd = _tiny_encoder(dev, dtype)
    model = enc.get_model()
    B, n_local, nw, nh = 2, 16, 4, 4
    px = W.synthetic_pixels(B * (1 + n_local), seed=77)
    feats = enc.get_vision_tower()(px.to(dev).to(dtype), out_dtype=dtype)                    # [34, 576, 128]
    tokens = ops.adapter_forward(model.mm_projector.packed(dtype), model.sampler.post_qformer.packed(576, dtype), feats,
                                 B, n_local, nw, nh, True, -1, torch.float32).cpu()
    heads = W.ADAPTER_TINY.num_heads
    for i in range(B):
        f = O.tower_forward(tsd, W.TINY, px[i * 17:(i + 1) * 17])
        glob = O.gated_block_forward(W.sub_state(asd, "mm_projector."), f[:1], heads)[0]
        comp = O.resampler_forward(W.sub_state(asd, "sampler.post_qformer."), f[1:], heads)
        loc = O.mlp_projector(W.sub_state(asd, "mm_projector."), comp)
        merged = O.spatial_merge(loc, nw, nh, 12)
        assert rel_l2(tokens[i, :576], glob) < TOL[dtype] * 1.5
        assert rel_l2(tokens[i, 576:], merged) < TOL[dtype] * 1.5


def test_stacked_local_crops_of_576_take_the_mlp(dev):
    """the ragged path projects the batch-stacked compressed local crops [sum n_i, 144, D] in one call; with
    exactly 576 local crops in the batch GatedBlock.forward's shape test (dim0 == 576) must not be what decides -- the MLP
    is called directly, like the reference's per-image calls always end up doing."""
    from slime_amd.model.llava_arch import _project_local
    enc, _, asd = _tiny_encoder(dev, torch.bfloat16)
    proj = enc.get_model().mm_projector
    comp = torch.randn(576, 144, 128, generator=torch.Generator().manual_seed(5)).to(dev)
    out = _project_local(proj, comp)
    assert out.shape == (576, 144, 256)
    part = proj.projection(comp[:5], out_dtype=torch.float32)
    assert torch.equal(out[:5], part)
    with pytest.raises(ValueError, match="GatedBlock expects"):
        proj(comp)                                             # the shape test itself is the reference's (builder.py:180)


def test_router_matches_reference_goldens(dev):
    """the kept count and the first kept rows of the REFERENCE's own router run (tiny_stages.npz:
    n*_router_rows / n*_router_first / n*_router_scores) are reproduced exactly by the HIP router when it is fed the same
    merged tokens (the oracle's, which equal the reference's to 2e-5): 273/288, 545/576, 817/864 here."""
    import os
    from conftest import GOLDEN
    from slime_amd import ops, weights as W
    from oracle import slime_oracle as O
    g = np.load(os.path.join(GOLDEN, "tiny_stages.npz"))
    tsd = W.strip_tower_prefix(W.make_tower_state_dict(W.TINY, seed=11))
    asd = W.make_adapter_state_dict(W.ADAPTER_TINY, seed=12)
    for n_local, size in ((2, (336, 336)), (4, (672, 672)), (6, (1344, 1344))):
        k = f"n{n_local}_"
        px = W.synthetic_pixels(1 + n_local, seed=20 + n_local)
        merged = O.encode_image(tsd, asd, W.TINY, W.ADAPTER_TINY, px, size)["merged"]
        text = torch.randn(9, W.ADAPTER_TINY.hidden_size, generator=torch.Generator().manual_seed(300 + n_local))
        mask = torch.tensor([1, 1, 1, 1, 1, 1, 0, 0, 1], dtype=torch.bool)
        sc = ops.router_scores(merged.to(dev), text.to(dev), mask.to(dev))
        assert rel_l2(sc.cpu(), g[k + "router_scores"]) < 2e-5
        keep = ops.router_topp(merged.to(dev), text.to(dev), mask.to(dev), 0.95, 1.0).cpu()
        assert keep.numel() == int(g[k + "router_rows"][0])
        assert rel_l2(merged[keep[:4]], g[k + "router_first"]) < 2e-5
        assert torch.equal(keep, O.router_select(torch.from_numpy(g[k + "router_scores"]), 0.95, 1.0))


def test_router_exact_on_separated_scores_and_batched_equals_single(dev):
    """Selection is discontinuous in the scores, so exactness is demanded where it is well defined: scores separated by
    far more than fp32 rounding -> the kept SET equals the oracle's for every (top-p, temperature); and the batched
    launch (fused strided layout and ragged concatenation, with an empty image) equals per-image calls bit for bit."""
    from slime_amd import ops
    from oracle import slime_oracle as O
    gen = torch.Generator().manual_seed(17)
    for T in (37, 288, 1008, 4096):
        sc = (torch.linspace(-4, 4, T)[torch.randperm(T, generator=gen)]).contiguous()
        for topp, temp in ((0.95, 1.0), (0.5, 0.3), (0.999, 2.0), (1e-4, 1.0)):     # (top-p = 1.0 is a tie with the total: not a separated case)
            keep, cnt = ops.router_select(sc.to(dev), topp, temp)
            got = keep[: int(cnt.item())].cpu().long()
            assert torch.equal(got, O.router_select(sc, topp, temp)), (T, topp, temp)
    B, P, T, H, L = 3, 5, 288, 256, 9
    tokens = torch.randn(B, P + T, H, generator=gen).to(dev)
    text = torch.randn(B, L, H, generator=gen).to(dev)
    mask = (torch.rand(B, L, generator=gen) > 0.3).to(dev)
    single = [ops.router_topp(tokens[i, P:].contiguous(), text[i], mask[i], 0.95, 1.0) for i in range(B)]
    fused = ops.router_topp_batched(tokens.view(B * (P + T), H), [i * (P + T) + P for i in range(B)], [T] * B, text, mask, 0.95, 1.0)
    for a, b in zip(single, fused):
        assert torch.equal(a, b)
    lens = [T, 0, 144]
    cat = torch.cat([tokens[0, P:], tokens[2, P:P + 144]], 0).contiguous()
    ragged = ops.router_topp_batched(cat, [0, T, T], lens, text, None, 0.9, 0.7)
    assert ragged[1].numel() == 0
    assert torch.equal(ragged[0], ops.router_topp(tokens[0, P:].contiguous(), text[0], None, 0.9, 0.7))
    assert torch.equal(ragged[2], ops.router_topp(tokens[2, P:P + 144].contiguous(), text[2], None, 0.9, 0.7))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_config1_plugin_api_one_crop_full_size(dev, dtype):
    """BASELINE config 1 through the reference-shaped plugin API at CLIP-ViT-L/14-336 / SliME-8B dims: one 336 x 336 image,
    ``image_aspect_ratio='pad'`` (mm_utils.py:234-238 -> ONE crop), ``CLIPVisionTower.forward`` tensor branch
    (clip_encoder.py:46-58) and ``encode_images`` without split sizes (llava_arch.py:261-267: GatedBlock on the global
    view), against the fp32 oracle on the same pixels.  fp16 meets north_star's 1e-3 at the projector output."""
    from slime_amd import weights as W, mm_utils as M
    from slime_amd.model.llava_arch import SlimeVisualEncoder, default_slime_config
    from oracle import slime_oracle as O
    cfg = default_slime_config("synthetic:1234", image_aspect_ratio="pad")
    tower_sd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
    adapter_sd = W.make_adapter_state_dict(W.ADAPTER_8B, seed=4321)
    enc = SlimeVisualEncoder(cfg)
    enc.load_visual_state(tower_sd, adapter_sd)
    enc.to(dev)
    enc.get_vision_tower().vision_tower.to(dtype)
    tower = enc.get_vision_tower()
    img = Image.fromarray(np.random.default_rng(3).integers(0, 256, (336, 336, 3), dtype=np.uint8), "RGB")
    px = M.process_images([img], tower.image_processor, cfg)
    assert px.shape == (1, 3, 336, 336)                        # 'pad' yields one crop ('anyres' would yield 3: SURVEY a-1)
    tsd = W.strip_tower_prefix(tower_sd)
    ref_t = O.tower_forward(tsd, W.CLIP_L_336, px)
    ref_g = O.gated_block_forward(W.sub_state(adapter_sd, "mm_projector."), ref_t[0], W.ADAPTER_8B.num_heads)
    feats = tower(px.to(dev).to(dtype))
    assert feats.shape == (1, 576, 1024) and feats.dtype == dtype
    assert rel_l2(feats.float().cpu(), ref_t) < TOL[dtype]
    out, ss = enc.encode_images(px.to(dev).to(dtype))
    assert ss is None and out.shape == (1, 576, 4096) and out.dtype == dtype
    err = rel_l2(out[0].float().cpu(), ref_g)
    print(f"config 1 through the plugin API, {dtype}: tower {rel_l2(feats.float().cpu(), ref_t):.3e} projector {err:.3e}")
    assert err <= {torch.float16: 1e-3, torch.bfloat16: 1.2e-2}[dtype], err


@pytest.mark.parametrize("ptype", ["linear", "mlp2x_gelu"])
def test_encode_images_plain_branch_with_other_projector_types(dev, ptype):
    """llava_arch.py:261-267 with ``mm_projector_type`` 'linear' / 'mlp2x_gelu' and no sampler: encode_images hands every projector
    type its fp32 tower features plus the images' 16-bit operand type (round 6: the 'linear' projector did not take the keywords
    the branch passes).  Against torch on the tower's own features."""
    import torch.nn.functional as F
    from slime_amd import weights as W
    from slime_amd.model.llava_arch import SlimeVisualEncoder, default_slime_config
    from slime_amd.model.multimodal_encoder.clip_encoder import HipCLIPVisionModel
    from slime_amd.image_processor import ClipImageProcessor
    dtype = torch.float16
    torch.manual_seed(3)
    enc = SlimeVisualEncoder(default_slime_config("synthetic:1", hidden_size=256, mm_hidden_size=128, mm_projector_type=ptype, mm_resampler_type=None))
    vt = enc.get_vision_tower()
    vt.vision_tower, vt.image_processor, vt.is_loaded = HipCLIPVisionModel(W.TINY), ClipImageProcessor(), True
    enc.load_visual_state(W.make_tower_state_dict(W.TINY, seed=11), None)          # the projector keeps its nn.Linear initialisation
    enc.to(dev)
    vt.vision_tower.to(dtype)
    px = W.synthetic_pixels(2, seed=8).to(dev).to(dtype)
    out, ss = enc.encode_images(px)
    assert ss is None and out.shape == (2, 576, 256) and out.dtype == dtype
    feats = enc.get_vision_tower()(px, out_dtype=torch.float32)
    p = enc.get_model().mm_projector
    if ptype == "linear":
        ref = F.linear(feats, p.weight.float(), p.bias.float())
    else:
        ref = F.linear(F.gelu(F.linear(feats, p[0].weight.float(), p[0].bias.float())), p[2].weight.float(), p[2].bias.float())
    assert rel_l2(out.float().cpu(), ref.cpu()) < 2e-3
