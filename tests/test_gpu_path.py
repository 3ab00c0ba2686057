"""GPU parity of the composed hot path (tower, resampler, projector, gated block) against the CPU oracle
and the committed golden vectors (produced by the reference itself, oracle/make_golden.py).

Stated tolerances (rel-L2 vs the fp32 oracle; MFMA operands 16-bit, fp32 accumulate / residual
stream / LayerNorm / softmax statistics):
    fp16 operands : tower 1.2e-3, adapter stages 1.2e-3 (measured 5.6e-4; the reference's own fp16-vs-fp32 drift: 1.5e-3)
    bf16 operands : tower 8e-3, adapter stages 8e-3     (measured 4.4e-3; the reference's own bf16-vs-fp32 drift: 1.2e-2)
    projector outputs at the end of the chain (tower + adapter): fp16 1e-3 (north_star's target), bf16 1.2e-2 (measured 6.3e-3)
Round 4: the bounds sit at ~2x what the kernels deliver instead of 3-5x, so a regression that doubles an error
fails here.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 1.2e-3, torch.bfloat16: 8e-3}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from slime_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def tiny_weights():
    from slime_amd import weights as W
    return (W.strip_tower_prefix(W.make_tower_state_dict(W.TINY, seed=11)),
            W.make_adapter_state_dict(W.ADAPTER_TINY, seed=12))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_tiny_tower_vs_oracle_and_golden(dev, tiny_weights, dtype):
    from slime_amd import ops, weights as W
    from oracle import slime_oracle as O
    tsd, _ = tiny_weights
    px = W.synthetic_pixels(3, seed=22)
    pt = ops.pack_tower(tsd, W.TINY, dtype, dev)
    out, hidden = ops.tower_forward(pt, px.to(dev), out_dtype=torch.float32, want_hidden=True)
    ref = O.tower_forward(tsd, W.TINY, px)
    g = np.load(os.path.join(GOLDEN, "tiny_stages.npz"))
    assert out.shape == (3, 576, 128)
    assert rel_l2(out.cpu(), ref) < TOL[dtype]
    assert rel_l2(out.cpu(), g["n2_tower"]) < TOL[dtype]                 # the reference's own output
    assert torch.equal(hidden[:, 1:].cpu(), out.cpu())                   # 'patch' = drop the class token
    # select_layer 0 / 1 (embeddings / one layer) pin the front end separately
    for sel, key in ((0, "n2_hidden0"), (1, "n2_hidden1")):
        p2 = ops.pack_tower(tsd, W.TINY, dtype, dev, select_layer=sel)
        o2 = ops.tower_forward(p2, px.to(dev), out_dtype=torch.float32, keep_cls=True)
        assert rel_l2(o2.cpu()[:, ::9, ::4], g[key]) < TOL[dtype], key
    # output dtype follows the input dtype (clip_encoder.py:52,56)
    o16 = ops.tower_forward(pt, px.to(dev).to(dtype))
    assert o16.dtype == dtype and rel_l2(o16.float().cpu(), ref) < TOL[dtype] * 1.5


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_tiny_adapter_vs_oracle(dev, tiny_weights, dtype):
    from slime_amd import ops, weights as W
    from oracle import slime_oracle as O
    tsd, asd = tiny_weights
    A = W.ADAPTER_TINY
    feats = O.tower_forward(tsd, W.TINY, W.synthetic_pixels(3, seed=22))     # fp32 oracle features
    proj_sd, post_sd = W.sub_state(asd, "mm_projector."), W.sub_state(asd, "sampler.post_qformer.")
    x = feats.to(dev)
    pr = ops.pack_resampler(post_sd, A.mm_hidden_size, A.num_heads, 576, dtype, dev, A.ln_eps)
    comp = ops.resampler_forward(pr, x[1:])
    ref_comp = O.resampler_forward(post_sd, feats[1:], A.num_heads, A.ln_eps)
    assert rel_l2(comp.cpu(), ref_comp) < TOL[dtype]
    pg = ops.pack_gated(proj_sd, A, dtype, dev)
    loc = ops.mlp_forward(pg.mlp, comp.reshape(-1, A.mm_hidden_size)).view(2, 144, -1)
    assert rel_l2(loc.cpu(), O.mlp_projector(proj_sd, ref_comp)) < TOL[dtype]
    glob = ops.gated_forward(pg, x[:1])[0]
    assert rel_l2(glob.cpu(), O.gated_block_forward(proj_sd, feats[0], A.num_heads)) < TOL[dtype]
    for lg in (0, 1):
        e = ops.gated_forward(pg, x[:1], learnable_gated=lg)[0]
        assert rel_l2(e.cpu(), O.gated_block_forward(proj_sd, feats[0], A.num_heads, learnable_gated=lg)) < TOL[dtype]
    g = np.load(os.path.join(GOLDEN, "tiny_stages.npz"))
    assert rel_l2(glob.cpu(), g["n2_global"]) < TOL[dtype] * 1.5          # golden used the reference's tower output
    assert rel_l2(comp.cpu(), g["n2_compressed"]) < TOL[dtype] * 1.5


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_full_vit_l_vs_golden(dev, dtype):
    """CLIP-ViT-L/14-336 geometry, 1+4 crops: sub-sampled reference outputs + per-crop statistics."""
    from slime_amd import ops, weights as W
    g = np.load(os.path.join(GOLDEN, "full_stages.npz"))
    tsd = W.strip_tower_prefix(W.make_tower_state_dict(W.CLIP_L_336, seed=1234))
    asd = W.make_adapter_state_dict(W.ADAPTER_8B, seed=4321)
    px = W.synthetic_pixels(5, seed=44)
    pt = ops.pack_tower(tsd, W.CLIP_L_336, dtype, dev)
    feats = ops.tower_forward(pt, px.to(dev), out_dtype=torch.float32)
    assert rel_l2(feats.cpu()[:, ::9, ::16], g["n4_tower"]) < TOL[dtype]
    st = g["n4_tower_stats"]
    norms = feats.double().reshape(5, -1).norm(dim=1).cpu().numpy()
    assert np.allclose(norms, st[:, 1], rtol=5e-3)
    A = W.ADAPTER_8B
    pr = ops.pack_resampler(W.sub_state(asd, "sampler.post_qformer."), 1024, 8, 576, dtype, dev, A.ln_eps)
    pg = ops.pack_gated(W.sub_state(asd, "mm_projector."), A, dtype, dev)
    comp = ops.resampler_forward(pr, feats[1:])
    assert rel_l2(comp.cpu()[:, ::3, ::16], g["n4_compressed"]) < TOL[dtype]
    loc = ops.mlp_forward(pg.mlp, comp.reshape(-1, 1024)).view(4, 144, 4096)
    assert rel_l2(loc.cpu()[:, ::3, ::64], g["n4_local"]) < TOL[dtype]
    glob = ops.gated_forward(pg, feats[:1])[0]
    assert rel_l2(glob.cpu()[::9, ::64], g["n4_global"]) < TOL[dtype]


def test_tower_batch_invariance(dev, tiny_weights):
    """Per-crop independence: a crop's features do not depend on its batch neighbours, bit
    for bit -- the property the multi-GPU sharding relies on."""
    from slime_amd import ops, weights as W
    tsd, _ = tiny_weights
    pt = ops.pack_tower(tsd, W.TINY, torch.bfloat16, dev)
    px = W.synthetic_pixels(5, seed=3).to(dev)
    full = ops.tower_forward(pt, px, out_dtype=torch.float32)
    for lo, hi in ((0, 2), (2, 5), (4, 5)):
        part = ops.tower_forward(pt, px[lo:hi].contiguous(), out_dtype=torch.float32)
        assert torch.equal(part.cpu(), full[lo:hi].cpu())


def test_tower_rejects_wrong_size(dev, tiny_weights):
    from slime_amd import ops, weights as W
    tsd, _ = tiny_weights
    pt = ops.pack_tower(tsd, W.TINY, torch.bfloat16, dev)
    with pytest.raises(ValueError, match="doesn't match model"):
        ops.tower_forward(pt, torch.zeros(1, 3, 224, 224, device=dev))


def test_full_size_shard_invariance(dev):
    """BASELINE config 2 at full ViT-L/14-336 size: the 40-crop tower output (two-stream product path) equals, bit for
    bit, the concatenation of 8 shards of 5 crops (the 8-GPU layout of SURVEY 8e) and of 2 shards of 20 -- although
    the shards dispatch to different GEMM kernels (stream kernel, 192/256-row ping-pong tiles) -- and is deterministic
    from run to run.  This is the size-independent property the multi-GPU all-gather relies on."""
    from slime_amd import weights as W
    from slime_amd.model.multimodal_encoder.clip_encoder import HipCLIPVisionModel
    vm = HipCLIPVisionModel(W.CLIP_L_336)
    vm.load_state_dict(W.make_tower_state_dict(W.CLIP_L_336, seed=1234))
    vm.to(dev).to(torch.bfloat16)
    px = W.synthetic_pixels(40, seed=9).to(dev).to(torch.bfloat16)
    full = vm.encode(px)
    again = vm.encode(px)
    assert full.shape == (40, 576, 1024) and torch.isfinite(full.float()).all()
    assert torch.equal(full, again)
    # 1 / 2 / 3 crops: the 64x64 ring tile (grids that leave half the CUs without a 128x128 workgroup) against the direct-B /
    # ping-pong kernels of the 20-crop halves; 1 crop is BASELINE config 1's launch shape
    for shard in (1, 2, 3, 5, 20):
        parts = torch.cat([vm.encode(px[i:i + shard].contiguous()) for i in range(0, 40, shard)])
        assert torch.equal(parts, full), shard


# ------------------------------------------------------------------------------------------------ production sizes
@pytest.fixture(scope="module")
def full20():
    """One CPU-oracle pass over 20 ViT-L/14-336 crops (~20-40 s on the GPU box's host cores), shared by the tests below.
    Crops 0..16 are one BASELINE-config-3 image (1 global + 16 local crops, 4 x 4 grid); 17..19 fill the half batch."""
    from slime_amd import weights as W
    from oracle import slime_oracle as O
    tsd = W.strip_tower_prefix(W.make_tower_state_dict(W.CLIP_L_336, seed=1234))
    asd = W.make_adapter_state_dict(W.ADAPTER_8B, seed=4321)
    px = W.synthetic_pixels(20, seed=77)
    feats = O.tower_forward(tsd, W.CLIP_L_336, px)
    return tsd, asd, px, feats


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_production_half_batch_tower_vs_oracle(dev, full20, dtype):
    """the tower at its PRODUCTION launch shapes -- one 20-crop half batch, M = 11540 rows: four-wave
    stream kernel for qkv / fc1, 256-row ping-pong for out_proj / fc2 (K = 4096 -> KTAG 1), attn64r -- against the fp32
    oracle directly (not through a bit-equality chain), full tensors, bf16 and fp16."""
    from slime_amd import ops, weights as W
    tsd, _, px, ref = full20
    pt = ops.pack_tower(tsd, W.CLIP_L_336, dtype, dev)
    out = ops.tower_forward(pt, px.to(dev), out_dtype=torch.float32).cpu()
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < TOL[dtype]
    per_crop = ((out - ref).flatten(1).norm(dim=1) / ref.flatten(1).norm(dim=1))
    assert float(per_crop.max()) < TOL[dtype] * 1.3, per_crop
    per_tok = ((out - ref).norm(dim=-1) / ref.norm(dim=-1)).flatten()
    assert float(per_tok.max()) < TOL[dtype] * 4, "no single token far off (a wrong tile would hide in the aggregate)"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_config1_single_crop_global_only_vs_oracle(dev, full20, dtype):
    """BASELINE config 1 at full size on the HIP path: ONE 336 x 336 crop (global view only, batch 1) through the ViT-L/14-336
    tower -- every GEMM on the 64x64 ring tile, attention over 16 (crop, head) pairs -- and the GatedBlock on that global view
    (clip_encoder.py:46-58 tensor branch; llava_arch.py:261-267 plain batch; projector/builder.py:179-209), against the fp32
    oracle directly.  north_star's 1e-3 is asserted at `global` for fp16; bf16 is held to its stated end-of-chain bound.  The
    1-crop pass must also equal crop 0 of the 20-crop pass bit for bit (shard invariance at the smallest grid)."""
    from slime_amd import ops, weights as W
    from oracle import slime_oracle as O
    tsd, asd, px, ref_feats = full20
    A = W.ADAPTER_8B
    proj_sd = W.sub_state(asd, "mm_projector.")
    pt = ops.pack_tower(tsd, W.CLIP_L_336, dtype, dev)
    one = ops.tower_forward(pt, px[:1].to(dev), out_dtype=torch.float32)
    assert one.shape == (1, 576, 1024)
    assert rel_l2(one.cpu(), ref_feats[:1]) < TOL[dtype]
    per_tok = ((one.cpu() - ref_feats[:1]).norm(dim=-1) / ref_feats[:1].norm(dim=-1)).flatten()
    assert float(per_tok.max()) < TOL[dtype] * 4
    batch = ops.tower_forward(pt, px.to(dev), out_dtype=torch.float32)
    assert torch.equal(one, batch[:1])
    g_ref = O.gated_block_forward(proj_sd, ref_feats[0], A.num_heads)
    pg = ops.pack_gated(proj_sd, A, dtype, dev)
    feats_t = ops.tower_forward(pt, px[:1].to(dev), out_dtype=dtype)
    bound = {torch.float16: 1e-3, torch.bfloat16: 1.2e-2}[dtype]
    # fused adapter with no local crops (one C-ABI call on the tower's 16-bit features) and the per-module GatedBlock launch sequence
    fused = ops.adapter_forward(pg, None, feats_t, 1, 0, 1, 1, False, -1, torch.float32)
    assert fused.shape == (1, 576, 4096)
    eg = rel_l2(fused[0].cpu(), g_ref)
    print(f"config 1 (1 crop, global only) {dtype}: tower {rel_l2(one.cpu(), ref_feats[:1]):.3e} global {eg:.3e}")
    assert eg <= bound, (dtype, eg)
    mod = ops.gated_forward(pg, one)[0]
    assert rel_l2(mod.cpu(), g_ref) <= bound
    for lg in (0, 1):                                     # stage-1 expert selection (projector/builder.py:198-201)
        e = ops.gated_forward(pg, one, learnable_gated=lg)[0]
        assert rel_l2(e.cpu(), O.gated_block_forward(proj_sd, ref_feats[0], A.num_heads, learnable_gated=lg)) <= bound


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_config3_image_1_plus_16_crops_vs_oracle(dev, full20, dtype):
    """BASELINE config 3's unit at real size: one image of 1 global + 16 local crops at ViT-L dims through the tower and the
    FUSED adapter (GatedBlock on the global view, post_qformer + MLP + 4 x 4 spatial merge on the 16 local crops) against
    the oracle stage by stage (llava_arch.py:212-255 order).  (The reference slicer never yields 16 local crops -- SURVEY
    section 8d: the tensor is fed directly, the grid is explicit.)"""
    from slime_amd import ops, weights as W
    from oracle import slime_oracle as O
    tsd, asd, px, ref_feats = full20
    A = W.ADAPTER_8B
    pt = ops.pack_tower(tsd, W.CLIP_L_336, dtype, dev)
    feats = ops.tower_forward(pt, px[:17].to(dev), out_dtype=dtype)
    assert rel_l2(feats.float().cpu(), ref_feats[:17]) < TOL[dtype] * 1.2
    pg = ops.pack_gated(W.sub_state(asd, "mm_projector."), A, dtype, dev)
    post = ops.pack_resampler(W.sub_state(asd, "sampler.post_qformer."), 1024, 8, 576, dtype, dev, A.ln_eps)
    tokens = ops.adapter_forward(pg, post, feats, 1, 16, 4, 4, True, -1, torch.float32)[0].cpu()      # [576 + 16*144, 4096]
    proj_sd, post_sd = W.sub_state(asd, "mm_projector."), W.sub_state(asd, "sampler.post_qformer.")
    g_ref = O.gated_block_forward(proj_sd, ref_feats[0], A.num_heads)
    comp = O.resampler_forward(post_sd, ref_feats[1:17], A.num_heads, A.ln_eps)
    merged_ref = O.spatial_merge(O.mlp_projector(proj_sd, comp), 4, 4, 12)
    assert tokens.shape == (576 + 16 * 144, 4096)
    assert rel_l2(tokens[:576], g_ref) < TOL[dtype] * 2
    assert rel_l2(tokens[576:], merged_ref) < TOL[dtype] * 2


def test_projector_outputs_within_north_star_tolerance(dev, full20, record_property):
    """north_star: "projector outputs within 1e-3 rel-err of reference".  ViT-L/14-336 tower + fused adapter at SliME-8B dims,
    fp16 operands -- the reference's inference dtype (llava/model/builder.py:43: torch_dtype=torch.float16) -- on (a) one
    BASELINE-config-2 image (1 global + 4 local crops, 2 x 2 merge) and (b) the 17-crop config-3 image of the 20-crop fixture:
    rel-L2 <= 1e-3 at `global` and `merged_local` against the fp32 oracle.  The bf16 figures (BASELINE's bench dtype; 2^-9
    operand rounding) are recorded beside them and held to the stated bf16 end-of-chain tolerance, not to 1e-3: measured
    fp16 7.5e-4 / 6.5e-4, bf16 6.2e-3 / 5.2e-3 (tools/north_star_parity.py, profiles/r03_north_star_parity.txt)."""
    from slime_amd import ops, weights as W
    from oracle import slime_oracle as O
    tsd, asd, px, ref_feats = full20
    A = W.ADAPTER_8B
    proj_sd, post_sd = W.sub_state(asd, "mm_projector."), W.sub_state(asd, "sampler.post_qformer.")
    cases = {"cfg2_1+4": (5, 4, 2, 2), "cfg3_1+16": (17, 16, 4, 4)}
    refs = {}
    for name, (n, nl, nw, nh) in cases.items():
        g_ref = O.gated_block_forward(proj_sd, ref_feats[0], A.num_heads)
        comp = O.resampler_forward(post_sd, ref_feats[1:n], A.num_heads, A.ln_eps)
        refs[name] = (g_ref, O.spatial_merge(O.mlp_projector(proj_sd, comp), nw, nh, 12))
    for dtype, bound in ((torch.float16, 1e-3), (torch.bfloat16, 1.2e-2)):
        pt = ops.pack_tower(tsd, W.CLIP_L_336, dtype, dev)
        pg = ops.pack_gated(proj_sd, A, dtype, dev)
        post = ops.pack_resampler(post_sd, 1024, 8, 576, dtype, dev, A.ln_eps)
        for name, (n, nl, nw, nh) in cases.items():
            feats = ops.tower_forward(pt, px[:n].to(dev), out_dtype=dtype)
            tok = ops.adapter_forward(pg, post, feats, 1, nl, nw, nh, True, -1, torch.float32)[0].cpu()
            eg, el = rel_l2(tok[:576], refs[name][0]), rel_l2(tok[576:], refs[name][1])
            record_property(f"{name}_{str(dtype).split('.')[-1]}", f"global {eg:.2e} merged_local {el:.2e}")
            print(f"north_star parity {name} {dtype}: global {eg:.3e} merged_local {el:.3e}")
            assert eg <= bound and el <= bound, (name, dtype, eg, el)


def test_config3_68_crops_shard_invariance(dev):
    """BASELINE config 3's batch (4 images x (1+16) = 68 crops) at full size: the product path's output equals, bit for bit,
    the concatenation of the 8-GPU partition of SURVEY section 8e (blocks of ceil(68/8) = 9 crops, the last of 5) and is
    deterministic -- the property that makes the sharded result identical to the 1-GPU result."""
    from slime_amd import weights as W
    from slime_amd.dist import shard_bounds
    from slime_amd.model.multimodal_encoder.clip_encoder import HipCLIPVisionModel
    vm = HipCLIPVisionModel(W.CLIP_L_336)
    vm.load_state_dict(W.make_tower_state_dict(W.CLIP_L_336, seed=1234))
    vm.to(dev).to(torch.bfloat16)
    px = W.synthetic_pixels(68, seed=19).to(dev).to(torch.bfloat16)
    full = vm.encode(px)
    assert full.shape == (68, 576, 1024) and torch.isfinite(full.float()).all()
    parts = []
    for r in range(8):
        lo, hi, per = shard_bounds(68, 8, r)
        assert per == 9 and hi - lo == (9 if r < 7 else 5)
        parts.append(vm.encode(px[lo:hi].contiguous()))
    assert torch.equal(torch.cat(parts), full)


def outlier_tower_state(cfg, seed, scale):
    """Seeded tower whose residual stream carries a few 'massive activation' channels (the shape real CLIP checkpoints have,
    which N(0, sigma) weights hide): fc2 of layers 0 and L/2 gets +-`scale`/2 bias on 4 channels and 2 weight rows x `scale`."""
    from slime_amd import weights as W
    sd = W.strip_tower_prefix(W.make_tower_state_dict(cfg, seed=seed))
    D = cfg.hidden_size
    for layer in (0, cfg.num_hidden_layers // 2):
        b = sd[f"encoder.layers.{layer}.mlp.fc2.bias"].clone()
        w = sd[f"encoder.layers.{layer}.mlp.fc2.weight"].clone()
        for j, ch in enumerate((5, D // 3, D // 2 + 1, D - 7)):
            b[ch] = (scale / 2) * (1 if j % 2 == 0 else -1)
        for ch in (11, D - 20):
            w[ch] *= scale
        sd[f"encoder.layers.{layer}.mlp.fc2.bias"], sd[f"encoder.layers.{layer}.mlp.fc2.weight"] = b, w
    return sd


@pytest.mark.parametrize("scale", [30.0, 100.0])
def test_outlier_channel_stress(dev, scale):
    """residual channels at x30 / x100 the typical magnitude.  fp32 residual stream + fp32 LayerNorm
    statistics keep 16-bit operands within the stated per-stage tolerances (bf16 1.5e-2, fp16 3e-3) at both geometries."""
    from slime_amd import ops, weights as W
    from oracle import slime_oracle as O
    for cfg, n in ((W.TINY, 3), (W.CLIP_L_336, 2)):
        sd = outlier_tower_state(cfg, 5, scale)
        px = W.synthetic_pixels(n, seed=6)
        ref, hs = O.tower_forward(sd, cfg, px), O.clip_hidden_states(sd, cfg, px, 1)
        assert float(hs[1].abs().max()) > scale * 0.4, "the generator did produce outlier activations"
        for dtype in (torch.float16, torch.bfloat16):
            out = ops.tower_forward(ops.pack_tower(sd, cfg, dtype, dev), px.to(dev), out_dtype=torch.float32).cpu()
            err = rel_l2(out, ref)
            print(f"outlier x{scale:g} {'tiny' if cfg is W.TINY else 'ViT-L'} {dtype}: rel-L2 {err:.2e}")
            assert torch.isfinite(out).all() and err < TOL[dtype], (cfg.hidden_size, dtype, err)
