"""GPU parity of the composed hot path (tower, resampler, projector, gated block) against the CPU oracle
and the committed golden vectors (produced by the reference itself, oracle/make_golden.py).

Stated tolerances (rel-L2 vs the fp32 oracle; MFMA operands 16-bit, fp32 accumulate / residual
stream / LayerNorm / softmax statistics):
    fp16 operands : tower 3e-3, adapter stages 3e-3   (the reference's own fp16-vs-fp32 drift: 1.5e-3)
    bf16 operands : tower 1.5e-2, adapter stages 1.5e-2 (the reference's own bf16-vs-fp32 drift: 1.2e-2)
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 3e-3, torch.bfloat16: 1.5e-2}


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
    for shard in (5, 20):
        parts = torch.cat([vm.encode(px[i:i + shard].contiguous()) for i in range(0, 40, shard)])
        assert torch.equal(parts, full), shard
