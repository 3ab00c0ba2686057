"""Pin the CPU oracle (oracle/slime_oracle.py) against golden vectors produced by the reference
itself (oracle/make_golden.py, run in the build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import slime_oracle as O
from slime_amd import weights as W
from conftest import GOLDEN, rel_l2

TOL = 2e-5   # fp32 restatement vs fp32 reference (different op order / sdpa vs explicit softmax)


@pytest.fixture(scope="module")
def tiny():
    g = np.load(os.path.join(GOLDEN, "tiny_stages.npz"))
    tsd = W.strip_tower_prefix(W.make_tower_state_dict(W.TINY, seed=11))
    asd = W.make_adapter_state_dict(W.ADAPTER_TINY, seed=12)
    return g, tsd, asd


def test_hidden_states_semantics(tiny):
    g, tsd, _ = tiny
    assert int(g["hidden_states_len"][0]) == W.TINY.num_hidden_layers + 1
    px = W.synthetic_pixels(3, seed=22)
    hs = O.clip_hidden_states(tsd, W.TINY, px)
    assert len(hs) == W.TINY.num_hidden_layers + 1
    assert rel_l2(hs[0][:, ::9, ::4], g["n2_hidden0"]) < TOL
    assert rel_l2(hs[1][:, ::9, ::4], g["n2_hidden1"]) < TOL
    # select_layer=-2 == hidden_states[L-1]; the last layer is dead compute
    assert torch.equal(O.tower_forward(tsd, W.TINY, px), hs[-2][:, 1:])


@pytest.mark.parametrize("n_local,size", [(2, (336, 336)), (4, (672, 672)), (6, (1344, 1344))])
def test_tiny_stages(tiny, n_local, size):
    g, tsd, asd = tiny
    k = f"n{n_local}_"
    assert tuple(g[k + "image_size"]) == size
    px = W.synthetic_pixels(1 + n_local, seed=20 + n_local)
    tg = torch.Generator().manual_seed(300 + n_local)
    text = torch.randn(9, W.ADAPTER_TINY.hidden_size, generator=tg)
    mask = torch.tensor([1, 1, 1, 1, 1, 1, 0, 0, 1], dtype=torch.bool)
    out = O.encode_image(tsd, asd, W.TINY, W.ADAPTER_TINY, px, size, text, mask)
    assert tuple(g[k + "grid"]) == O.anyres_grid_shape(size)
    if n_local == 2:
        pairs = [("tower", out["tower"]), ("global", out["global"]),
                 ("compressed", out["compressed"]), ("local", out["local"])]
    else:
        pairs = [("tower", out["tower"][:, ::7, ::5]), ("global", out["global"][::7, ::5]),
                 ("compressed", out["compressed"][:, ::3, ::5]), ("local", out["local"][:, ::3, ::5])]
    for name, mine in pairs:
        assert rel_l2(mine, g[k + name]) < TOL, name
    rows = [0, 1, 11, 12, 13, 143, 144, 145, out["merged"].shape[0] - 1]
    assert rel_l2(out["merged"][rows], g[k + "merged_rows"]) < TOL
    assert rel_l2(out["router_scores"], g[k + "router_scores"]) < TOL
    assert out["router_keep"].numel() == int(g[k + "router_rows"][0])
    assert rel_l2(out["merged"][out["router_keep"]][:4], g[k + "router_first"]) < TOL


def test_gated_variants(tiny):
    g, tsd, asd = tiny
    proj = W.sub_state(asd, "mm_projector.")
    feats = O.tower_forward(tsd, W.TINY, W.synthetic_pixels(2, seed=31))
    H = W.ADAPTER_TINY.num_heads
    assert rel_l2(O.gated_block_forward(proj, feats, H)[:, ::7, ::5], g["batched_gated"]) < TOL
    for lg in (0, 1):
        assert rel_l2(O.gated_block_forward(proj, feats[0], H, learnable_gated=lg)[::7, ::5], g[f"expert{lg}"]) < TOL
    pos = O.get_abs_pos(asd["sampler.post_qformer.pos_embed"], (24, 24)).float()
    assert np.array_equal(pos.numpy()[::5, ::3], g["abs_pos_12_to_24"])


def test_gate_mix_on_hidden_rows_reproduces_the_reference_output(tiny):
    """The arithmetic slime_gated_forward / slime_adapter_forward run since round 4 (slime_gate_premix): the gates applied to the
    projection MLP's HIDDEN rows, projection[2] once per token.  Restated in fp32 on the oracle's pieces and held against the golden
    vector the REFERENCE's GatedBlock.forward produced (projector/builder.py:190-206: two complete experts, then the mix): the
    identity W2 (g0 a0 + g1 a1) + b2 = g0 (W2 a0 + b2) + g1 (W2 a1 + b2) - (1 - g0 - g1) b2, with g0 + g1 = 1 / (1 + 1e-6)."""
    import torch.nn.functional as F
    g, tsd, asd = tiny
    proj = W.sub_state(asd, "mm_projector.")
    feats = O.tower_forward(tsd, W.TINY, W.synthetic_pixels(2, seed=31)).float()
    H = W.ADAPTER_TINY.num_heads
    attn_sd = {k[len("attn."):]: v for k, v in proj.items() if k.startswith("attn.")}
    w1, b1 = proj["projection.0.weight"].float(), proj["projection.0.bias"].float()
    w2, b2 = proj["projection.2.weight"].float(), proj["projection.2.bias"].float()
    a0 = F.gelu(F.linear(feats, w1, b1))
    a1 = F.gelu(F.linear(O.resampler_forward(attn_sd, feats, H), w1, b1))
    N, C, D = feats.shape
    gt = O.gate_weights(proj, feats.reshape(N * C, D)).reshape(N, C, 2)
    assert float((gt.sum(-1) - 1.0 / (1.0 + 1e-6)).abs().max()) < 1e-6
    out = F.linear(gt[..., 0:1] * a0 + gt[..., 1:2] * a1, w2, b2)
    assert rel_l2(out[:, ::7, ::5], g["batched_gated"]) < TOL
    assert rel_l2(out, O.gated_block_forward(proj, feats, H)) < 2e-6


@pytest.mark.parametrize("n_local,size", [(4, (672, 672)), (6, (1344, 1344))])
def test_full_dims(n_local, size):
    """ViT-L/14-336 + 1024->4096 adapter geometry, sub-sampled tensors and per-crop statistics."""
    g = np.load(os.path.join(GOLDEN, "full_stages.npz"))
    tsd = W.strip_tower_prefix(W.make_tower_state_dict(W.CLIP_L_336, seed=1234))
    asd = W.make_adapter_state_dict(W.ADAPTER_8B, seed=4321)
    px = W.synthetic_pixels(1 + n_local, seed=40 + n_local)
    out = O.encode_image(tsd, asd, W.CLIP_L_336, W.ADAPTER_8B, px, size)
    k = f"n{n_local}_"
    assert rel_l2(out["tower"][:, ::9, ::16], g[k + "tower"]) < TOL
    assert rel_l2(out["global"][::9, ::64], g[k + "global"]) < TOL
    assert rel_l2(out["compressed"][:, ::3, ::16], g[k + "compressed"]) < TOL
    assert rel_l2(out["local"][:, ::3, ::64], g[k + "local"]) < TOL
    from oracle.make_golden import stats
    for name, t in (("tower", out["tower"]), ("compressed", out["compressed"]), ("local", out["local"])):
        assert np.allclose(stats(t), g[k + name + "_stats"], rtol=1e-4, atol=1e-6), name


def test_grid_table():
    g = np.load(os.path.join(GOLDEN, "slicer_grid.npz"))
    for (w, h), grid, uhd in zip(g["sizes"], g["grid"], g["uhd"]):
        assert O.select_best_resolution_uhd((int(w), int(h))) == tuple(uhd), (w, h)
        assert O.anyres_grid_shape((int(w), int(h))) == tuple(grid), (w, h)


# ---- Pillow resample restatement (oracle/pil_resample.py): Pillow itself is the golden source -------------
RESAMPLE_CASES = [((672, 672), (336, 336)), ((1344, 1344), (336, 336)), ((640, 480), (672, 504)),
                  ((300, 200), (672, 448)), ((1500, 200), (1680, 224)), ((500, 336), (336, 336)),
                  ((336, 500), (336, 226)), ((37, 53), (336, 336)), ((673, 672), (672, 672)), ((336, 336), (336, 336))]


@pytest.mark.parametrize("src,dst", RESAMPLE_CASES)
def test_pil_resample_restatement_matches_pillow(src, dst):
    from PIL import Image
    from oracle import pil_resample as R
    (w, h), (ow, oh) = src, dst
    img = np.random.default_rng(w * 7919 + h).integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh)))
    assert np.array_equal(R.resize_bicubic_u8(img, ow, oh), ref)


def test_pil_resample_slicer_matches_reference_pixels():
    """uint8 slicer restatement (uhd grid -> resize+pad -> thumbnail) + CLIP normalisation reproduces the
    reference's own ``process_images(..., 'anyres')`` pixels (slicer_pixels.npz) bit for bit."""
    from oracle import pil_resample as R
    from slime_amd.image_processor import ClipImageProcessor
    g = np.load(os.path.join(GOLDEN, "slicer_pixels.npz"))
    proc = ClipImageProcessor()
    for i, (w, h) in enumerate(g["img_specs"]):
        w, h = int(w), int(h)
        arr = np.random.default_rng(100 + i).integers(0, 256, (h, w, 3), dtype=np.uint8)
        tw, th = O.select_best_resolution_uhd((w, h), (336, 336))
        canvas = R.resize_and_pad_u8(arr, tw, th)
        views = [R.resize_bicubic_u8(arr, 336, 336)]
        views += [canvas[y:y + 336, x:x + 336] for y in range(0, th, 336) for x in range(0, tw, 336)]
        out = np.stack([proc.normalize_u8(v) for v in views])
        assert tuple(out.shape) == tuple(g[f"img{i}_anyres_shape"])
        flat = out.reshape(out.shape[0], -1)
        assert np.array_equal(flat[:, g["sample_idx"]].astype(np.float32), g[f"img{i}_anyres_samples"]), i
