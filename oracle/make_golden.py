#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (build container only).

TEST INFRASTRUCTURE.  Run:  python oracle/make_golden.py   (needs /root/reference; CPU; ~2 min)

The reference (yfzhang114/SliME @ /root/reference) is pure Python and ships no tests or golden
vectors, so the oracle is pinned with vectors produced here by importing the reference's own
modules -- ``CLIPVisionTower`` (llava/model/multimodal_encoder/clip_encoder.py), ``build_vision_projector``
(llava/model/multimodal_projector/builder.py), ``build_vision_sampler`` (llava/model/multimodal_resampler/
builder.py), ``process_images`` / ``get_anyres_image_grid_shape`` (llava/mm_utils.py), ``cal_num_of_slices`` /
``process_image_any_res`` / ``process_image_naive`` (llava/process_image.py) -- loading the seeded
weights of ``slime_amd.weights`` into them, and saving inputs' seeds + outputs.  Nothing of the
reference travels: only the numbers below are committed.

Import shims (no edits to the reference):
  1. ``llava/process_image.py:4,7`` imports torchvision (absent here) for commented-out code only ->
     import that one module under temporary stub modules, then import ``llava`` normally.
  2. ``process_anyres_image`` calls ``processor.crop_size.values()`` (mm_utils.py:194); transformers 5.x
     returns a SizeDict without ``.values()`` -> pass a namespace exposing dict-typed sizes and the
     real processor's ``preprocess``.
"""
from __future__ import annotations

import importlib
import os
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

from slime_amd import weights as W          # noqa: E402  (seeded generators; no compute)

OUT = os.path.join(REPO, "tests", "golden")


def import_reference():
    import transformers  # noqa: F401  (must see the true torchvision state first)
    sys.path.insert(0, REF)
    stub = types.ModuleType("llava")
    stub.__path__ = [os.path.join(REF, "llava")]
    sys.modules["llava"] = stub
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvt.ToTensor = object
    tvt.ToPILImage = object
    tvf = types.ModuleType("torchvision.transforms.functional")
    tv.transforms = tvt
    tvt.functional = tvf
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.functional": tvf})
    importlib.import_module("llava.process_image")
    for k in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "llava"):
        del sys.modules[k]
    import llava  # noqa: F401
    return llava


def make_processor_shim():
    from transformers import CLIPImageProcessor
    real = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336})
    shim = SimpleNamespace(crop_size={"height": 336, "width": 336}, size={"shortest_edge": 336},
                           image_mean=list(real.image_mean), image_std=list(real.image_std),
                           preprocess=real.preprocess)
    return real, shim


def build_reference_tower(vcfg: W.VisionConfig, tower_sd):
    """A reference ``CLIPVisionTower`` holding the seeded weights (random-init HF model saved to a
    temp dir, whose absolute path the reference's builder accepts: multimodal_encoder/builder.py:7-8)."""
    from transformers import CLIPVisionConfig, CLIPVisionModel, CLIPImageProcessor
    from llava.model.multimodal_encoder.builder import build_vision_tower
    tmp = tempfile.mkdtemp(prefix="slime_tower_")
    hf_cfg = CLIPVisionConfig(hidden_size=vcfg.hidden_size, intermediate_size=vcfg.intermediate_size,
                              num_hidden_layers=vcfg.num_hidden_layers,
                              num_attention_heads=vcfg.num_attention_heads, image_size=vcfg.image_size,
                              patch_size=vcfg.patch_size, hidden_act="quick_gelu",
                              layer_norm_eps=vcfg.layer_norm_eps)
    CLIPVisionModel(hf_cfg).save_pretrained(tmp)
    CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336}).save_pretrained(tmp)
    args = SimpleNamespace(mm_vision_tower=tmp, mm_vision_select_layer=-2, mm_vision_select_feature="patch")
    tower = build_vision_tower(args)
    model_keys = set(tower.vision_tower.state_dict().keys())
    flat = W.strip_tower_prefix(tower_sd)
    sd = {}
    for k, v in flat.items():
        if k in model_keys:
            sd[k] = v
        elif "vision_model." + k in model_keys:
            sd["vision_model." + k] = v
    missing, unexpected = tower.vision_tower.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "position_ids" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    tower.eval()
    return tower


def build_reference_adapter(acfg: W.AdapterConfig, adapter_sd, learnable_gated=-1):
    from llava.model.multimodal_projector.builder import build_vision_projector
    from llava.model.multimodal_resampler.builder import build_vision_sampler
    cfg = SimpleNamespace(mm_projector_type="gated", mm_hidden_size=acfg.mm_hidden_size,
                          hidden_size=acfg.hidden_size, mm_learnable_gated=learnable_gated,
                          mm_resampler_type="cosine", mm_resampler_topp=0.95,
                          mm_resampler_dim=acfg.local_queries, mm_resampler_temp=1.0, pad_token_id=0)
    proj = build_vision_projector(cfg)
    samp = build_vision_sampler(cfg)
    # the reference hard-codes num_heads = mm_hidden_size // 128 (projector/builder.py:46,
    # resampler/builder.py:242): the fixture geometries must use head_dim 128
    assert acfg.head_dim == 128
    proj.load_state_dict(W.sub_state(adapter_sd, "mm_projector."), strict=True)
    samp.load_state_dict(W.sub_state(adapter_sd, "sampler."), strict=True)
    return proj.eval(), samp.eval()


def stats(x: torch.Tensor):
    x = x.double().reshape(x.shape[0], -1)
    return np.stack([x.mean(1).numpy(), x.norm(dim=1).numpy(), x.abs().amax(1).numpy()], axis=1)


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    os.makedirs(OUT, exist_ok=True)
    import_reference()
    from llava import mm_utils as ref_mm
    from llava import process_image as ref_pi
    from llava.model.multimodal_resampler.sampler import get_abs_pos as ref_get_abs_pos
    from PIL import Image

    # ------------------------------------------------------------------ (1) slicer / grid logic
    sizes = [(336, 336), (672, 672), (1344, 1344), (640, 480), (1920, 1080), (4000, 300), (300, 4000),
             (100, 100), (336, 337), (337, 336), (335, 335), (672, 336), (336, 672), (1008, 336),
             (1000, 1000), (1024, 768), (768, 1024), (800, 600), (600, 800), (1280, 720), (720, 1280),
             (2048, 2048), (3000, 2000), (2000, 3000), (500, 500), (475, 476), (476, 475), (674, 674),
             (823, 823), (824, 824), (952, 952), (953, 953), (1500, 200), (200, 1500), (336, 1008),
             (1344, 336), (336, 1344), (900, 1100), (1100, 900), (1, 1), (14, 14), (5000, 5000),
             (1680, 336), (336, 1680), (2016, 336), (1008, 672), (672, 1008), (1344, 672), (672, 1344),
             (999, 333), (333, 999), (224, 224), (448, 448), (512, 384), (384, 512), (1600, 1200)]
    pin = "[(336, 672), (672, 336), (672, 672), (1008, 336), (336, 1008)]"   # train.py:1109 value, dead
    grid = np.array([ref_mm.get_anyres_image_grid_shape(s, pin, 336) for s in sizes], dtype=np.int64)
    uhd = np.array([ref_mm.select_best_resolution_uhd(s, (336, 336)) for s in sizes], dtype=np.int64)
    slices = np.array([ref_pi.cal_num_of_slices(w, h) for (w, h) in sizes], dtype=np.int64)
    pts = [(336, 672), (672, 336), (672, 672), (1008, 336), (336, 1008)]
    sbr = np.array([ref_mm.select_best_resolution(s, pts) for s in sizes], dtype=np.int64)
    np.savez(os.path.join(OUT, "slicer_grid.npz"), sizes=np.array(sizes, dtype=np.int64), grid=grid,
             uhd=uhd, slices=slices, select_best_resolution=sbr)
    print("slicer_grid:", len(sizes), "sizes")

    # ------------------------------------------------------------------ (2) process_images pixels
    real_proc, shim = make_processor_shim()
    rng = np.random.default_rng(0)
    img_specs = [(672, 672), (336, 336), (640, 480), (500, 900), (1344, 1344), (300, 200)]
    rec = {}
    sample_idx = rng.integers(0, 3 * 336 * 336, size=512)
    rec["sample_idx"] = sample_idx
    for i, (w, h) in enumerate(img_specs):
        arr = np.random.default_rng(100 + i).integers(0, 256, (h, w, 3), dtype=np.uint8)
        img = Image.fromarray(arr, "RGB")
        for mode in ("anyres", "pad", "any_res", "pad_then_devide"):
            cfg = SimpleNamespace(image_aspect_ratio=mode, image_grid_pinpoints=pin)
            proc = shim if mode == "anyres" else real_proc
            out = ref_mm.process_images([img], proc, cfg)
            out = out[0] if isinstance(out, (list, tuple)) or out.dim() == 5 else out
            if out.dim() == 3:
                out = out.unsqueeze(0)
            flat = out.reshape(out.shape[0], -1).double()
            rec[f"img{i}_{mode}_shape"] = np.array(out.shape, dtype=np.int64)
            rec[f"img{i}_{mode}_sum"] = flat.sum(1).numpy()
            rec[f"img{i}_{mode}_sumsq"] = (flat * flat).sum(1).numpy()
            rec[f"img{i}_{mode}_samples"] = flat[:, sample_idx].float().numpy()
    rec["img_specs"] = np.array(img_specs, dtype=np.int64)
    rec["image_mean"] = np.array(real_proc.image_mean, dtype=np.float64)
    rec["image_std"] = np.array(real_proc.image_std, dtype=np.float64)
    np.savez(os.path.join(OUT, "slicer_pixels.npz"), **rec)
    print("slicer_pixels: done")

    # ------------------------------------------------------------------ (3) tiny geometry, full tensors
    vt, at_ref = W.TINY, W.ADAPTER_TINY
    tsd = W.make_tower_state_dict(vt, seed=11)
    asd = W.make_adapter_state_dict(at_ref, seed=12)
    tower = build_reference_tower(vt, tsd)
    proj, samp = build_reference_adapter(at_ref, asd)
    hs_count = None
    g = {}
    for n_local, (iw, ih) in ((2, (336, 336)), (4, (672, 672)), (6, (1344, 1344))):
        px = W.synthetic_pixels(1 + n_local, seed=20 + n_local)
        outs = tower.vision_tower(px, output_hidden_states=True)
        hs_count = len(outs.hidden_states)
        feats = tower(px)
        assert torch.equal(feats, outs.hidden_states[-2][:, 1:])
        glob = proj(feats[0])
        comp = samp.post_qformer(feats[1:])
        loc = proj(comp)
        nw, nh = ref_mm.get_anyres_image_grid_shape((iw, ih), pin, 336)
        merged = loc.view(nh, nw, samp.grid_size, samp.grid_size, -1).permute(0, 2, 1, 3, 4).contiguous().flatten(0, 3)
        tg = torch.Generator().manual_seed(300 + n_local)
        text = torch.randn(9, at_ref.hidden_size, generator=tg)
        mask = torch.tensor([1, 1, 1, 1, 1, 1, 0, 0, 1], dtype=torch.bool)
        scores = samp.selector(merged, text, mask)
        routed = samp(merged, text, mask)
        k = f"n{n_local}_"
        if n_local == 2:
            g[k + "tower"] = feats.numpy()
            g[k + "hidden0"] = outs.hidden_states[0].numpy()[:, ::9, ::4]
            g[k + "hidden1"] = outs.hidden_states[1].numpy()[:, ::9, ::4]
            g[k + "global"] = glob.numpy()
            g[k + "compressed"] = comp.numpy()
            g[k + "local"] = loc.numpy()
        else:
            g[k + "tower"] = feats.numpy()[:, ::7, ::5]
            g[k + "global"] = glob.numpy()[::7, ::5]
            g[k + "compressed"] = comp.numpy()[:, ::3, ::5]
            g[k + "local"] = loc.numpy()[:, ::3, ::5]
        g[k + "tower_stats"] = stats(feats)
        g[k + "merged_stats"] = stats(merged.unsqueeze(0))
        g[k + "merged_rows"] = merged.numpy()[[0, 1, 11, 12, 13, 143, 144, 145, merged.shape[0] - 1]]
        g[k + "grid"] = np.array([nw, nh], dtype=np.int64)
        g[k + "image_size"] = np.array([iw, ih], dtype=np.int64)
        g[k + "router_scores"] = scores.numpy()
        g[k + "router_rows"] = np.array([routed.shape[0]], dtype=np.int64)
        g[k + "router_first"] = routed.numpy()[:4]
        print(f"tiny n={n_local}: tower {tuple(feats.shape)} global {tuple(glob.shape)} "
              f"local {tuple(loc.shape)} routed {routed.shape[0]}/{merged.shape[0]}")
    # expert-select variants and the (N,576,D) batched gated path (llava_arch.py:261-267 branch)
    px = W.synthetic_pixels(2, seed=31)
    feats = tower(px)
    g["batched_gated"] = proj(feats).numpy()[:, ::7, ::5]
    for lg in (0, 1):
        p2, _ = build_reference_adapter(at_ref, asd, learnable_gated=lg)
        g[f"expert{lg}"] = p2(feats[0]).numpy()[::7, ::5]
    g["hidden_states_len"] = np.array([hs_count], dtype=np.int64)
    g["abs_pos_12_to_24"] = ref_get_abs_pos(asd["sampler.post_qformer.pos_embed"], (24, 24)).float().numpy()[::5, ::3]
    np.savez(os.path.join(OUT, "tiny_stages.npz"), **g)
    print("tiny_stages: done, hidden_states len", hs_count)

    # ------------------------------------------------------------------ (4) full ViT-L dims, sub-sampled
    vl, al = W.CLIP_L_336, W.ADAPTER_8B
    tsd = W.make_tower_state_dict(vl, seed=1234)
    asd = W.make_adapter_state_dict(al, seed=4321)
    tower = build_reference_tower(vl, tsd)
    proj, samp = build_reference_adapter(al, asd)
    f = {}
    for n_local, (iw, ih) in ((4, (672, 672)), (6, (1344, 1344))):
        px = W.synthetic_pixels(1 + n_local, seed=40 + n_local)
        feats = tower(px)
        glob = proj(feats[0])
        comp = samp.post_qformer(feats[1:])
        loc = proj(comp)
        k = f"n{n_local}_"
        f[k + "tower"] = feats.numpy()[:, ::9, ::16]
        f[k + "tower_stats"] = stats(feats)
        f[k + "global"] = glob.numpy()[::9, ::64]
        f[k + "global_stats"] = stats(glob.unsqueeze(0))
        f[k + "compressed"] = comp.numpy()[:, ::3, ::16]
        f[k + "compressed_stats"] = stats(comp)
        f[k + "local"] = loc.numpy()[:, ::3, ::64]
        f[k + "local_stats"] = stats(loc)
        print(f"full n={n_local}: done")
    np.savez(os.path.join(OUT, "full_stages.npz"), **f)

    total = sum(os.path.getsize(os.path.join(OUT, x)) for x in os.listdir(OUT))
    print("golden bytes:", total)


if __name__ == "__main__":
    main()
