"""INTEGRATION.md section 1 as a test: the REFERENCE's own glue -- ``LlavaMetaModel.__init__``
(llava/model/llava_arch.py:30-44), ``initialize_vision_modules`` (:52-119), ``LlavaMetaForCausalLM.encode_images`` (:212-269),
``prepare_inputs_labels_for_multimodal`` (:274-459) and the three statements of ``load_pretrained_model``
(llava/model/builder.py:161-166) -- executed UNCHANGED over ``slime_amd``'s three builders + ``get_anyres_image_grid_shape``,
swapped in exactly as the import swap of INTEGRATION.md does.

Build container only: the reference cannot travel, so this module is skipped where /root/reference is absent (the GPU box).
There is no GPU here either, so inside this test -- and only here -- ``slime_amd.ops``' device entry points are replaced by
the CPU oracle (the oracle is the checker's stand-in for libslime_hip; the product path never does this).  What is pinned is
the PLUGIN CONTRACT: constructor arguments, attributes, call signatures, shapes, dtypes and state-dict keys the reference's
glue relies on.  Outputs are compared with the reference-generated vectors of tests/golden/tiny_stages.npz."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import rel_l2

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "llava")), reason="the reference is only present in the build container")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ref_arch():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle.make_golden import import_reference
    import_reference()
    import llava.model.llava_arch as A
    return A


class _OracleOps:
    """CPU stand-ins for the ``slime_amd.ops`` calls the plugin modules make (same signatures), backed by oracle/slime_oracle.py."""

    def __init__(self):
        from oracle import slime_oracle as O
        from slime_amd import weights as W
        self.O, self.W = O, W

    # tower -----------------------------------------------------------------------------------
    def pack_tower(self, state_dict, cfg, dtype, device, select_layer=-2):
        sd = {k: v.detach().float().cpu() for k, v in self.W.strip_tower_prefix(state_dict).items()}
        return SimpleNamespace(sd=sd, cfg=cfg, select_layer=select_layer, dtype=dtype, layers_run=None, tensors={}, desc=None)

    def tower_forward(self, pt, pixels, out_dtype=None, keep_cls=False, want_hidden=False, out=None):
        f = self.O.tower_forward(pt.sd, pt.cfg, pixels.float().cpu(), pt.select_layer, "cls_patch" if keep_cls else "patch")
        f = f.to(out_dtype or pixels.dtype)
        if out is not None:
            out.copy_(f)
            return out
        return f

    # resampler / mlp / gated -------------------------------------------------------------------
    def pack_resampler(self, sd, dim, heads, n_kv, dtype, device, eps=1e-6):
        return SimpleNamespace(sd={k: v.detach().cpu() for k, v in sd.items()}, heads=heads, eps=eps, n_kv=n_kv, dtype=dtype)

    def resampler_forward(self, pr, x, want_t=False):
        return self.O.resampler_forward(pr.sd, x.float().cpu(), pr.heads, pr.eps)

    def pack_mlp(self, w1, b1, w2, b2, dtype, device):
        return SimpleNamespace(sd={"projection.0.weight": w1.detach().float().cpu(), "projection.0.bias": b1.detach().float().cpu(),
                                   "projection.2.weight": w2.detach().float().cpu(), "projection.2.bias": b2.detach().float().cpu()})

    def mlp_forward(self, pm, x):
        return self.O.mlp_projector(pm.sd, x.float().cpu())

    def pack_gated(self, sd, cfg, dtype, device):
        return SimpleNamespace(sd={k: v.detach().cpu() for k, v in sd.items()}, heads=cfg.num_heads)

    def gated_forward(self, pg, x, learnable_gated=-1):
        return torch.stack([self.O.gated_block_forward(pg.sd, xi.float().cpu(), pg.heads, learnable_gated) for xi in x])

    # router ------------------------------------------------------------------------------------
    def router_topp(self, local_f, text, mask, topp, temp):
        if local_f.shape[0] == 0:
            return torch.zeros((0,), dtype=torch.long)
        return self.O.router_select(self.O.router_cosine_scores(local_f.float(), text.float(), mask), topp, temp)


@pytest.fixture()
def swapped(ref_arch, monkeypatch):
    """The import swap of INTEGRATION.md section 1, applied to the reference's module object, + the oracle behind slime_amd.ops."""
    from slime_amd import ops
    from slime_amd.model.multimodal_encoder.builder import build_vision_tower
    from slime_amd.model.multimodal_projector.builder import build_vision_projector
    from slime_amd.model.multimodal_resampler.builder import build_vision_sampler
    from slime_amd.mm_utils import get_anyres_image_grid_shape
    for name, fn in (("build_vision_tower", build_vision_tower), ("build_vision_projector", build_vision_projector),
                     ("build_vision_sampler", build_vision_sampler), ("get_anyres_image_grid_shape", get_anyres_image_grid_shape)):
        assert hasattr(ref_arch, name), f"the reference no longer imports {name} into llava_arch"
        monkeypatch.setattr(ref_arch, name, fn)
    oo = _OracleOps()
    for name in ("pack_tower", "tower_forward", "pack_resampler", "resampler_forward", "pack_mlp", "mlp_forward", "pack_gated",
                 "gated_forward", "router_topp"):
        monkeypatch.setattr(ops, name, getattr(oo, name))
    return ref_arch


def _tower_dir(tmp_path):
    from slime_amd import weights as W
    from test_checkpoints import _write_hf_dir
    d = tmp_path / "clip_tiny"
    d.mkdir()
    _write_hf_dir(str(d), W.make_tower_state_dict(W.TINY, seed=11), W.TINY, nested_config=False)
    return str(d)


def _reference_model(A, cfg, vocab):
    """The reference's mixins over a minimal host: LlavaMetaModel needs a base whose __init__ takes the config and that owns
    ``embed_tokens`` (LlamaModel in the reference, llava_llama.py:38-42); LlavaMetaForCausalLM needs get_model() + config."""
    class _Base(nn.Module):
        def __init__(self, config):
            super().__init__()
            self.config = config
            self.embed_tokens = nn.Embedding(vocab, config.hidden_size)

    class Model(A.LlavaMetaModel, _Base):
        pass

    class Wrapper(A.LlavaMetaForCausalLM):
        def __init__(self, config):
            self.model = Model(config)
            self.config = config

        def get_model(self):
            return self.model

        @property
        def device(self):                                 # PreTrainedModel.device in the reference (used at llava_arch.py:400)
            return self.model.embed_tokens.weight.device

    return Wrapper(cfg)


def _config(tower_path, acfg):
    return SimpleNamespace(mm_vision_tower=tower_path, mm_vision_select_layer=-2, mm_vision_select_feature="patch",
                           mm_projector_type="gated", mm_hidden_size=acfg.mm_hidden_size, hidden_size=acfg.hidden_size,
                           mm_learnable_gated=-1, mm_resampler_type="cosine", mm_resampler_topp=0.95, mm_resampler_dim=acfg.local_queries,
                           mm_resampler_temp=1.0, mm_patch_merge_type="spatial", image_aspect_ratio="anyres", seperator=1919,
                           image_grid_pinpoints="[(336, 672), (672, 336), (672, 672), (1008, 336), (336, 1008)]",
                           tokenizer_padding_side="right", tokenizer_model_max_length=None, pad_token_id=0)


def test_reference_glue_runs_unchanged_over_swapped_builders(swapped, tmp_path):
    A = swapped
    from slime_amd import weights as W
    from slime_amd.model.multimodal_encoder.clip_encoder import CLIPVisionTower
    from slime_amd.model.multimodal_projector.builder import GatedBlock
    from slime_amd.model.multimodal_resampler.builder import TextGuidedSampler
    acfg = W.ADAPTER_TINY
    cfg = _config(_tower_dir(tmp_path), acfg)
    asd = W.make_adapter_state_dict(acfg, seed=12)
    VOCAB = 2048

    # ---- LlavaMetaModel.__init__ (llava_arch.py:30-44): delay_load tower + projector + sampler from the config
    m = _reference_model(A, cfg, VOCAB)
    model = m.get_model()
    assert isinstance(model.vision_tower, CLIPVisionTower) and not model.vision_tower.is_loaded
    assert isinstance(model.mm_projector, GatedBlock) and isinstance(model.sampler, TextGuidedSampler) and model.has_sampler

    # ---- initialize_vision_modules (:52-119) with pretrain_* checkpoint files written in the reference's key layout
    torch.save({"model.mm_projector." + k: v for k, v in W.sub_state(asd, "mm_projector.").items()}, tmp_path / "mm_projector.bin")
    torch.save({"model.sampler." + k: v for k, v in W.sub_state(asd, "sampler.").items()}, tmp_path / "sampler.bin")
    args = SimpleNamespace(vision_tower=cfg.mm_vision_tower, mm_vision_select_layer=-2, mm_vision_select_feature="patch",
                           pretrain_mm_mlp_adapter=str(tmp_path / "mm_projector.bin"), pretrain_mm_re_sampler=str(tmp_path / "sampler.bin"),
                           mm_patch_merge_type="spatial", mm_resampler_type="cosine", mm_resampler_topp=0.95,
                           mm_resampler_dim=acfg.local_queries, mm_resampler_temp=1.0, mm_projector_type="gated", mm_learnable_gated=-1)
    model.initialize_vision_modules(args)
    tower = m.get_vision_tower()
    assert tower.is_loaded and model.config.mm_hidden_size == tower.hidden_size == W.TINY.hidden_size
    assert model.config.use_mm_proj and model.config.seperator == 1919
    for k, v in W.sub_state(asd, "mm_projector.").items():
        assert torch.equal(model.mm_projector.state_dict()[k].float(), v.float()), k
    for k, v in W.sub_state(asd, "sampler.").items():
        assert torch.equal(model.sampler.state_dict()[k].float(), v.float()), k

    # ---- encode_images (:212-269), sampler branch, on the goldens' crops; the router's text = the goldens' text rows, delivered
    # through embed_tokens + get_pure_text_embedding (:162-210) exactly as the reference delivers it
    g = np.load(os.path.join(GOLD, "tiny_stages.npz"))
    mask9 = torch.tensor([1, 1, 1, 1, 1, 1, 0, 0, 1], dtype=torch.long)
    emb = model.embed_tokens
    with torch.no_grad():
        emb.weight.copy_(torch.randn(VOCAB, acfg.hidden_size, generator=torch.Generator().manual_seed(77)) * 0.1)
    for n_local, (iw, ih) in ((2, (336, 336)), (4, (672, 672))):
        text = torch.randn(9, acfg.hidden_size, generator=torch.Generator().manual_seed(300 + n_local))
        ids = torch.arange(100, 109)
        with torch.no_grad():
            emb.weight[ids] = text
        input_ids = torch.cat([ids, torch.tensor([A.IMAGE_TOKEN_INDEX])])[None]        # 9 text tokens + <image>
        attn = torch.cat([mask9, torch.tensor([1])])[None]
        px = W.synthetic_pixels(1 + n_local, seed=20 + n_local)
        feats, split = m.encode_images(px, input_ids=input_ids, split_sizes=[1 + n_local], attention_mask=attn, images_mask=None,
                                       image_sizes=[(iw, ih)], labels=None)
        assert split == [1 + n_local] and len(feats) == 1 and feats[0].dim() == 3 and feats[0].shape[0] == 1
        out = feats[0][0]
        k = f"n{n_local}_"
        kept = int(g[k + "router_rows"][0])
        assert out.shape == (576 + 1 + kept, acfg.hidden_size) and out.dtype == px.dtype
        glob = out[:576]
        want_glob = torch.from_numpy(g[k + "global"])
        assert rel_l2(glob if n_local == 2 else glob[::7, ::5], want_glob) < 5e-5
        assert torch.equal(out[576], emb.weight[1919].detach())                    # the separator row (:219, :254)
        assert rel_l2(out[577:581], torch.from_numpy(g[k + "router_first"])) < 5e-5
        # the stages in between, through the SAME module objects the glue called
        tw = tower(px)
        assert rel_l2(tw if n_local == 2 else tw[:, ::7, ::5], torch.from_numpy(g[k + "tower"])) < 5e-5
        comp = model.sampler.post_qformer(tw[1:])
        assert rel_l2(comp if n_local == 2 else comp[:, ::3, ::5], torch.from_numpy(g[k + "compressed"])) < 5e-5
        nw, nh = A.get_anyres_image_grid_shape((iw, ih), cfg.image_grid_pinpoints, tower.config.image_size)
        assert [nw, nh] == g[k + "grid"].tolist()

    # ---- prepare_inputs_labels_for_multimodal (:274-459) of the reference over the same plugins, two images, right padding
    from oracle import prefill_oracle as P
    px = [W.synthetic_pixels(3, seed=22), W.synthetic_pixels(3, seed=23)]
    ids = torch.randint(1, 1900, (2, 12), generator=torch.Generator().manual_seed(1))
    ids[0, 3] = A.IMAGE_TOKEN_INDEX
    ids[1, 0] = A.IMAGE_TOKEN_INDEX
    am = torch.ones_like(ids)
    am[1, 9:] = 0
    lab = ids.clone()
    sizes = [(336, 336), (336, 336)]
    f_direct, _ = m.encode_images(torch.cat(px, 0), ids, [3, 3], am, None, sizes, labels=lab)
    out = m.prepare_inputs_labels_for_multimodal(ids, None, am, None, lab, px, image_sizes=sizes)
    none_ids, pos, mask, pkv, new_emb, new_lab = out
    assert none_ids is None and pos is None and pkv is None
    want, wl, wm, _ = P.splice(emb.weight.detach(), [f[0] for f in f_direct], ids, am, lab)
    assert new_emb.shape == want.shape and torch.equal(new_emb, want)
    assert torch.equal(new_lab, wl) and torch.equal(mask.to(wm.dtype), wm)

    # ---- load_pretrained_model's tower statements (llava/model/builder.py:161-166); last, because the fp16 cast rounds the weights
    vision_tower = m.get_vision_tower()
    if not vision_tower.is_loaded:
        vision_tower.load_model(device_map="auto")
    vision_tower.load_model()                                    # idempotent (clip_encoder.py:26-28)
    vision_tower.to(device="cpu", dtype=torch.float16)
    image_processor = vision_tower.image_processor
    assert vision_tower.dtype == torch.float16 and image_processor.crop_size["height"] == 336
    out16 = vision_tower(px[0].to(torch.float16))                # forward keeps the INPUT dtype (clip_encoder.py:52,56)
    assert out16.dtype == torch.float16 and out16.shape == (3, 576, W.TINY.hidden_size)


def test_reference_glue_batch_branch_and_plain_projectors(swapped, tmp_path):
    """The no-sampler branches of the reference's encode_images (:257-267) over the swapped builders: identity sampler +
    mlp2x_gelu projector (list branch with split_sizes; plain batch) and the gated projector's [N, 576, D] batch call."""
    A = swapped
    from slime_amd import weights as W
    acfg = W.ADAPTER_TINY
    cfg = _config(_tower_dir(tmp_path), acfg)
    cfg.mm_resampler_type = None
    cfg.mm_projector_type = "mlp2x_gelu"
    m = _reference_model(A, cfg, 64)
    model = m.get_model()
    assert not model.has_sampler
    m.get_vision_tower().load_model()
    asd = W.make_adapter_state_dict(acfg, seed=12)
    model.mm_projector.load_state_dict({k[len("projection."):]: v for k, v in W.sub_state(asd, "mm_projector.").items() if k.startswith("projection.")})
    from oracle import slime_oracle as O
    px = W.synthetic_pixels(3, seed=31)
    tsd = {k: v.float() for k, v in W.strip_tower_prefix(W.make_tower_state_dict(W.TINY, seed=11)).items()}
    want = O.mlp_projector(W.sub_state(asd, "mm_projector."), O.tower_forward(tsd, W.TINY, px))
    ids = torch.zeros((1, 4), dtype=torch.long)
    feats, split = m.encode_images(px, input_ids=ids, split_sizes=[1, 2])
    assert split == [1, 2] and [tuple(f.shape) for f in feats] == [(1, 576, acfg.hidden_size), (2, 576, acfg.hidden_size)]
    assert rel_l2(torch.cat(feats, 0), want) < 5e-5
    feats, split = m.encode_images(px, input_ids=ids)
    assert split is None and rel_l2(feats, want) < 5e-5
    # gated projector, plain batch branch (:261-265): the reference passes text_embedding / attn_mask keywords
    cfg2 = _config(cfg.mm_vision_tower, acfg)
    cfg2.mm_resampler_type = None
    m2 = _reference_model(A, cfg2, 64)
    m2.get_vision_tower().load_model()
    m2.get_model().mm_projector.load_state_dict(W.sub_state(asd, "mm_projector."))
    g = np.load(os.path.join(GOLD, "tiny_stages.npz"))
    feats, _ = m2.encode_images(W.synthetic_pixels(2, seed=31), input_ids=ids, attention_mask=torch.ones_like(ids))
    assert rel_l2(feats[:, ::7, ::5], torch.from_numpy(g["batched_gated"])) < 5e-5
