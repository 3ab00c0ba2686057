"""Multimodal glue for the visual hot path: the build's counterpart of ``LlavaMetaModel`` /
``LlavaMetaForCausalLM.encode_images`` (llava/model/llava_arch.py:28-44,212-269).

Same plugin construction (three builders driven by the same config attribute names), same stage
order and outputs as the reference's ``encode_images`` sampler branch:

    tower(all crops) -> global = mm_projector(crop 0) ; local = mm_projector(post_qformer(crops 1..n))
    -> drop masked crops -> merge ('spatial' raster / 'flat') -> text-guided router -> cat(global, sep, local)

but scheduled for the GPU instead of per image: the reference loops over images in Python and calls the
ViT once per image (llava_arch.py:222); here ALL crops of the batch go through the tower as one batch,
all global views through one GatedBlock launch sequence and all local crops through one
post_qformer + MLP sequence; only the tiny data-dependent tail (merge, router, concat) is per image.
When every image of the batch has the same crop layout (the usual case: one resolution bucket per batch) the
whole adapter -- both experts of the GatedBlock, post_qformer, the shared projection MLP over the stacked rows,
gate mix and spatial merge -- is ONE C-ABI call (``slime_adapter_forward``) on the tower's 16-bit features, as
in the reference, which hands fp16/bf16 features to its adapter; ragged batches take the per-module sequence,
where the tower hands over fp32 features.

``SlimeMetaForCausalLM`` can be mixed into an HF causal LM exactly like ``LlavaMetaForCausalLM``
(INTEGRATION.md); ``SlimeVisualEncoder`` is the standalone form used by bench.py and the tests.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional, Sequence

import torch
import torch.nn as nn

from .multimodal_encoder.builder import build_vision_tower
from .multimodal_projector.builder import build_vision_projector, GatedBlock, HipMlp, _operand_dtype
from .multimodal_resampler.builder import build_vision_sampler
from .. import ops
from ..constants import IMAGE_TOKEN_INDEX, IGNORE_INDEX
from ..mm_utils import get_anyres_image_grid_shape


def _split_indices(split_sizes, device):
    """Row indices of the global views (crop 0 of every image) and of the local crops in the flat crop
    batch (host-side index math, no device sync)."""
    g, l, off = [], [], 0
    for s in split_sizes:
        g.append(off)
        l.extend(range(off + 1, off + s))
        off += s
    return (torch.tensor(g, dtype=torch.long, device=device), torch.tensor(l, dtype=torch.long, device=device))


def _uniform_layout(split_sizes, image_sizes, cfg, crop: int, merge_type: str):
    """(n_local, nw, nh, merge) when all images share one crop count and one grid, else None."""
    if not split_sizes or any(s != split_sizes[0] for s in split_sizes):
        return None
    n_local = split_sizes[0] - 1
    if n_local == 0:
        return (0, 1, 1, False)
    if merge_type == "flat":
        return (n_local, n_local, 1, False)
    if merge_type != "spatial" or image_sizes is None:
        return None
    grids = {tuple(get_anyres_image_grid_shape(sz, cfg.image_grid_pinpoints, crop)) for sz in image_sizes}
    if len(grids) != 1:
        return None
    nw, nh = next(iter(grids))
    return (n_local, nw, nh, True) if nw * nh == n_local else None


def _project_local(projector, comp: torch.Tensor, operand_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """``mm_projector`` on the compressed local crops [sum n_i, g*g, D] of a whole batch.  The reference calls the projector
    per image (n_i <= 7 crops), where GatedBlock.forward's test ``x.shape[0] != 576 and x.shape[1] != 576``
    (projector/builder.py:180-181) always takes the plain-MLP return; on the batch-stacked tensor dim 0 is sum n_i and may
    equal 576 by accident, so the MLP is called directly instead of through that shape test."""
    kw = {"operand_dtype": operand_dtype} if operand_dtype is not None else {}
    if isinstance(projector, GatedBlock):
        return projector.projection(comp, out_dtype=torch.float32, **kw)
    if isinstance(projector, HipMlp):
        return projector(comp, out_dtype=torch.float32, **kw)
    return projector(comp, out_dtype=torch.float32)


def _adapter_operand_dtype(model, images: torch.Tensor) -> dict:
    """``{"operand_dtype": T}`` for the adapter modules that take the extension (GatedBlock, HipMlp, Resampler) when the call's
    images are 16-bit: the reference's tower returns features in the input dtype and its adapter computes in it."""
    if images.dtype not in (torch.bfloat16, torch.float16):
        return {}
    from .multimodal_resampler.sampler import Resampler
    from .multimodal_projector.builder import HipLinear
    return {"operand_dtype": images.dtype} if isinstance(model.mm_projector, (GatedBlock, HipMlp, HipLinear, Resampler)) else {}


def _require_inference(model) -> None:
    """The HIP path has no backward.  The reference's pretrain / finetune stages train mm_projector and sampler through
    this code (train.py:1100-1130); silently detaching them under @torch.no_grad would freeze the adapter without a
    message, so training-mode use is an error here ("inference only")."""
    if not torch.is_grad_enabled():
        return
    for name in ("mm_projector", "sampler"):
        mod = getattr(model, name, None)
        if mod is not None and any(p.requires_grad and p.is_floating_point() for p in mod.parameters()) and mod.training:
            raise RuntimeError(
                f"slime_amd is inference-only: {name} has trainable parameters and autograd is enabled, but the HIP adapter "
                "has no backward.  Call .eval() / .requires_grad_(False), or run under torch.no_grad().")


def _fused_adapter(model, images: torch.Tensor, layout, out_dtype: torch.dtype, out: Optional[torch.Tensor] = None):
    """tower -> slime_adapter_forward for a uniform batch: tokens [B, 576 + n*g*g, H]."""
    n_local, nw, nh, merge = layout
    tower = model.get_vision_tower()
    T = _operand_dtype(images, model.mm_projector.projection[0].weight)
    feats = tower(images, out_dtype=T)
    B = feats.shape[0] // (1 + n_local)
    post = model.sampler.post_qformer.packed(feats.shape[1], T) if n_local else None
    return ops.adapter_forward(model.mm_projector.packed(T), post, feats, B, n_local, nw, nh, merge,
                               int(model.mm_projector.learnable_gated), out_dtype, out)


def _check_token_ids(t, vocab_size):
    """nn.Embedding raises IndexError for an id outside [0, vocab); the splice would instead read id -1 as padding and an id
    <= -2 as an image-feature row.  IMAGE_TOKEN_INDEX is the one legal negative id."""
    import numpy as np
    bad = (t != IMAGE_TOKEN_INDEX) & ((t < 0) | (t >= vocab_size))
    if bad.any():
        raise IndexError(f"input_ids contain token id {int(t[np.argmax(bad)])} outside [0, {vocab_size}) (index out of range in self)")


def splice_plan(input_ids, attention_mask, labels, feat_lens, max_length=None, padding_side: str = "right", vocab_size=None):
    """Index plan of ``prepare_inputs_labels_for_multimodal`` (llava_arch.py:362-459), integer host logic on numpy arrays.
    ``input_ids`` [B, L] int64 (IMAGE_TOKEN_INDEX marks an image), ``attention_mask`` [B, L] or None, ``labels`` [B, L] or
    None, ``feat_lens[j]`` = token rows of image feature j (consumed in order of appearance; a sequence without an image
    token still consumes one feature, of which it uses zero rows: :377-385).
    Returns int64 arrays [B, max_len]: ``src`` (>= 0 token id, -2 - k row k of the concatenated features, -1 padding),
    ``labels`` (IGNORE_INDEX on image and padding rows), ``mask``, ``position_ids``.
    ``vocab_size``: validate the ids that survive the mask (the only ones the reference embeds, :361-373) as nn.Embedding would."""
    import numpy as np
    ids = np.asarray(input_ids, dtype=np.int64)
    B = ids.shape[0]
    am = np.ones_like(ids, dtype=bool) if attention_mask is None else np.asarray(attention_mask) != 0
    lb = np.full_like(ids, IGNORE_INDEX) if labels is None else np.asarray(labels, dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(np.asarray(feat_lens, dtype=np.int64))])
    seqs, labs, img = [], [], 0
    for b in range(B):
        t, y = ids[b][am[b]], lb[b][am[b]]                                  # :366-368 drop padding through the mask
        if vocab_size is not None:
            _check_token_ids(t, vocab_size)
        where = np.nonzero(t == IMAGE_TOKEN_INDEX)[0]
        if where.size == 0:                                                 # :376-385
            if img >= len(feat_lens):
                raise ValueError("fewer image features than sequences / image tokens")
            seqs.append(t); labs.append(y); img += 1
            continue
        if img + where.size > len(feat_lens):
            raise ValueError("fewer image features than image tokens")
        ps, pl, prev = [], [], 0
        for w in where:                                                     # :387-411
            ps.append(t[prev:w]); pl.append(y[prev:w])
            n = int(feat_lens[img])
            ps.append(-2 - (starts[img] + np.arange(n, dtype=np.int64)))
            pl.append(np.full(n, IGNORE_INDEX, dtype=np.int64))
            img += 1
            prev = w + 1
        ps.append(t[prev:]); pl.append(y[prev:])
        seqs.append(np.concatenate(ps)); labs.append(np.concatenate(pl))
    if max_length is not None:                                              # :420-424
        seqs = [q[:max_length] for q in seqs]
        labs = [q[:max_length] for q in labs]
    max_len = max(q.shape[0] for q in seqs)                                 # :427
    src = np.full((B, max_len), -1, dtype=np.int64)
    lab = np.full((B, max_len), IGNORE_INDEX, dtype=np.int64)
    mask = np.zeros((B, max_len), dtype=np.int64)
    pos = np.zeros((B, max_len), dtype=np.int64)
    for b, (q, y) in enumerate(zip(seqs, labs)):                            # :435-455
        n = q.shape[0]
        sl = slice(max_len - n, max_len) if padding_side == "left" else slice(0, n)
        src[b, sl], lab[b, sl], mask[b, sl], pos[b, sl] = q, y, 1, np.arange(n)
    return src, lab, mask, pos


class SlimeMetaModel:
    """Mixin: builds the three hot-path plugins from the config (llava_arch.py:28-44)."""

    def __init__(self, config):
        super(SlimeMetaModel, self).__init__(config)
        if hasattr(config, "mm_vision_tower"):
            self.vision_tower = build_vision_tower(config, delay_load=True)
            self.mm_projector = build_vision_projector(config)
            self.sampler = build_vision_sampler(config)
            t = getattr(config, "mm_resampler_type", None)
            self.has_sampler = t != "identity" and t is not None and t != "spatial"

    def get_vision_tower(self):
        vt = getattr(self, "vision_tower", None)
        return vt[0] if type(vt) is list else vt


class SlimeMetaForCausalLM(ABC):
    @abstractmethod
    def get_model(self):
        ...

    def get_vision_tower(self):
        return self.get_model().get_vision_tower()

    # ------------------------------------------------------------------ text side (router input)
    def get_pure_text_embedding(self, input_ids, attention_mask=None, labels=None):
        """Embeddings of the text tokens with the <image> placeholders removed, zero rows appended
        (or prepended for left padding) in their place (llava_arch.py:162-210)."""
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        embed = self.get_model().embed_tokens
        left = getattr(self.config, "tokenizer_padding_side", "right") == "left"
        embs, masks = [], []
        for ids, am in zip(input_ids, attention_mask):
            keep = ids != IMAGE_TOKEN_INDEX
            n_img = int((~keep).sum())
            e, m = embed(ids[keep]), am[keep]
            if n_img > 0:
                ze = torch.zeros((n_img, e.shape[1]), dtype=e.dtype, device=e.device)
                zm = torch.zeros((n_img,), dtype=m.dtype, device=m.device)
                e, m = (torch.cat((ze, e)), torch.cat((zm, m))) if left else (torch.cat((e, ze)), torch.cat((m, zm)))
            embs.append(e)
            masks.append(m)
        max_len = getattr(self.config, "tokenizer_model_max_length", None)
        if max_len is not None:
            embs, masks = [x[:max_len] for x in embs], [x[:max_len] for x in masks]
        embs, masks = torch.stack(embs, 0), torch.stack(masks, 0)
        assert masks.shape == embs.shape[:2]
        return embs, masks

    # ------------------------------------------------------------------ the hot path
    def encode_images(self, images, input_ids=None, split_sizes=None, attention_mask=None, images_mask=None,
                      image_sizes=None, labels=None):
        _require_inference(self.get_model())
        with torch.no_grad():
            return self._encode_images(images, input_ids, split_sizes, attention_mask, images_mask, image_sizes, labels)

    def _encode_images(self, images, input_ids, split_sizes, attention_mask, images_mask, image_sizes, labels):
        model = self.get_model()
        tower = model.get_vision_tower()
        cfg = self.config
        merge_type = getattr(cfg, "mm_patch_merge_type", "flat")
        use_local_only = getattr(cfg, "use_local_only", False)
        use_global_only = getattr(cfg, "use_global_only", False)
        out_dtype = images.dtype

        if model.has_sampler and split_sizes is not None:
            sep = model.embed_tokens(torch.tensor(cfg.seperator, dtype=input_ids.dtype, device=input_ids.device))
            text_emb, text_mask = self.get_pure_text_embedding(input_ids, attention_mask, labels)
            B = len(split_sizes)
            layout = None
            if (getattr(cfg, "fused_adapter", True) and isinstance(model.mm_projector, GatedBlock) and images_mask is None
                    and not use_local_only and not use_global_only):
                layout = _uniform_layout(list(split_sizes), image_sizes, cfg, tower.config.image_size, merge_type)
            if layout is not None and layout[0] > 0:
                tokens = _fused_adapter(model, images, layout, torch.float32)        # [B, 576 + n*g*g, H] fp32
                P = tower.num_patches
                sep32 = sep.to(device=tokens.device, dtype=torch.float32).unsqueeze(0)
                rows = tokens.shape[1]
                # router: all B images in one launch pair, one D2H for the B kept counts (the reference syncs per image)
                keeps = model.sampler.select_batched(tokens.view(B * rows, -1), [i * rows + P for i in range(B)], [rows - P] * B,
                                                     text_emb, text_mask)
                outs = []
                for i in range(B):
                    routed = tokens[i, P:].index_select(0, keeps[i])
                    outs.append(torch.cat([tokens[i, :P], sep32, routed], dim=0).to(out_dtype).unsqueeze(0))
                return outs, split_sizes
            feats = tower(images, out_dtype=torch.float32)                      # [sum(1+n_i), 576, D], one batch
            dev = feats.device
            g_idx, l_idx = _split_indices(split_sizes, dev)
            glob = loc = None
            # the adapter's MFMA operands are of the IMAGES' 16-bit type, as in the reference (its tower hands the adapter features
            # in the input dtype, clip_encoder.py:52,56) -- not of a default picked from the fp32 hand-over (round 6: an fp16 call
            # into fp32 adapter parameters ran the adapter on bf16 operands)
            T16 = _adapter_operand_dtype(model, images)
            if not use_local_only:
                glob = model.mm_projector(feats.index_select(0, g_idx), out_dtype=torch.float32, **T16)      # [B,576,H]
            if not use_global_only:
                comp = model.sampler.post_qformer(feats.index_select(0, l_idx), out_dtype=torch.float32, **T16)   # [sum n_i,144,D]
                loc = _project_local(model.mm_projector, comp, T16.get("operand_dtype"))         # [sum n_i,144,H]
            g = model.sampler.grid_size
            merged_list = []
            lstart = 0
            for i in range(B):
                if loc is None:
                    break
                n_i = split_sizes[i] - 1
                li = loc[lstart:lstart + n_i]
                lstart += n_i
                if images_mask is not None and images_mask[0].size(0) - 1 == li.size(0):
                    li = li[torch.nonzero(images_mask[i][1:]).squeeze(1)]    # padded crops (train.py:903-926)
                H = li.shape[-1]
                merged = torch.empty((li.shape[0] * g * g, H), dtype=torch.float32, device=dev)
                if li.shape[0] > 0:
                    if merge_type == "spatial":
                        nw, nh = get_anyres_image_grid_shape(image_sizes[i], cfg.image_grid_pinpoints, tower.config.image_size)
                        if nw * nh != li.shape[0]:
                            raise ValueError(f"image {i}: grid {nw}x{nh} does not match {li.shape[0]} local crops")
                        ops.merge_rows(li.contiguous(), merged, 0, nw, nh, g, True)
                    elif merge_type == "flat":
                        ops.merge_rows(li.contiguous(), merged, 0, li.shape[0], 1, g, False)
                    else:
                        raise NotImplementedError(f"mm_patch_merge_type={merge_type!r}")
                merged_list.append(merged)
            keeps = None
            if merged_list:
                # one ragged concatenation -> one batched router call (one D2H for all kept counts)
                cat = torch.cat(merged_list, 0)
                offs, o = [], 0
                for m_ in merged_list:
                    offs.append(o)
                    o += m_.shape[0]
                keeps = model.sampler.select_batched(cat, offs, [m_.shape[0] for m_ in merged_list], text_emb, text_mask)
            outs = []
            for i in range(B):
                pieces = []
                if glob is not None:
                    pieces.append(glob[i])
                if loc is not None:
                    if glob is not None:
                        pieces.append(sep.to(device=dev, dtype=torch.float32).unsqueeze(0))
                    pieces.append(merged_list[i].index_select(0, keeps[i]))
                outs.append(torch.cat(pieces, dim=0).to(out_dtype).unsqueeze(0))
            return outs, split_sizes

        T16 = _adapter_operand_dtype(model, images)
        if split_sizes is not None:                                               # no sampler: per-image lists
            feats = tower(images, out_dtype=torch.float32)
            proj = model.mm_projector(feats, out_dtype=out_dtype, **T16) if not isinstance(model.mm_projector, GatedBlock) \
                else torch.cat([model.mm_projector(f, out_dtype=out_dtype, **T16) for f in torch.split(feats, 1)], 0)
            return list(torch.split(proj, split_sizes, dim=0)), split_sizes

        feats = tower(images, out_dtype=torch.float32)
        return model.mm_projector(feats, out_dtype=out_dtype, **T16), split_sizes


    # ------------------------------------------------------------------ splice (the step after the hot path)
    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, images,
                                             image_sizes=None, images_mask=None):
        """Signature, branches and return tuple of llava_arch.py:274-459.  The visual features come from ``encode_images``
        above; the splice itself is an integer plan on the host (``splice_plan``: token ids are a few hundred ints) and ONE
        device launch (``slime_splice_rows``) that writes the whole padded [B, max_len, H] embedding tensor -- embedding
        lookup, image rows, zero padding -- where the reference runs per-sequence embed / split / cat / stack."""
        vision_tower = self.get_vision_tower()
        if vision_tower is None or images is None or input_ids.shape[1] == 1:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels
        merge_type = getattr(self.config, "mm_patch_merge_type", "flat")
        # The ids come to the host once (the splice plan below is integer host logic).  A token id outside the embedding table is
        # rejected HERE, before anything touches the device: the reference's nn.Embedding lookups would hit a device-side assert,
        # which on ROCm aborts the process instead of raising.  Scope as in the reference: the splice embeds only the ids under the
        # attention mask (llava_arch.py:361-373: a pad id past the table is legal there) -- splice_plan checks those --, the
        # text-guided router embeds EVERY non-image id, padding included (get_pure_text_embedding, :162-179).
        ids_np = input_ids.detach().cpu().numpy()
        vocab = self.get_model().embed_tokens.weight.shape[0]
        if getattr(self.get_model(), "has_sampler", False) and ids_np.size:
            _check_token_ids(ids_np.reshape(-1), vocab)
        if type(images) is list or images.ndim == 5:
            if type(images) is list:
                images = [x.unsqueeze(0) if x.ndim == 3 else x for x in images]
            concat_images = torch.cat([im for im in images], dim=0)
            split_sizes = [im.shape[0] for im in images]
            feats, split_sizes = self.encode_images(concat_images, input_ids, split_sizes, attention_mask, images_mask, image_sizes,
                                                    labels=labels)
            if type(feats) is not list:
                feats = list(torch.split(feats, split_sizes, dim=0))
            flat = []
            for f in feats:
                if f.dim() == 3 and f.shape[0] == 1:                      # sampler branch: [1, 576 + 1 + k, H] (:254-255,322)
                    flat.append(f[0])
                elif merge_type == "flat":                                  # :290-291
                    flat.append(f.flatten(0, 1))
                elif merge_type.startswith("spatial"):
                    raise NotImplementedError("LLaVA-NeXT style spatial merge of un-sampled crops (llava_arch.py:293-330) is outside "
                                              "the SliME hot path: SliME merges inside encode_images (sampler branch)")
                else:
                    raise ValueError(f"Unexpected mm_patch_merge_type: {merge_type}")
            feats = flat
        else:
            f, _ = self.encode_images(images, input_ids=input_ids, attention_mask=attention_mask, labels=labels)
            feats = [x for x in f]
        if getattr(self.config, "tune_mm_mlp_adapter", False) and getattr(self.config, "mm_use_im_start_end", False):
            raise NotImplementedError

        embed = self.get_model().embed_tokens
        table = embed.weight
        dev = table.device
        src, lab, mask, pos = splice_plan(ids_np,
                                          None if attention_mask is None else attention_mask.detach().cpu().numpy(),
                                          None if labels is None else labels.detach().cpu().numpy(),
                                          [f.shape[0] for f in feats], getattr(self.config, "tokenizer_model_max_length", None),
                                          getattr(self.config, "tokenizer_padding_side", "right"), vocab_size=vocab)
        B, T = src.shape
        allf = torch.cat([f.reshape(-1, f.shape[-1]) for f in feats], 0).to(dev) if feats else None
        out_dtype = table.dtype if table.dtype in (torch.float32, torch.bfloat16, torch.float16) else torch.float32
        new_input_embeds = ops.splice_rows(table.detach(), allf, torch.from_numpy(src).reshape(-1).to(dev), out_dtype).view(B, T, -1)
        new_labels = None if labels is None else torch.from_numpy(lab).to(labels.device, labels.dtype)
        new_mask = None if attention_mask is None else torch.from_numpy(mask).to(attention_mask.device, attention_mask.dtype)
        new_pos = None if position_ids is None else torch.from_numpy(pos).to(position_ids.device, position_ids.dtype)
        return None, new_pos, new_mask, past_key_values, new_input_embeds, new_labels


class _Cfg:
    """Attribute bag with the config names the reference persists (llava_arch.py:64-93)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def default_slime_config(vision_tower: str = "synthetic:1234", hidden_size: int = 4096, mm_hidden_size: int = 1024,
                         **over) -> _Cfg:
    """SliME-8B inference configuration (scripts/llama/llama3_8b_sft.sh:14-48)."""
    d = dict(mm_vision_tower=vision_tower, mm_vision_select_layer=-2, mm_vision_select_feature="patch",
             mm_projector_type="gated", mm_hidden_size=mm_hidden_size, hidden_size=hidden_size,
             mm_patch_merge_type="spatial", image_aspect_ratio="anyres", mm_resampler_type="cosine",
             mm_resampler_topp=0.95, mm_resampler_dim=144, mm_resampler_temp=1.0, mm_learnable_gated=-1,
             image_grid_pinpoints="[(336, 672), (672, 336), (672, 672), (1008, 336), (336, 1008)]",
             seperator=1919, pad_token_id=0, use_local_only=False, use_global_only=False)
    d.update(over)
    return _Cfg(**d)


class _ModelBase(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config


class _VisualModel(SlimeMetaModel, _ModelBase):
    def __init__(self, config, embed_tokens: Optional[nn.Module] = None):
        super().__init__(config)
        self.embed_tokens = embed_tokens


class SlimeVisualEncoder(nn.Module, SlimeMetaForCausalLM):
    """Standalone visual front end: tower + adapter + glue, no LLM.  ``embed_tokens`` (the LLM's
    embedding table) is only needed for the separator token and the text-guided router."""

    def __init__(self, config, embed_tokens: Optional[nn.Module] = None):
        super().__init__()
        self.config = config
        self.model = _VisualModel(config, embed_tokens)
        self.eval()                     # inference front end (a from_pretrained HF model arrives in eval mode as well)

    def get_model(self):
        return self.model

    def load_visual_state(self, tower_sd=None, adapter_sd=None):
        vt = self.get_vision_tower()
        if not vt.is_loaded:
            vt.load_model()
        if tower_sd is not None:
            from .multimodal_encoder.clip_encoder import check_tower_keys
            check_tower_keys(vt.vision_tower.load_state_dict(tower_sd, strict=False), "<state dict>")
        if adapter_sd is not None:
            from ..weights import sub_state
            self.model.mm_projector.load_state_dict(sub_state(adapter_sd, "mm_projector."), strict=True)
            if len(self.model.sampler.state_dict()) > 0:          # IdentityMap (no sampler) has no parameters
                self.model.sampler.load_state_dict(sub_state(adapter_sd, "sampler."), strict=True)
        return self

    @torch.no_grad()
    def encode_visual(self, images: torch.Tensor, split_sizes: Sequence[int], image_sizes=None, merge: Optional[str] = None):
        """Router-free form of the sampler branch (what bench.py times): returns per image
        ``(global [576,H], merged_local [n*144,H])`` in fp32."""
        model = self.model
        tower = model.get_vision_tower()
        merge = merge or getattr(self.config, "mm_patch_merge_type", "flat")
        if getattr(self.config, "fused_adapter", True) and isinstance(model.mm_projector, GatedBlock):
            layout = _uniform_layout(list(split_sizes), image_sizes, self.config, tower.config.image_size, merge)
            if layout is not None and layout[0] > 0:
                tokens = _fused_adapter(model, images, layout, torch.float32)
                P = tower.num_patches
                return [(tokens[i, :P], tokens[i, P:]) for i in range(len(split_sizes))]
        feats = tower(images, out_dtype=torch.float32)
        dev = feats.device
        g_idx, l_idx = _split_indices(split_sizes, dev)
        T16 = _adapter_operand_dtype(model, images)
        glob = model.mm_projector(feats.index_select(0, g_idx), out_dtype=torch.float32, **T16)
        n_local = feats.shape[0] - len(split_sizes)
        if n_local == 0:
            return [(glob[i], glob.new_zeros((0, glob.shape[-1]))) for i in range(len(split_sizes))]
        comp = model.sampler.post_qformer(feats.index_select(0, l_idx), out_dtype=torch.float32, **T16)
        loc = _project_local(model.mm_projector, comp, T16.get("operand_dtype"))
        g = model.sampler.grid_size
        outs, lstart = [], 0
        for i, s in enumerate(split_sizes):
            n_i = s - 1
            li = loc[lstart:lstart + n_i].contiguous()
            lstart += n_i
            merged = torch.empty((n_i * g * g, li.shape[-1]), dtype=torch.float32, device=dev)
            if n_i > 0:
                if merge == "spatial":
                    nw, nh = get_anyres_image_grid_shape(image_sizes[i], self.config.image_grid_pinpoints,
                                                         tower.config.image_size)
                    ops.merge_rows(li, merged, 0, nw, nh, g, True)
                else:
                    ops.merge_rows(li, merged, 0, n_i, 1, g, False)
            outs.append((glob[i], merged))
        return outs
