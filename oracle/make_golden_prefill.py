#!/usr/bin/env python3
"""Generate tests/golden/prefill.npz by running the REFERENCE (build container only).

TEST INFRASTRUCTURE.  Run:  python oracle/make_golden_prefill.py   (needs /root/reference; CPU; seconds)

  (1) splice: the reference's own ``LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal``
      (llava/model/llava_arch.py:274-459) is executed on seeded token ids / embedding table / image features.  The class is
      instantiated through a minimal host object that supplies what the method reads (``get_model().embed_tokens``,
      ``get_vision_tower()``, ``config``, ``device``) and whose ``encode_images`` returns the seeded per-image features in the
      sampler branch's format (list of [1, T_i, H]; llava_arch.py:254-255) -- the splice is what is pinned here, the encode path
      has its own fixtures.
  (2) Llama attention: the reference's prefill attention is HF ``LlamaAttention`` with its forward replaced by
      llava/train/llama_flash_attn_monkey_patch.py:16-93 (flash-attn, absent offline: the patch cannot be imported).  The
      patch computes what the module it replaces computes -- RoPE, GQA, causal softmax attention over the un-padded tokens,
      o_proj -- so the vectors come from the installed HF ``LlamaAttention`` (eager, fp32) at a small Llama-3-shaped config
      (head_dim 128, 4 query heads per kv head, rope_theta 5e5), with no padding, right padding and left padding; rows at
      padded positions are not recorded (the patch zero-fills them, HF leaves them unspecified).
"""
from __future__ import annotations

import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
from make_golden import import_reference, OUT          # noqa: E402

IMG = -200


def splice_cases():
    """(name, input_ids, attention_mask | None, labels | None, feat_lens, max_length, padding_side)."""
    g = torch.Generator().manual_seed(77)

    def ids(n):
        return torch.randint(1, 50, (n,), generator=g).tolist()

    cases = []
    # A: right padding, masks with holes at the end, labels, one image per sequence in different places
    a = [ids(3) + [IMG] + ids(5) + [0, 0], [IMG] + ids(9), ids(6) + [IMG] + [0, 0, 0]]
    am = [[1] * 9 + [0, 0], [1] * 10 + [0], [1] * 7 + [0, 0, 0]]
    L = max(len(x) for x in a)
    a = [x + [0] * (L - len(x)) for x in a]
    am = [x + [0] * (L - len(x)) for x in am]
    lab = [[(t if t > 0 else -100) for t in row] for row in a]
    cases.append(("A", a, am, lab, [7, 5, 9], None, "right"))
    # B: left padding side, a sequence without image (still consumes a feature), two images in one, truncation
    b = [[0, 0] + ids(4) + [IMG] + ids(2) + [IMG] + ids(3), [0] * 5 + ids(8), [0, 0, 0] + [IMG] + ids(9)]
    bm = [[0, 0] + [1] * 11, [0] * 5 + [1] * 8, [0, 0, 0] + [1] * 10]
    blab = [[(t if t > 0 else -100) for t in row] for row in b]
    cases.append(("B", b, bm, blab, [6, 4, 3, 8], 17, "left"))
    # C: no mask / labels / position ids given -> the reference returns None for them
    c = [ids(2) + [IMG] + ids(4), ids(1) + [IMG] + ids(5)]
    cases.append(("C", c, None, None, [5, 5], None, "right"))
    return cases


def run_splice(rec):
    from llava.model.llava_arch import LlavaMetaForCausalLM
    H, V = 16, 64
    table = torch.randn(V, H, generator=torch.Generator().manual_seed(5))
    rec["splice_table"] = table.numpy()

    class Host(LlavaMetaForCausalLM):
        def __init__(self, feats, cfg):
            self._feats = feats
            self.config = cfg
            self.device = torch.device("cpu")
            emb = nn.Embedding(V, H)
            emb.weight.data.copy_(table)
            self._model = SimpleNamespace(embed_tokens=emb, get_vision_tower=lambda: object())

        def get_model(self):
            return self._model

        def encode_images(self, images, input_ids=None, split_sizes=None, attention_mask=None, images_mask=None,
                          image_sizes=None, labels=None):
            return [f.unsqueeze(0) for f in self._feats], split_sizes

    for name, ids, am, lab, flens, max_len, side in splice_cases():
        fg = torch.Generator().manual_seed(1000 + len(flens))
        feats = [torch.randn(n, H, generator=fg) for n in flens]
        cfg = SimpleNamespace(mm_patch_merge_type="spatial", image_aspect_ratio="anyres", tokenizer_padding_side=side,
                              image_grid_pinpoints="[(336, 672)]")
        if max_len is not None:
            cfg.tokenizer_model_max_length = max_len
        host = Host(feats, cfg)
        t_ids = torch.tensor(ids, dtype=torch.long)
        t_am = None if am is None else torch.tensor(am, dtype=torch.long)
        t_lab = None if lab is None else torch.tensor(lab, dtype=torch.long)
        images = [torch.zeros(1, 3, 2, 2) for _ in flens]
        out = host.prepare_inputs_labels_for_multimodal(t_ids, None if am is None else torch.arange(t_ids.shape[1])[None].expand_as(t_ids),
                                                        t_am, None, t_lab, images, image_sizes=[(336, 336)] * len(flens))
        none_ids, pos, mask, pkv, emb, labels = out
        assert none_ids is None and pkv is None
        k = f"splice_{name}_"
        rec[k + "input_ids"] = t_ids.numpy()
        rec[k + "has_mask"] = np.array([am is not None])
        if am is not None:
            rec[k + "attention_mask"] = t_am.numpy()
            rec[k + "labels"] = t_lab.numpy()
            rec[k + "out_mask"] = mask.numpy()
            rec[k + "out_labels"] = labels.numpy()
            rec[k + "out_position_ids"] = pos.numpy()
        else:
            assert pos is None and mask is None and labels is None
        rec[k + "feat_lens"] = np.array(flens, dtype=np.int64)
        for j, f in enumerate(feats):
            rec[k + f"feat{j}"] = f.numpy()
        rec[k + "max_length"] = np.array([-1 if max_len is None else max_len], dtype=np.int64)
        rec[k + "left"] = np.array([side == "left"])
        rec[k + "out_embeds"] = emb.numpy()
        print(f"splice {name}: embeds {tuple(emb.shape)}")


def run_llama_attention(rec):
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaAttention, LlamaRotaryEmbedding
    D, HQ, HKV, S, B = 1024, 8, 2, 333, 3
    cfg = LlamaConfig(hidden_size=D, num_attention_heads=HQ, num_key_value_heads=HKV, head_dim=128, intermediate_size=256,
                      num_hidden_layers=1, rope_theta=500000.0, max_position_embeddings=8192, attention_bias=False,
                      vocab_size=32)
    cfg._attn_implementation = "eager"
    attn = LlamaAttention(cfg, 0).eval()
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for p in attn.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (p.shape[1] ** -0.5))
    rope = LlamaRotaryEmbedding(cfg)
    hidden = torch.randn(B, S, D, generator=g)
    rec["llama_dims"] = np.array([D, HQ, HKV, S, B], dtype=np.int64)
    rec["llama_seed"] = np.array([9], dtype=np.int64)
    rec["llama_theta"] = np.array([500000.0])
    # the test regenerates weights and hidden states from the seed with the same draw order; checks of that regeneration:
    rec["llama_wq_probe"] = attn.q_proj.weight.detach().numpy()[::97, ::61]
    rec["llama_hidden_probe"] = hidden.numpy()[:, ::41, ::53]
    lens = {"nopad": [S, S, S], "right": [S, 201, 77], "left": [150, S, 290]}
    for mode, ln in lens.items():
        mask = torch.zeros(B, S, dtype=torch.long)
        pos = torch.zeros(B, S, dtype=torch.long)
        for b, n in enumerate(ln):
            sl = slice(S - n, S) if mode == "left" else slice(0, n)
            mask[b, sl] = 1
            pos[b, sl] = torch.arange(n)
        cos, sin = rope(hidden, pos)
        causal = torch.ones(S, S, dtype=torch.bool).tril()
        allowed = causal[None] & mask[:, None, :].bool()
        add = torch.zeros(B, 1, S, S).masked_fill(~allowed[:, None], torch.finfo(torch.float32).min)
        with torch.no_grad():
            out, _ = attn(hidden, position_embeddings=(cos, sin), attention_mask=add)
        out = out * mask[..., None]                                   # padded rows: unspecified in HF, zero in the patch
        rec[f"llama_{mode}_mask"] = mask.numpy()
        rec[f"llama_{mode}_pos"] = pos.numpy()
        rec[f"llama_{mode}_out"] = out.numpy()[:, ::3, ::7]
        rec[f"llama_{mode}_norm"] = out.double().norm(dim=-1).numpy()
        print(f"llama attention {mode}: out {tuple(out.shape)}")


def main():
    torch.set_grad_enabled(False)
    import_reference()
    rec = {}
    run_splice(rec)
    run_llama_attention(rec)
    path = os.path.join(OUT, "prefill.npz")
    np.savez(path, **rec)
    print("prefill.npz bytes:", os.path.getsize(path))


if __name__ == "__main__":
    main()
