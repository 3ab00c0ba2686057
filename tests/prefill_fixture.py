"""Shared loader for tests/golden/prefill.npz: rebuilds the seeded inputs the reference was run
on in oracle/make_golden_prefill.py and checks the regeneration against the probes stored next to the outputs."""
import os

import numpy as np
import torch

from conftest import GOLDEN


def load():
    return np.load(os.path.join(GOLDEN, "prefill.npz"))


def splice_case(g, name):
    k = f"splice_{name}_"
    has_mask = bool(g[k + "has_mask"][0])
    feats = [torch.from_numpy(g[k + f"feat{j}"]) for j in range(len(g[k + "feat_lens"]))]
    ml = int(g[k + "max_length"][0])
    return dict(
        table=torch.from_numpy(g["splice_table"]), feats=feats, input_ids=torch.from_numpy(g[k + "input_ids"]),
        attention_mask=torch.from_numpy(g[k + "attention_mask"]) if has_mask else None,
        labels=torch.from_numpy(g[k + "labels"]) if has_mask else None,
        max_length=None if ml < 0 else ml, padding_side="left" if bool(g[k + "left"][0]) else "right",
        out_embeds=torch.from_numpy(g[k + "out_embeds"]),
        out_mask=torch.from_numpy(g[k + "out_mask"]) if has_mask else None,
        out_labels=torch.from_numpy(g[k + "out_labels"]) if has_mask else None,
        out_position_ids=torch.from_numpy(g[k + "out_position_ids"]) if has_mask else None)


def llama_inputs(g):
    """(dims, weights dict, hidden [B,S,D]) regenerated with the generator's draw order (q, k, v, o weights, then hidden)."""
    D, HQ, HKV, S, B = (int(x) for x in g["llama_dims"])
    gen = torch.Generator().manual_seed(int(g["llama_seed"][0]))
    shapes = {"q_proj": (HQ * 128, D), "k_proj": (HKV * 128, D), "v_proj": (HKV * 128, D), "o_proj": (D, HQ * 128)}
    w = {n: torch.randn(s, generator=gen) * (s[1] ** -0.5) for n, s in shapes.items()}
    hidden = torch.randn(B, S, D, generator=gen)
    assert np.array_equal(w["q_proj"].numpy()[::97, ::61], g["llama_wq_probe"]), "weight regeneration drifted"
    assert np.array_equal(hidden.numpy()[:, ::41, ::53], g["llama_hidden_probe"]), "input regeneration drifted"
    return (D, HQ, HKV, S, B), w, hidden
