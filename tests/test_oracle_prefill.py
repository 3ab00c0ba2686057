"""Pin oracle/prefill_oracle.py (splice plan, RoPE, Llama GQA causal attention) against tests/golden/prefill.npz, which holds
outputs of the reference's prepare_inputs_labels_for_multimodal and of HF LlamaAttention (oracle/make_golden_prefill.py).
CPU only."""
import pytest
import torch

from oracle import prefill_oracle as P
from conftest import rel_l2
import prefill_fixture as F


@pytest.mark.parametrize("name", ["A", "B", "C"])
def test_splice_matches_reference(name):
    c = F.splice_case(F.load(), name)
    emb, lab, mask, pos = P.splice(c["table"], c["feats"], c["input_ids"], c["attention_mask"], c["labels"],
                                   c["max_length"], c["padding_side"])
    assert emb.shape == c["out_embeds"].shape
    assert torch.equal(emb, c["out_embeds"]), "splice is pure data movement: bit-exact"
    if c["out_mask"] is not None:
        assert torch.equal(mask, c["out_mask"]) and torch.equal(lab, c["out_labels"]) and torch.equal(pos, c["out_position_ids"])


def test_splice_plan_edge_cases():
    # a sequence without <image> still consumes one feature (llava_arch.py:377-385); truncation; all-padding row
    src, lab, mask, pos = P.splice_plan([[5, 6, 7], [P.IMAGE_TOKEN_INDEX, 9, 0]], [[1, 1, 1], [1, 1, 0]], None, [4, 2], 3, "right")
    assert src[0] == [5, 6, 7] and src[1] == [P.feat_row(4), P.feat_row(5), 9]
    assert mask == [[1, 1, 1], [1, 1, 1]] and pos[1] == [0, 1, 2] and lab[1] == [-100, -100, -100]
    src, lab, mask, pos = P.splice_plan([[5, 6], [0, 0]], [[1, 1], [0, 0]], [[1, 2], [3, 4]], [0, 0], None, "left")
    assert src == [[5, 6], [P.PAD_ROW, P.PAD_ROW]] and mask == [[1, 1], [0, 0]] and lab[1] == [-100, -100]


@pytest.mark.parametrize("mode", ["nopad", "right", "left"])
def test_llama_attention_matches_hf(mode):
    g = F.load()
    (D, HQ, HKV, S, B), w, hidden = F.llama_inputs(g)
    mask = torch.from_numpy(g[f"llama_{mode}_mask"])
    pos = torch.from_numpy(g[f"llama_{mode}_pos"])
    out = P.llama_attention_forward(hidden, w["q_proj"], w["k_proj"], w["v_proj"], w["o_proj"], HQ, HKV, pos,
                                    None if mode == "nopad" else mask, float(g["llama_theta"][0]))
    assert rel_l2(out[:, ::3, ::7], g[f"llama_{mode}_out"]) < 2e-5
    assert rel_l2(out.double().norm(dim=-1), g[f"llama_{mode}_norm"]) < 2e-5
    assert float(out[mask == 0].abs().max() if (mask == 0).any() else 0.0) == 0.0     # pad_input zero rows
