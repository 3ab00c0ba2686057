"""ingestion of real checkpoint FILES (written here with safetensors / torch.save -- no reference
code): an HF CLIP directory through build_vision_tower(<abs path>), both key generations, and the adapter files
mm_projector.bin / sampler.bin / non_lora_trainables.bin with the prefixes llava/model/builder.py:64-109 and
llava/model/llava_arch.py:107-119 handle.  CPU tests check the loaded tensors; the -m gpu tests run the loaded model."""
import json
import os
from types import SimpleNamespace

import pytest
import torch

from conftest import rel_l2


def _write_hf_dir(tmp, sd, cfg, nested_config: bool, extra=None):
    from safetensors.torch import save_file
    v = dict(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
             num_attention_heads=cfg.num_attention_heads, image_size=cfg.image_size, patch_size=cfg.patch_size,
             layer_norm_eps=cfg.layer_norm_eps, hidden_act="quick_gelu")
    json.dump({"model_type": "clip", "vision_config": v, "text_config": {"hidden_size": 8}} if nested_config else
              dict(v, model_type="clip_vision_model"), open(os.path.join(tmp, "config.json"), "w"))
    tensors = {k: t.contiguous() for k, t in sd.items()}
    tensors.update(extra or {})
    save_file(tensors, os.path.join(tmp, "model.safetensors"))
    json.dump({"crop_size": {"height": 336, "width": 336}, "size": {"shortest_edge": 336}, "do_normalize": True,
               "image_mean": [0.48145466, 0.4578275, 0.40821073], "image_std": [0.26862954, 0.26130258, 0.27577711],
               "rescale_factor": 1 / 255, "resample": 3}, open(os.path.join(tmp, "preprocessor_config.json"), "w"))


def _tower_from_dir(path):
    from slime_amd.model.multimodal_encoder.builder import build_vision_tower
    args = SimpleNamespace(mm_vision_tower=str(path), mm_vision_select_layer=-2, mm_vision_select_feature="patch")
    return build_vision_tower(args)


@pytest.mark.parametrize("flavour", ["prefixed_full_clip", "flat_vision_only", "pytorch_bin"])
def test_tower_loads_from_hf_directory(tmp_path, flavour):
    from slime_amd import weights as W
    sd = W.make_tower_state_dict(W.TINY, seed=21)                 # 'vision_model.*' (transformers 4.37 names)
    if flavour == "prefixed_full_clip":                           # a full CLIPModel checkpoint: text tower etc. must be ignored
        extra = {"text_model.embeddings.token_embedding.weight": torch.zeros(4, 8), "logit_scale": torch.tensor(1.0),
                 "visual_projection.weight": torch.zeros(8, W.TINY.hidden_size), "text_projection.weight": torch.zeros(8, 8),
                 "vision_model.embeddings.position_ids": torch.arange(577)[None]}
        _write_hf_dir(tmp_path, sd, W.TINY, True, extra)
    elif flavour == "flat_vision_only":                           # transformers 5.x CLIPVisionModel: prefix-free keys
        _write_hf_dir(tmp_path, W.strip_tower_prefix(sd), W.TINY, False)
    else:
        _write_hf_dir(tmp_path, sd, W.TINY, False)
        os.remove(os.path.join(tmp_path, "model.safetensors"))
        torch.save(sd, os.path.join(tmp_path, "pytorch_model.bin"))
    t = _tower_from_dir(tmp_path)
    assert t.is_loaded and t.hidden_size == W.TINY.hidden_size and t.num_patches == 576
    got = t.vision_tower.state_dict()
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    assert t.image_processor.crop_size == {"height": 336, "width": 336} and abs(t.image_processor.image_mean[0] - 0.48145466) < 1e-9
    from slime_amd.model.multimodal_encoder.builder import build_vision_tower
    d = build_vision_tower(SimpleNamespace(mm_vision_tower=str(tmp_path), mm_vision_select_layer=-2), delay_load=True)
    assert not d.is_loaded and d.config.hidden_size == W.TINY.hidden_size      # delay_load reads config.json only


def test_tower_directory_with_wrong_layout_raises(tmp_path):
    from slime_amd import weights as W
    sd = {k.replace("encoder.layers", "transformer.resblocks"): v for k, v in W.make_tower_state_dict(W.TINY, seed=21).items()}
    _write_hf_dir(tmp_path, sd, W.TINY, False)
    with pytest.raises(RuntimeError, match="does not match the CLIP ViT layout"):
        _tower_from_dir(tmp_path)
    with pytest.raises(ValueError, match="Unknown vision tower"):
        _tower_from_dir(os.path.join(tmp_path, "nope"))


def _adapter_files(tmp, asd, how):
    proj = {k: v for k, v in asd.items() if k.startswith("mm_projector.")}
    samp = {k: v for k, v in asd.items() if k.startswith("sampler.")}
    if how == "pretrain":             # mm_projector.bin + sampler.bin, 'model.' prefix, fp16 values (builder.py:106-107)
        torch.save({"model." + k: v.to(torch.float16) if v.dtype == torch.float32 else v for k, v in proj.items()},
                   os.path.join(tmp, "mm_projector.bin"))
        torch.save({"model." + k: v for k, v in samp.items()}, os.path.join(tmp, "sampler.bin"))
    elif how == "lora":               # non_lora_trainables.bin, 'base_model.model.model.' prefix + foreign tensors
        blob = {"base_model.model.model." + k: v for k, v in asd.items()}
        blob["base_model.model.model.embed_tokens.weight"] = torch.zeros(4, 4)
        blob["base_model.model.lm_head.weight"] = torch.zeros(4, 4)
        torch.save(blob, os.path.join(tmp, "non_lora_trainables.bin"))
    else:                             # bare keys in safetensors
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in proj.items()}, os.path.join(tmp, "mm_projector.safetensors"))
        save_file({k: v.contiguous() for k, v in samp.items()}, os.path.join(tmp, "sampler.safetensors"))


@pytest.mark.parametrize("how", ["pretrain", "lora", "bare"])
def test_adapter_checkpoint_files(tmp_path, how):
    from slime_amd import weights as W
    from slime_amd.model.builder import load_adapter_checkpoint, read_adapter_state, canonical_adapter_key
    from slime_amd.model.llava_arch import SlimeVisualEncoder, default_slime_config
    asd = W.make_adapter_state_dict(W.ADAPTER_TINY, seed=12)
    _adapter_files(tmp_path, asd, how)
    state = read_adapter_state(str(tmp_path))
    assert set(state) == set(asd)
    enc = SlimeVisualEncoder(default_slime_config("synthetic:1", hidden_size=256, mm_hidden_size=128))
    load_adapter_checkpoint(enc, str(tmp_path))
    m = enc.get_model()
    got = {"mm_projector." + k: v for k, v in m.mm_projector.state_dict().items()}
    got.update({"sampler." + k: v for k, v in m.sampler.state_dict().items()})
    for k, v in asd.items():
        want = v.to(torch.float16).to(v.dtype) if (how == "pretrain" and k.startswith("mm_projector.") and v.dtype == torch.float32) else v
        assert got[k].dtype == v.dtype and torch.equal(got[k], want), k
    assert m.mm_projector.w_gate.dtype == torch.bfloat16 and m.sampler.post_qformer.pos_embed.dtype == torch.float16
    assert canonical_adapter_key("base_model.model.model.sampler.post_qformer.query") == "sampler.post_qformer.query"
    assert canonical_adapter_key("model.mm_projector.attn.ln_q.weight") == "mm_projector.attn.ln_q.weight"
    assert canonical_adapter_key("model.layers.0.resampler.weight") is None and canonical_adapter_key("lm_head.weight") is None


def test_adapter_checkpoint_errors(tmp_path):
    from slime_amd import weights as W
    from slime_amd.model.builder import load_adapter_checkpoint, read_adapter_state
    from slime_amd.model.llava_arch import SlimeVisualEncoder, default_slime_config
    with pytest.raises(FileNotFoundError):
        read_adapter_state(str(tmp_path))
    asd = W.make_adapter_state_dict(W.ADAPTER_TINY, seed=12)
    part = {"model." + k: v for k, v in asd.items() if k.startswith("mm_projector.") and "attn.ln_q" not in k}
    torch.save(part, os.path.join(tmp_path, "mm_projector.bin"))
    enc = SlimeVisualEncoder(default_slime_config("synthetic:1", hidden_size=256, mm_hidden_size=128))
    with pytest.raises(RuntimeError, match="missing"):
        load_adapter_checkpoint(enc, str(tmp_path))
    load_adapter_checkpoint(enc, str(tmp_path), strict=False)     # the reference's behaviour (strict=False), on request


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_loaded_files_run_on_the_gpu_vs_oracle(tmp_path, dtype):
    """Tower from an HF directory + adapter from non_lora_trainables.bin -> encode_visual vs the oracle on the same tensors."""
    from slime_amd import weights as W
    from slime_amd.model.builder import load_adapter_checkpoint
    from slime_amd.model.llava_arch import SlimeVisualEncoder, default_slime_config
    from oracle import slime_oracle as O
    dev = torch.device("cuda:0")
    tsd = W.make_tower_state_dict(W.TINY, seed=33)
    asd = W.make_adapter_state_dict(W.ADAPTER_TINY, seed=34)
    tdir, adir = tmp_path / "clip", tmp_path / "ckpt"
    tdir.mkdir(); adir.mkdir()
    _write_hf_dir(tdir, tsd, W.TINY, True)
    _adapter_files(adir, asd, "lora")
    enc = SlimeVisualEncoder(default_slime_config(str(tdir), hidden_size=256, mm_hidden_size=128))
    enc.get_vision_tower().load_model()
    load_adapter_checkpoint(enc, str(adir))
    enc.to(dev)
    enc.get_vision_tower().vision_tower.to(dtype)
    px = W.synthetic_pixels(5, seed=35)
    (glob, merged), = enc.encode_visual(px.to(dev), [5], [(672, 672)], merge="spatial")
    ref = O.encode_image(W.strip_tower_prefix(tsd), asd, W.TINY, W.ADAPTER_TINY, px, (672, 672))
    tol = {torch.float16: 6e-3, torch.bfloat16: 3e-2}[dtype]
    assert rel_l2(glob.cpu(), ref["global"]) < tol and rel_l2(merged.cpu(), ref["merged"]) < tol


def _load_verify_tool():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("slime_verify_checkpoint", os.path.join(root, "tools", "verify_checkpoint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_verify_checkpoint_tool_cpu_part(tmp_path, capsys):
    """tools/verify_checkpoint.py on an HF-format directory written from make_tower_state_dict + adapter files
    with the LoRA prefixes: the product loaders, the fp32 oracle, and the per-layer outlier statistic -- a channel planted at x60
    in layer 0's fc2 bias must show up as THE argmax channel of hidden state 1 with max / rms far above the unplanted states."""
    from slime_amd import weights as W
    tsd = W.make_tower_state_dict(W.TINY, seed=33)
    planted = 37
    tsd["vision_model.encoder.layers.0.mlp.fc2.bias"] = tsd["vision_model.encoder.layers.0.mlp.fc2.bias"].clone()
    tsd["vision_model.encoder.layers.0.mlp.fc2.bias"][planted] = 60.0
    asd = W.make_adapter_state_dict(W.ADAPTER_TINY, seed=34)
    tdir, adir = tmp_path / "clip", tmp_path / "ckpt"
    tdir.mkdir(); adir.mkdir()
    _write_hf_dir(tdir, tsd, W.TINY, True)
    _adapter_files(adir, asd, "lora")
    tool = _load_verify_tool()
    out = tmp_path / "report.json"
    rep = tool.main([str(tdir), str(adir), "--cpu-only", "--json", str(out)])
    text = capsys.readouterr().out
    assert rep["crops"] == 5 and rep["hip"] is None and "HIP path: skipped (--cpu-only)" in text
    assert len(rep["outliers"]) == W.TINY.num_hidden_layers                       # hidden_states[0 .. L-1]: what select_layer -2 needs
    s0, s1 = rep["outliers"][0], rep["outliers"][1]
    assert s1["argmax_channel"] == planted and s1["max_abs"] > 50 and s1["max_over_rms"] > 2 * s0["max_over_rms"] and s1["channels_over_20_rms"] <= 2
    assert json.load(open(out))["outliers"] == rep["outliers"]
    assert any("tensors from" in n for n in rep["notes"])
    # without adapter files: the tower is real, the adapter a seeded stand-in, and the report says so
    rep2 = tool.main([str(tdir), "--cpu-only", "--size", "336", "336"])
    assert rep2["crops"] == 3 and any("NO checkpoint given" in n for n in rep2["notes"])


@pytest.mark.gpu
def test_verify_checkpoint_tool_full(tmp_path):
    """The same tool end to end on the GPU box: HIP fp16 / bf16 against the oracle on the loaded files, per stage and per hidden state."""
    from slime_amd import weights as W
    tsd = W.make_tower_state_dict(W.TINY, seed=33)
    asd = W.make_adapter_state_dict(W.ADAPTER_TINY, seed=34)
    tdir, adir = tmp_path / "clip", tmp_path / "ckpt"
    tdir.mkdir(); adir.mkdir()
    _write_hf_dir(tdir, tsd, W.TINY, False)
    _adapter_files(adir, asd, "pretrain")
    rep = _load_verify_tool().main([str(tdir), str(adir)])
    for key, tol in (("fp16", 3e-3), ("bf16", 2e-2)):
        r = rep["hip"][key]
        assert r["tower"] < tol and r["global"] < tol and r["merged_local"] < tol and r["compressed"] < tol, (key, r)
        assert len(r["hidden_states"]) == W.TINY.num_hidden_layers and max(r["hidden_states"]) < tol
