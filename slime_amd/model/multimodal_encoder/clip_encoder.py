"""``CLIPVisionTower`` backed by the HIP tower: the plugin contract of
llava/model/multimodal_encoder/clip_encoder.py:8-89 (constructor arguments, attributes, ``load_model``,
``feature_select``, ``forward`` on a tensor or a list, the dtype/device/config/hidden_size/num_patches
properties) with the arithmetic of HF ``CLIPVisionModel`` replaced by ``slime_vit_forward``.

Differences that are deliberate and documented in DESIGN.md:
  * only the layers that feed ``hidden_states[select_layer]`` run (the reference executes the dead
    last layer and ``post_layernorm`` and materialises all 25 hidden states);
  * MFMA operands are bf16/fp16 but the residual stream, LayerNorm and softmax statistics are fp32;
  * all crops of a call go through the tower as ONE batch, split over two HIP streams so one
    kernel's partial last round of workgroups is filled by the other half's kernels.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn

from ... import ops
from ...image_processor import ClipImageProcessor
from ...weights import VisionConfig, make_tower_state_dict, canonical_tower_key

SYNTHETIC_PREFIX = "synthetic:"


class _Embeddings(nn.Module):
    def __init__(self, c: VisionConfig):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.empty(c.hidden_size))
        self.patch_embedding = nn.Conv2d(3, c.hidden_size, c.patch_size, c.patch_size, bias=False)
        self.position_embedding = nn.Embedding(c.seq_len, c.hidden_size)


class _Attention(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj, self.v_proj = nn.Linear(d, d), nn.Linear(d, d)
        self.q_proj, self.out_proj = nn.Linear(d, d), nn.Linear(d, d)


class _Mlp(nn.Module):
    def __init__(self, d, f):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(d, f), nn.Linear(f, d)


class _Layer(nn.Module):
    def __init__(self, c: VisionConfig):
        super().__init__()
        self.self_attn = _Attention(c.hidden_size)
        self.layer_norm1 = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.mlp = _Mlp(c.hidden_size, c.intermediate_size)
        self.layer_norm2 = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class _Encoder(nn.Module):
    def __init__(self, c: VisionConfig):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(c) for _ in range(c.num_hidden_layers)])


class _VisionTransformer(nn.Module):
    def __init__(self, c: VisionConfig):
        super().__init__()
        self.embeddings = _Embeddings(c)
        self.pre_layrnorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)   # sic (HF key)
        self.encoder = _Encoder(c)
        self.post_layernorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class HipCLIPVisionModel(nn.Module):
    """Parameter container with HF ``CLIPVisionModel`` state-dict keys (transformers-4.37 names,
    ``vision_model.`` prefix; prefix-free 5.x keys are accepted on load) whose forward is the HIP
    tower.  The torch sub-modules are never called: they only give the parameters their names."""

    def __init__(self, config: VisionConfig, compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        self.config = config
        self.vision_model = _VisionTransformer(config)
        self.compute_dtype = compute_dtype     # MFMA operand type used when the parameters are fp32
        self.two_streams = True
        # crop count from which encode() splits the batch over two streams: 8 is the measured break-even of ViT-L/14-336 on MI355X
        # (profiles/r03_stream_split_sweep.txt) -- another tower or device may want another value (SLIME_TOWER_SPLIT_MIN)
        self.split_min_crops = int(os.environ.get("SLIME_TOWER_SPLIT_MIN", "8"))
        self.force_streams = 0                # tools/stream_split_sweep.py: 1 / 2 = override the split policy below
        self._packed: Dict = {}
        self._streams: Optional[List[torch.cuda.Stream]] = None
        self.requires_grad_(False)

    # -- state dict: accept both HF key generations -------------------------------------------
    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for k in [k for k in state_dict if k.startswith(prefix)]:
            tail = k[len(prefix):]
            canon = "vision_model." + canonical_tower_key(tail)
            if tail != canon and canon.split(".")[1] in ("embeddings", "pre_layrnorm", "encoder", "post_layernorm"):
                state_dict[prefix + canon] = state_dict.pop(k)
        for k in [k for k in state_dict if k.endswith("embeddings.position_ids")]:
            state_dict.pop(k)                      # HF buffer, pure arange
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._packed.clear()

    def _apply(self, fn, *a, **kw):
        self._packed.clear()                       # .to(dtype/device) invalidates the packed weights
        return super()._apply(fn, *a, **kw)

    @property
    def dtype(self) -> torch.dtype:
        return self.vision_model.embeddings.class_embedding.dtype

    @property
    def device(self) -> torch.device:
        return self.vision_model.embeddings.class_embedding.device

    def operand_dtype(self) -> torch.dtype:
        return self.dtype if self.dtype in (torch.bfloat16, torch.float16) else self.compute_dtype

    def packed(self, select_layer: int, slot: int = 0) -> ops.PackedTower:
        key = (select_layer, self.operand_dtype(), str(self.device), slot)
        if key not in self._packed:
            base = self._packed.get((select_layer, self.operand_dtype(), str(self.device), 0))
            if base is not None:                   # extra slots share the weights, own a workspace
                self._packed[key] = ops.PackedTower(base.cfg, base.dtype, base.layers_run, base.tensors, base.desc)
            else:
                self._packed[key] = ops.pack_tower(self.state_dict(), self.config, self.operand_dtype(), self.device,
                                                   select_layer)
        return self._packed[key]

    @torch.no_grad()
    def encode(self, pixel_values: torch.Tensor, select_layer: int = -2, keep_cls: bool = False,
               out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """[N,3,S,S] -> hidden_states[select_layer] as [N, P(+1), D] (one batched launch sequence)."""
        out_dtype = out_dtype or pixel_values.dtype
        n = pixel_values.shape[0]
        # two half batches on two streams from 8 crops on: round 3's sweep (tools/stream_split_sweep.py, profiles/r03_stream_split_sweep.txt)
        # has the split ahead or equal at every count >= 8 (9 crops 4.79 -> 4.43 ms, 14: 6.65 -> 5.91, 40: 16.5 -> 15.1) and behind
        # below (7 crops 3.49 vs 4.01: two passes of the 2.4 ms floor); rounds 1-2 measured 14-16 as the break-even with the old
        # small-grid kernels
        split = self.two_streams and n >= self.split_min_crops
        if self.force_streams:
            split = self.force_streams == 2 and n >= 2
        if not split:
            return ops.tower_forward(self.packed(select_layer), pixel_values, out_dtype, keep_cls)
        # two independent half batches on two streams: fills each kernel's last partial round.  The first half stays on the CALLER's
        # stream, only the second gets a stream of its own: ROCm maps HIP streams onto 4 hardware queues by default, and with two
        # private tower streams + the caller's + a tail stream + RCCL's, the two halves landed on ONE queue and serialised (round 4:
        # the forced-collective bench line lost 15 % that way, profiles/r04_collective_hw_queues.txt).
        if self._streams is None:
            self._streams = [torch.cuda.Stream(device=self.device)]
        cur = torch.cuda.current_stream()
        side = self._streams[0]
        half = (n + 1) // 2
        rows = self.config.num_patches + (1 if keep_cls else 0)
        out = torch.empty((n, rows, self.config.hidden_size), dtype=out_dtype, device=pixel_values.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            ops.tower_forward(self.packed(select_layer, 1), pixel_values[half:], out_dtype, keep_cls, out=out[half:])
        ops.tower_forward(self.packed(select_layer, 0), pixel_values[:half], out_dtype, keep_cls, out=out[:half])   # both halves land in one tensor
        pixel_values.record_stream(side)
        out.record_stream(side)
        cur.wait_stream(side)
        return out

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor, output_hidden_states: bool = False, **_):
        """HF-like call: returns an object with ``hidden_states`` (L+1 entries: embeddings after pre_layrnorm, then the output
        of every layer -- ONE tower pass over all L layers that snapshots the fp32 residual stream after each,
        ``slime_vit_forward_states``; use :meth:`encode` on the hot path, which runs only the layers its state needs) and
        ``last_hidden_state`` (HF: hidden_states[-1], without post_layernorm)."""
        L = self.config.num_hidden_layers
        if output_hidden_states:
            st = ops.tower_hidden_states(self.packed(L), pixel_values).to(pixel_values.dtype)
            hs = tuple(st[i] for i in range(L + 1))
            return SimpleNamespace(hidden_states=hs, last_hidden_state=hs[-1])
        return SimpleNamespace(hidden_states=None, last_hidden_state=self.encode(pixel_values, L, keep_cls=True))


def _read_hf_vision_config(path: str) -> VisionConfig:
    cfg = json.load(open(os.path.join(path, "config.json")))
    cfg = cfg.get("vision_config", cfg)
    return VisionConfig(hidden_size=cfg.get("hidden_size", 1024), intermediate_size=cfg.get("intermediate_size", 4096),
                        num_hidden_layers=cfg.get("num_hidden_layers", 24),
                        num_attention_heads=cfg.get("num_attention_heads", 16), image_size=cfg.get("image_size", 336),
                        patch_size=cfg.get("patch_size", 14), layer_norm_eps=cfg.get("layer_norm_eps", 1e-5),
                        hidden_act=cfg.get("hidden_act", "quick_gelu"))


def _read_hf_weights(path: str) -> Dict[str, torch.Tensor]:
    st = os.path.join(path, "model.safetensors")
    if os.path.isfile(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
    return {k: v for k, v in sd.items() if "text_model" not in k and "projection" not in k and "logit_scale" not in k}


def _resolve_local(name: str) -> str:
    if os.path.isdir(name):
        return name
    try:
        from huggingface_hub import snapshot_download
        return snapshot_download(name, local_files_only=True)
    except Exception as e:   # no network in this environment: say so plainly
        raise FileNotFoundError(
            f"vision tower '{name}' is neither a local directory nor in the local HF cache (offline). "
            f"Pass a directory with config.json + model.safetensors, or '{SYNTHETIC_PREFIX}<seed>'.") from e


def check_tower_keys(result, source: str) -> None:
    """The parameter container starts uninitialised, so a checkpoint with another naming scheme (open_clip, a partial
    file) must not load silently: only the dead ``post_layernorm`` (never run: hidden_states[-2]) and HF's ``position_ids``
    buffer may be absent, and nothing may be left over."""
    missing = [k for k in result.missing_keys if "post_layernorm" not in k and "position_ids" not in k]
    unexpected = [k for k in result.unexpected_keys if "position_ids" not in k]
    if missing or unexpected:
        raise RuntimeError(f"vision tower checkpoint {source!r} does not match the CLIP ViT layout: "
                           f"{len(missing)} missing (e.g. {missing[:3]}), {len(unexpected)} unexpected (e.g. {unexpected[:3]})")


class CLIPVisionTower(nn.Module):
    def __init__(self, vision_tower, args, delay_load=False):
        super().__init__()
        self.is_loaded = False
        self.vision_tower_name = vision_tower
        self.select_layer = args.mm_vision_select_layer
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch")
        self._compute_dtype = getattr(args, "mm_vision_compute_dtype", torch.bfloat16)
        if not delay_load:
            self.load_model()
        elif getattr(args, "unfreeze_mm_vision_tower", False):
            self.load_model()
        else:
            self.cfg_only = self._config_only()

    def _config_only(self) -> VisionConfig:
        if self.vision_tower_name.startswith(SYNTHETIC_PREFIX):
            return VisionConfig()
        return _read_hf_vision_config(_resolve_local(self.vision_tower_name))

    def load_model(self, device_map=None):
        if self.is_loaded:
            print("{} is already loaded, `load_model` called again, skipping.".format(self.vision_tower_name))
            return
        name = self.vision_tower_name
        if name.startswith(SYNTHETIC_PREFIX):       # offline stand-in for the hub checkpoint
            cfg = VisionConfig()
            sd = make_tower_state_dict(cfg, seed=int(name[len(SYNTHETIC_PREFIX):] or 1234))
            self.image_processor = ClipImageProcessor()
        else:
            path = _resolve_local(name)
            cfg, sd = _read_hf_vision_config(path), _read_hf_weights(path)
            self.image_processor = ClipImageProcessor.from_pretrained(path)
        if cfg.hidden_act != "quick_gelu":
            raise ValueError(f"unsupported CLIP activation {cfg.hidden_act!r}: the HIP tower implements quick_gelu")
        self.vision_tower = HipCLIPVisionModel(cfg, self._compute_dtype)
        check_tower_keys(self.vision_tower.load_state_dict(sd, strict=False), name)
        if device_map not in (None, "auto") and not isinstance(device_map, dict):
            self.vision_tower.to(device_map)
        self.vision_tower.requires_grad_(False)
        self.is_loaded = True

    def feature_select(self, image_forward_outs):
        image_features = image_forward_outs.hidden_states[self.select_layer]
        return self._select(image_features)

    def _select(self, image_features):
        if self.select_feature == "patch":
            return image_features[:, 1:]
        if self.select_feature == "cls_patch":
            return image_features
        raise ValueError(f"Unexpected select feature: {self.select_feature}")

    def _keep_cls(self) -> bool:
        if self.select_feature == "patch":
            return False
        if self.select_feature == "cls_patch":
            return True
        raise ValueError(f"Unexpected select feature: {self.select_feature}")

    @torch.no_grad()
    def forward(self, images: Union[torch.Tensor, List[torch.Tensor]], out_dtype: Optional[torch.dtype] = None):
        """Tensor [N,3,336,336] -> [N,576,D]; list of [3,336,336] -> list of [1,576,D]; results in the
        input's dtype (clip_encoder.py:46-58).  ``out_dtype`` (extension) lets the fused adapter path ask
        for the fp32 residual stream instead."""
        keep = self._keep_cls()
        if type(images) is list:
            if len(images) == 0:
                return []
            batch = torch.stack([im.to(device=self.device) for im in images], dim=0)   # one batched tower run
            feats = self.vision_tower.encode(self._cast(batch), self.select_layer, keep, out_dtype or images[0].dtype)
            return [feats[i:i + 1].to(images[i].dtype if out_dtype is None else out_dtype) for i in range(len(images))]
        images = images.to(device=self.device)
        return self.vision_tower.encode(self._cast(images), self.select_layer, keep, out_dtype or images.dtype)

    def _cast(self, x: torch.Tensor) -> torch.Tensor:
        # `images.to(dtype=self.dtype)` in the reference: 16-bit towers round the pixels first
        dt = self.vision_tower.operand_dtype()
        return x if x.dtype in (torch.float32, dt) else x.to(dt)

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return self.vision_tower.dtype

    @property
    def device(self):
        return self.vision_tower.device

    @property
    def config(self):
        return self.vision_tower.config if self.is_loaded else self.cfg_only

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches_per_side(self):
        return self.config.image_size // self.config.patch_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2
