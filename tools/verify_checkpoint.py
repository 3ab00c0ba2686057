#!/usr/bin/env python3
"""Real-checkpoint verifier: one command for a user who HAS the files.

    python tools/verify_checkpoint.py <hf_clip_dir> [<slime_ckpt_dir>] [--image PATH | --size W H] [--cpu-only] [--json OUT]

  <hf_clip_dir>     a CLIP directory as the reference's CLIPVisionTower.load_model reads it (config.json, model.safetensors or
                    pytorch_model.bin, preprocessor_config.json; llava/model/multimodal_encoder/clip_encoder.py:25-34)
  <slime_ckpt_dir>  optional: a directory (or file) with mm_projector.bin / sampler.bin / non_lora_trainables.bin (or .safetensors),
                    the adapter files llava/model/builder.py:93-108,161-166 loads; without it a seeded synthetic adapter of the
                    checkpoint's width stands in and the report says so

What it does -- every weight goes through the PRODUCT loaders (build_vision_tower, load_adapter_checkpoint), nothing else:
  1. one image (file, or seeded synthetic uint8 noise; default 672 x 672 -> 1 global + 4 local crops) through process_images('anyres');
  2. the fp32 CPU oracle (oracle/slime_oracle.py: the pinned restatement of the reference) on those crops: every stage of
     encode_images, plus max |activation| of the residual stream after every layer -- the OUTLIER statistic of the real weights
     that the synthetic stress test (tests/test_gpu_path.py::test_outlier_channel_stress) can only guess at;
  3. on a GPU box: the HIP path in fp16 (the reference's inference dtype) and bf16 on the same crops; rel-L2 against the oracle per
     stage (tower, every hidden state, global, compressed, merged local) and whether north_star's 1e-3 holds for fp16.
`--cpu-only` (or no GPU) stops after 2: loaders + oracle + outlier statistics, the part the CPU suite runs (tests/test_checkpoints.py).
The oracle is the CHECKER here, as in tests/: this tool is not on any product path.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def load_models(clip_dir: str, ckpt: str | None, seed: int):
    """(encoder, fp32 tower state dict, fp32-or-stored adapter state dict, VisionConfig, AdapterConfig, notes)."""
    from slime_amd import weights as W
    from slime_amd.model.llava_arch import SlimeVisualEncoder, default_slime_config
    from slime_amd.model.builder import load_adapter_checkpoint, read_adapter_state
    notes = []
    clip_dir = os.path.abspath(clip_dir)
    if ckpt is not None:
        state = read_adapter_state(ckpt)
        H, D = state["mm_projector.projection.0.weight"].shape
    else:
        state, D, H = None, None, None
    # the tower first (its width decides the adapter geometry when no adapter files are given)
    probe = SimpleNamespace(mm_vision_tower=clip_dir, mm_vision_select_layer=-2, mm_vision_select_feature="patch")
    from slime_amd.model.multimodal_encoder.builder import build_vision_tower
    vcfg = build_vision_tower(probe, delay_load=True).config
    if D is None:
        D = vcfg.hidden_size
        H = 4096 if D == 1024 else 2 * D
    if D != vcfg.hidden_size:
        raise SystemExit(f"adapter expects tower width {D}, the CLIP directory has {vcfg.hidden_size}")
    acfg = W.AdapterConfig(mm_hidden_size=D, hidden_size=H)
    enc = SlimeVisualEncoder(default_slime_config(clip_dir, hidden_size=H, mm_hidden_size=D))
    enc.get_vision_tower().load_model()
    if ckpt is not None:
        load_adapter_checkpoint(enc, ckpt)
        notes.append(f"adapter: {len(state)} tensors from {ckpt}")
    else:
        asd = W.make_adapter_state_dict(acfg, seed=seed)
        enc.load_visual_state(None, asd)
        notes.append(f"adapter: NO checkpoint given -- seeded synthetic adapter (seed {seed}, {D} -> {H}); only the tower is real")
    m = enc.get_model()
    asd = {"mm_projector." + k: v.detach().clone() for k, v in m.mm_projector.state_dict().items()}
    asd.update({"sampler." + k: v.detach().clone() for k, v in m.sampler.state_dict().items()})
    tsd = {k: v.detach().float().clone() for k, v in W.strip_tower_prefix(enc.get_vision_tower().vision_tower.state_dict()).items()}
    return enc, tsd, asd, vcfg, acfg, notes


def make_crops(enc, image_path, size, seed):
    from PIL import Image
    from slime_amd import mm_utils as M
    if image_path:
        img = Image.open(image_path).convert("RGB")
        src = image_path
    else:
        w, h = size
        img = Image.fromarray(np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB")
        src = f"synthetic uint8 noise {w}x{h}, seed {seed}"
    px = M.process_images([img], enc.get_vision_tower().image_processor, enc.config)
    px = px[0] if isinstance(px, list) else px[0]
    return px, img.size, src


def oracle_report(tsd, asd, vcfg, acfg, px, image_size, select_layer=-2):
    from oracle import slime_oracle as O
    t0 = time.perf_counter()
    L = vcfg.num_hidden_layers
    idx = select_layer if select_layer >= 0 else L + 1 + select_layer
    hs = O.clip_hidden_states(tsd, vcfg, px, n_layers=idx)
    # the outlier statistic: per hidden state, the largest |activation|, the RMS, and how many channels exceed 20 x RMS
    layers = []
    for i, h in enumerate(hs):
        rms = float(h.pow(2).mean().sqrt())
        amax = h.abs().amax(dim=(0, 1))
        layers.append({"state": i, "max_abs": round(float(amax.max()), 3), "rms": round(rms, 4), "max_over_rms": round(float(amax.max()) / max(rms, 1e-30), 1),
                       "channels_over_20_rms": int((amax > 20 * rms).sum()), "argmax_channel": int(amax.argmax())})
    ref = O.encode_image(tsd, asd, vcfg, acfg, px, image_size, select_layer=select_layer)
    return hs, ref, layers, time.perf_counter() - t0


def hip_report(enc, px, image_size, hs_ref, ref, dtype, dev, acfg):
    from slime_amd import ops, mm_utils as M
    enc.to(dev)
    tower = enc.get_vision_tower()
    tower.vision_tower.to(dtype)
    model = enc.get_model()
    x = px.to(dev)
    out = {}
    feats = tower(x, out_dtype=torch.float32)
    out["tower"] = rel_l2(feats, ref["tower"])
    st = tower.vision_tower(x, output_hidden_states=True).hidden_states           # all L+1 states of ONE pass, fp32
    out["hidden_states"] = [round(rel_l2(st[i], hs_ref[i]), 6) for i in range(len(hs_ref))]
    n_local = px.shape[0] - 1
    nw, nh = M.get_anyres_image_grid_shape(image_size, enc.config.image_grid_pinpoints, tower.config.image_size)
    feats_t = tower(x.to(dtype), out_dtype=dtype)
    tok = ops.adapter_forward(model.mm_projector.packed(dtype), model.sampler.post_qformer.packed(feats_t.shape[1], dtype) if n_local else None,
                              feats_t, 1, n_local, nw, nh, True, int(model.mm_projector.learnable_gated), torch.float32)[0]
    P = tower.num_patches
    out["global"] = rel_l2(tok[:P], ref["global"])
    if n_local:
        out["merged_local"] = rel_l2(tok[P:], ref["merged"])
        comp = model.sampler.post_qformer(feats[1:], out_dtype=torch.float32, operand_dtype=dtype)
        out["compressed"] = rel_l2(comp, ref["compressed"])
    torch.cuda.synchronize()
    return {k: (round(v, 6) if isinstance(v, float) else v) for k, v in out.items()}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("clip_dir")
    ap.add_argument("ckpt", nargs="?", default=None)
    ap.add_argument("--image", default=None, help="an image file; default: seeded synthetic noise")
    ap.add_argument("--size", type=int, nargs=2, default=(672, 672), metavar=("W", "H"))
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--cpu-only", action="store_true", help="loaders + oracle + outlier statistics only")
    ap.add_argument("--json", default=None, help="also write the report to this file")
    args = ap.parse_args(argv)

    enc, tsd, asd, vcfg, acfg, notes = load_models(args.clip_dir, args.ckpt, args.seed)
    px, image_size, src = make_crops(enc, args.image, tuple(args.size), args.seed)
    print(f"tower : {args.clip_dir}: ViT {vcfg.hidden_size} x {vcfg.num_hidden_layers} layers, {vcfg.image_size}/{vcfg.patch_size}, {len(tsd)} tensors")
    for n in notes:
        print(n)
    print(f"image : {src} -> {px.shape[0]} crops (1 global + {px.shape[0] - 1} local)")
    hs, ref, layers, sec = oracle_report(tsd, asd, vcfg, acfg, px, image_size)
    print(f"oracle: fp32 CPU, {sec:.1f} s, {torch.get_num_threads()} threads")
    print("residual stream per hidden state (fp32 oracle): state  max|x|   rms   max/rms  channels > 20 rms  argmax channel")
    for r in layers:
        print(f"    {r['state']:3d}  {r['max_abs']:9.3f}  {r['rms']:7.4f}  {r['max_over_rms']:7.1f}  {r['channels_over_20_rms']:5d}  {r['argmax_channel']:5d}")
    report = {"clip_dir": args.clip_dir, "ckpt": args.ckpt, "image": src, "crops": int(px.shape[0]), "notes": notes,
              "oracle_seconds": round(sec, 2), "outliers": layers, "hip": None}
    worst = max(r["max_over_rms"] for r in layers)
    print(f"largest max/rms over the {len(layers)} states: {worst:.1f}  (seeded random weights: ~5; released CLIP-L checkpoints carry 'massive activation' channels far above that)")
    gpu = torch.cuda.is_available() and not args.cpu_only
    if not gpu:
        print("HIP path: skipped (" + ("--cpu-only" if args.cpu_only else "no GPU: slime_amd has no CPU path") + ")")
    else:
        dev = torch.device("cuda:0")
        report["hip"] = {}
        for dtype, key, bound in ((torch.float16, "fp16", 1e-3), (torch.bfloat16, "bf16", 1.2e-2)):
            r = hip_report(enc, px, image_size, hs, ref, dtype, dev, acfg)
            report["hip"][key] = r
            ok = max(r["global"], r.get("merged_local", 0.0)) <= bound
            print(f"HIP {key}: rel-L2 vs oracle  tower {r['tower']:.3e}  global {r['global']:.3e}"
                  + (f"  compressed {r['compressed']:.3e}  merged_local {r['merged_local']:.3e}" if "merged_local" in r else "")
                  + f"   projector outputs <= {bound:g}: {'yes' if ok else 'NO'}")
            print(f"         per hidden state: {' '.join(f'{e:.1e}' for e in r['hidden_states'])}")
            r["projector_within_bound"], r["bound"] = bool(ok), bound
    if args.json:
        with open(args.json, "w") as f:
            json.dump(report, f, indent=1)
    return report


if __name__ == "__main__":
    main()
