"""north_star's accuracy target, measured: rel-L2 of the projector outputs (global gated tokens, merged local tokens) against
the fp32 CPU oracle at ViT-L/14-336 + SliME-8B adapter dims, for fp16 and bf16 operands, one BASELINE-config-2 image
(1 global + 4 local crops, 2 x 2 grid).  Per-stage figures so a miss can be located.  (Checker only: imports oracle/.)
    python tools/north_star_parity.py [n_images]
"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from slime_amd import ops, weights as W  # noqa: E402
from oracle import slime_oracle as O  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda:0")
    A = W.ADAPTER_8B
    tsd = W.strip_tower_prefix(W.make_tower_state_dict(W.CLIP_L_336, seed=1234))
    asd = W.make_adapter_state_dict(A, seed=4321)
    proj_sd, post_sd = W.sub_state(asd, "mm_projector."), W.sub_state(asd, "sampler.post_qformer.")
    px = W.synthetic_pixels(5 * n_img, seed=77)
    t0 = time.time()
    ref_feats = O.tower_forward(tsd, W.CLIP_L_336, px)
    refs = []
    for i in range(n_img):
        f = ref_feats[5 * i:5 * i + 5]
        g_ref = O.gated_block_forward(proj_sd, f[0], A.num_heads)
        comp = O.resampler_forward(post_sd, f[1:], A.num_heads, A.ln_eps)
        refs.append((g_ref, comp, O.spatial_merge(O.mlp_projector(proj_sd, comp), 2, 2, 12)))
    print(f"oracle: {time.time() - t0:.1f} s for {5 * n_img} crops", flush=True)
    out = {}
    for dt in (torch.float16, torch.bfloat16):
        pt = ops.pack_tower(tsd, W.CLIP_L_336, dt, dev)
        pg = ops.pack_gated(proj_sd, A, dt, dev)
        post = ops.pack_resampler(post_sd, 1024, 8, 576, dt, dev, A.ln_eps)
        feats32 = ops.tower_forward(pt, px.to(dev), out_dtype=torch.float32)
        feats = ops.tower_forward(pt, px.to(dev), out_dtype=dt)
        tokens = ops.adapter_forward(pg, post, feats, n_img, 4, 2, 2, True, -1, torch.float32)
        # adapter alone on the ORACLE's tower output (isolates the adapter's own error)
        tokens_a = ops.adapter_forward(pg, post, ref_feats.to(dev).to(dt), n_img, 4, 2, 2, True, -1, torch.float32)
        comp = ops.resampler_forward(post, torch.cat([ref_feats[5 * i + 1:5 * i + 5] for i in range(n_img)]).to(dev))
        torch.cuda.synchronize()
        r = {"tower_fp32_out": rel(feats32, ref_feats), "tower_T_out": rel(feats.float(), ref_feats),
             "global": max(rel(tokens[i][:576], refs[i][0]) for i in range(n_img)),
             "merged_local": max(rel(tokens[i][576:], refs[i][2]) for i in range(n_img)),
             "adapter_only_global": max(rel(tokens_a[i][:576], refs[i][0]) for i in range(n_img)),
             "adapter_only_merged_local": max(rel(tokens_a[i][576:], refs[i][2]) for i in range(n_img)),
             "post_qformer_only": rel(comp.float(), torch.cat([r_[1] for r_ in refs]))}
        out[str(dt).replace("torch.", "")] = r
        print(str(dt), {k: f"{v:.2e}" for k, v in r.items()}, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
