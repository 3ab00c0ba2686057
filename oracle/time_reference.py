#!/usr/bin/env python3
"""Time the REFERENCE's own CPU path next to the oracle (build container only: needs /root/reference).

TEST INFRASTRUCTURE.  Run:  python oracle/time_reference.py [--runs 3]

Times, fp32, default torch threading on all cores of this container:
  reference : llava CLIPVisionTower.forward -> sampler.post_qformer -> mm_projector (+ spatial merge), the modules built by the
              reference's own builders and loaded with the seeded weights (oracle/make_golden.py helpers)
  oracle    : oracle/slime_oracle.py encode path on the same inputs
for BASELINE config 1 (one 336x336 crop, global only) and one config-2 image (1 + 4 crops), >= 1 warm-up, median of N runs.
It also checks that the two agree (<= 2e-5 rel-L2), i.e. that the oracle timed on the GPU box (bench.py cpu_baseline, kind
"port") is the reference's arithmetic at the reference's speed.  Prints one JSON object; DESIGN.md section 5 quotes it.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))


def med_time(fn, runs):
    fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=3)
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    import make_golden as MG
    from slime_amd import weights as W
    import slime_oracle as O
    MG.import_reference()
    from llava import mm_utils as ref_mm
    vl, al = W.CLIP_L_336, W.ADAPTER_8B
    tsd = W.make_tower_state_dict(vl, seed=1234)
    asd = W.make_adapter_state_dict(al, seed=4321)
    tower = MG.build_reference_tower(vl, tsd)
    proj, samp = MG.build_reference_adapter(al, asd)
    flat = W.strip_tower_prefix(tsd)
    pin = "[(336, 672), (672, 336), (672, 672), (1008, 336), (336, 1008)]"
    out = {"nproc": os.cpu_count(), "torch_threads": torch.get_num_threads(), "dtype": "fp32", "runs": args.runs}

    # config 1: one global crop -> tower -> gated projector
    px1 = W.synthetic_pixels(1, seed=7)

    def ref_cfg1():
        return proj(tower(px1)[0])

    def ora_cfg1():
        return O.gated_block_forward(W.sub_state(asd, "mm_projector."), O.tower_forward(flat, vl, px1)[0], al.num_heads)

    # one config-2 image: 1 + 4 crops -> tower -> gated global + post_qformer / MLP / spatial merge on the 4 local crops
    px5 = W.synthetic_pixels(5, seed=7)

    def ref_cfg2():
        f = tower(px5)
        g = proj(f[0])
        loc = proj(samp.post_qformer(f[1:]))
        nw, nh = ref_mm.get_anyres_image_grid_shape((672, 672), pin, 336)
        m = loc.view(nh, nw, samp.grid_size, samp.grid_size, -1).permute(0, 2, 1, 3, 4).contiguous().flatten(0, 3)
        return g, m

    def ora_cfg2():
        r = O.encode_image(flat, asd, vl, al, px5, (672, 672))
        return r["global"], r["merged"]

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())

    out["agreement_rel_l2"] = {"cfg1_global": rel(ora_cfg1(), ref_cfg1()), "cfg2_global": rel(ora_cfg2()[0], ref_cfg2()[0]),
                               "cfg2_merged_local": rel(ora_cfg2()[1], ref_cfg2()[1])}
    for name, crops, rf, of in (("cfg1_1_crop", 1, ref_cfg1, ora_cfg1), ("cfg2_one_image_1+4_crops", 5, ref_cfg2, ora_cfg2)):
        rm, rmin = med_time(rf, args.runs)
        om, omin = med_time(of, args.runs)
        out[name] = {"reference_s_median": round(rm, 3), "reference_crops_per_s": round(crops / rm, 3),
                     "oracle_s_median": round(om, 3), "oracle_crops_per_s": round(crops / om, 3),
                     "oracle_over_reference_time": round(om / rm, 3)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
