#!/usr/bin/env python3
"""A/B of product-library variants (tools/build_variants.sh) on the 40-crop tower: each variant runs in its own process (one library
per process), rounds interleaved; prints ms per 40 crops (two streams x 20 | one stream x 40) and a checksum of the features so that
bit-equality across variants is visible.  usage: lib_variant_ab.py [--rounds R] name[@KEY=VAL[,KEY=VAL]] ...   (name 'product' =
slime_amd/libslime_hip.so; the optional @ part sets environment variables for that variant's process, e.g. product@SLIME_KEEP_ROW_MAJOR=1)"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import time, hashlib, torch
    sys.path.insert(0, ROOT)
    from slime_amd import _lib
    label = sys.argv[2]
    name = label.split("@")[0]
    if name != "product":
        _lib.LIB_PATH = os.path.join(ROOT, "slime_amd", "variants", f"libslime_hip_{name}.so")
    from slime_amd import ops, weights as W
    dev = torch.device("cuda:0"); dt = torch.bfloat16
    tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
    px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
    pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
    side = torch.cuda.Stream(); parts = list(px.chunk(2))

    def run2():
        cur = torch.cuda.current_stream(); side.wait_stream(cur)
        with torch.cuda.stream(side): b = ops.tower_forward(pts[1], parts[1])
        a = ops.tower_forward(pts[0], parts[0]); cur.wait_stream(side)
        return a, b

    def run1():
        return ops.tower_forward(pts[0], px)
    a, b = run2(); torch.cuda.synchronize()
    h = hashlib.sha1(torch.cat([a, b]).float().cpu().numpy().tobytes()).hexdigest()[:12]
    ts = []
    for fn in (run2, run1):
        best = []
        for rep in range(3):
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(12): fn()
            torch.cuda.synchronize(); best.append((time.perf_counter() - t0) / 12 * 1e3)
        ts.append(sorted(best)[1])
    rec = {"name": label, "two": ts[0], "one": ts[1], "sha": h, "weight_MB": round(ops.packed_weight_bytes(pts[0]) / 1e6, 1)}
    if os.environ.get("AB_ADAPTER") == "1":
        # the bench step's second half: fused adapter over the 40 crops' features (8 images x (1+4)), alone and behind the two-stream tower
        asd = W.make_adapter_state_dict(W.ADAPTER_8B, seed=4321)
        pg = ops.pack_gated(W.sub_state(asd, "mm_projector."), W.ADAPTER_8B, dt, dev)
        post = ops.pack_resampler(W.sub_state(asd, "sampler.post_qformer."), 1024, 8, 576, dt, dev, W.ADAPTER_8B.ln_eps)
        feats = torch.cat([a, b])
        ad = lambda: ops.adapter_forward(pg, post, feats, 8, 4, 2, 2, True, -1, dt)
        out = ad(); torch.cuda.synchronize()
        rec["adapter_sha"] = hashlib.sha1(out.float().cpu().numpy().tobytes()).hexdigest()[:12]
        torch.save(out.float().cpu(), f"/tmp/ab_adapter_{label}.pt")

        def step():
            x, y = run2(); return ops.adapter_forward(pg, post, torch.cat([x, y]), 8, 4, 2, 2, True, -1, dt)
        for key, fn in (("adapter", ad), ("step", step)):
            best = []
            for rep in range(3):
                for _ in range(3): fn()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(12): fn()
                torch.cuda.synchronize(); best.append((time.perf_counter() - t0) / 12 * 1e3)
            rec[key] = sorted(best)[1]
    print(json.dumps(rec))
    sys.exit(0)

args = sys.argv[1:]; rounds = 3
if args and args[0] == "--rounds": rounds = int(args[1]); args = args[2:]
print("variant: two streams x 20 crops ms | one stream x 40 crops ms | sha1 of the 40-crop features (median of 3 x 12 steps per round)")
def _rel():
    import torch
    fs = [f"/tmp/ab_adapter_{n}.pt" for n in args]
    if os.environ.get("AB_ADAPTER") == "1" and all(os.path.exists(f) for f in fs) and len(fs) > 1:
        ref = torch.load(fs[0])
        for n, f in zip(args[1:], fs[1:]):
            t = torch.load(f); print(f"adapter output of {n} vs {args[0]}: rel-L2 {float((t - ref).norm() / ref.norm()):.3e}", flush=True)


for r in range(rounds):
    for n in args:
        env = dict(os.environ)
        if "@" in n:
            env.update(kv.split("=", 1) for kv in n.split("@", 1)[1].split(","))
        o = subprocess.run([sys.executable, __file__, "--child", n], capture_output=True, text=True, timeout=600, env=env)
        try:
            d = json.loads(o.stdout.strip().splitlines()[-1])
            extra = f" | adapter {d['adapter']:.3f} ms, tower + adapter {d['step']:.2f} ms, adapter sha {d['adapter_sha']}" if "adapter" in d else ""
            print(f"{d['name']:36s}: {d['two']:6.2f} | {d['one']:6.2f} | {d['sha']} | weights {d.get('weight_MB')} MB{extra}", flush=True)
        except Exception:
            print(n, "FAILED", o.stderr[-600:], flush=True)
_rel()
