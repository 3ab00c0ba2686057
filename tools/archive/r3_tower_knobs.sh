#!/bin/bash
# Round 3: does the direct-B geometry (two GEMM workgroups per CU) change the tower-level verdicts of round 2?
# attn32 vs attn64r, stream count, fc2 / out_proj tile choice -- all inside the 40-crop tower, interleaved rounds.
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== attention variants =="; timeout 200 python tools/tower_ab.py attn32 2>&1 | grep -v amdgpu.ids
echo "== stream count =="; timeout 200 python tools/tower_streams.py 2>&1 | grep -v amdgpu.ids
echo "== fc2 / out_proj tiles (auto = fc2 ping-pong 256, out_proj direct-B) =="
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time, torch
sys.path.insert(0, ".")
from slime_amd import ops, _lib, weights as W
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
tsd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)
pts = [ops.pack_tower(tsd, W.CLIP_L_336, dt, dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
parts = list(px.chunk(2))
def run():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    for pt, s, p in zip(pts, streams, parts):
        with torch.cuda.stream(s): ops.tower_forward(pt, p)
    for s in streams: cur.wait_stream(s)
CFG = [("auto", {}), ("fc2 pp192", {(1024, 4096): 9}), ("fc2 direct-B", {(1024, 4096): 12}), ("fc2 stream192", {(1024, 4096): 10}),
       ("out pp256", {(1024, 1024): 4}), ("out pp192", {(1024, 1024): 9}), ("qkv stream192 (r2)", {(3072, 1024): 10}), ("fc1 stream256 (r2)", {(4096, 1024): 11})]
for _ in range(3): run()
for rep in range(3):
    for name, rules in CFG:
        lib.slime_gemm_set_shape_tile(0, 0, 0)
        for (N, K), t in rules.items(): lib.slime_gemm_set_shape_tile(N, K, t)
        for _ in range(2): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): run()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 8
        print(f"round {rep} {name:20s}: {t*1e3:6.2f} ms", flush=True)
lib.slime_gemm_set_shape_tile(0, 0, 0)
PY
