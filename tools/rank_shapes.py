#!/usr/bin/env python3
"""what ONE RANK runs when the crop list is sharded over 2 / 4 / 8 GPUs -- tower latency and per-kernel
launch times at 1 ... 40 crops (8 GPUs: 5 crops for config 5 / strong config 2, 9 for config 3; 4 GPUs: 10 / 17; 2 GPUs: 20 / 34),
one tower pass versus micro-batches of 3 (the chunked gather's old default), and the adapter on the image-owning rank.
Product library, bf16.  Writes JSON to stdout (-> profiles/r03_small_batch_latency.json)."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib, weights as W
from slime_amd.model.multimodal_encoder.clip_encoder import HipCLIPVisionModel
dev = torch.device("cuda:0"); dt = torch.bfloat16
lib = _lib.load_diag()          # same kernels as the product library + the per-shape tile override used in the last section
vm = HipCLIPVisionModel(W.CLIP_L_336)
vm.load_state_dict(W.make_tower_state_dict(W.CLIP_L_336, seed=1234))
vm.to(dev).to(dt)
px = W.synthetic_pixels(40, seed=0).to(dev).to(dt)


def t_ms(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


out = {"tower_ms": {}, "tower_ms_forced_one_stream": {}, "chunks_of_3_ms": {}, "kernels": {}}
for n in (1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 14, 16, 17, 20, 24, 34, 40):
    x = px[:n].contiguous()
    out["tower_ms"][n] = round(t_ms(lambda: vm.encode(x)), 3)
    vm.two_streams = False
    out["tower_ms_forced_one_stream"][n] = round(t_ms(lambda: vm.encode(x)), 3)
    vm.two_streams = True
    if n in (5, 9, 10, 17):
        parts = [px[i:min(i + 3, n)].contiguous() for i in range(0, n, 3)]
        out["chunks_of_3_ms"][n] = round(t_ms(lambda: [vm.encode(p) for p in parts]), 3)
    print(f"{n:3d} crops: encode {out['tower_ms'][n]:7.3f} ms  one stream {out['tower_ms_forced_one_stream'][n]:7.3f} ms"
          + (f"  3-crop micro-batches {out['chunks_of_3_ms'][n]:7.3f} ms" if n in out["chunks_of_3_ms"] else ""), file=sys.stderr, flush=True)

# per-kernel launch durations (probe events, mean over the 23 layers) at the per-rank shapes
LABEL = {1: ("qkv", 3072, 1024), 2: ("attention", 0, 0), 3: ("out_proj", 1024, 1024), 5: ("fc1", 4096, 1024), 6: ("fc2", 1024, 4096)}
pt = vm.packed(-2, 0)
for n in (5, 9, 10, 17, 20):
    x = px[:n].contiguous()
    names = ops.tower_kernel_names(pt, n)
    rec = {}
    for kid, (label, N, K) in LABEL.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record(); torch.cuda.synchronize()
        ms = []
        for layer in range(pt.layers_run):
            pt.probe = (layer, kid, e0, e1)
            ops.tower_forward(pt, x); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        pt.probe = None
        avg = sum(ms) / len(ms)
        fl = 4.0 * n * 16 * 577 * 577 * 64 if kid == 2 else 2.0 * n * 577 * N * K
        rec[label] = {"us": round(avg * 1e3, 1), "min_us": round(min(ms) * 1e3, 1), "tflops": round(fl / avg / 1e9), "kernel": names[kid]}
    out["kernels"][n] = rec
    print(f"{n} crops: " + "  ".join(f"{k} {v['us']} us ({v['tflops']} TF/s)" for k, v in rec.items()), file=sys.stderr, flush=True)
# small grids: the shipped dispatch (128x128 three-stage ring where the grid has at most one workgroup per CU) against the two-stage
# form of the same kernel forced per shape, and the direct-B kernel forced onto the small grids
out["forced_tiles_ms"] = {}
SH4 = [(1024, 4096), (1024, 1024), (3072, 1024), (4096, 1024)]
RULES = {"auto": {}, "fc2,out two-stage": {SH4[0]: 3, SH4[1]: 3}, "all four two-stage": {k: 3 for k in SH4},
         "all four three-stage": {k: 15 for k in SH4}, "all four direct-B 128": {k: 12 for k in SH4}}
for n in (1, 2, 3, 4, 5, 6, 8, 9, 10):
    x = px[:n].contiguous()
    row = {}
    for name, rules in RULES.items():
        lib.slime_gemm_set_shape_tile(0, 0, 0)
        for (N, K), tile in rules.items(): lib.slime_gemm_set_shape_tile(N, K, tile)
        row[name] = round(t_ms(lambda: vm.encode(x)), 3)
    lib.slime_gemm_set_shape_tile(0, 0, 0)
    out["forced_tiles_ms"][n] = row
    print(f"{n:3d} crops: " + "  ".join(f"{k} {v:.3f}" for k, v in row.items()), file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
