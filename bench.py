#!/usr/bin/env python3
"""Contract benchmark: image-crops/sec of the SliME visual hot path (ViT + projector) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One STEP = one pass of the hot path over one batch of synthetic input of BASELINE.json configs[1]:
8 images x (1 global + 4 local) 336x336 crops = 40 crops per GPU (weak scaling: the global batch is
8*N images), bf16 MFMA operands, inputs already resident in HBM:

    CLIP-ViT-L/14-336 tower over all crops of the rank (23 live layers)  ->  [N>1] RCCL all-gather of the
    bf16 tower features of all ranks  ->  gated global adapter (576 tokens/image), post_qformer local
    compression (144 tokens/crop) + MLP projector, spatial merge into LLM-ready token rows.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : the dominant kernel (the fused-epilogue MFMA GEMM; the heaviest launch shape of the
                 step), algorithmic FLOPs / live HIP-event duration, against the dense bf16 MFMA peak;
  cpu_baseline : the CPU oracle (oracle/slime_oracle.py, a restatement validated against the
                 reference) timed on this box's host cores on a bounded sample (N == 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IMAGES_PER_GPU = 8
LOCAL_CROPS = 4                        # 672x672 input -> 2x2 local grid
CROPS_PER_IMAGE = 1 + LOCAL_CROPS
GF_VIT_PER_CROP = 366.034              # live ViT path, 23 layers, S = 577
GF_GLOBAL_PER_IMAGE = 54.512           # gated adapter on the global view
GF_LOCAL_PER_CROP = 9.399              # post_qformer + MLP per local crop
PEAK_BF16_TFLOPS = 2500.0              # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run gather + adapter of a step on the tower's stream instead of a second stream")
    return ap.parse_args()


def event_time_ms(fn, iters=10, warm=2):
    """Average duration of ``fn`` (one kernel launch) with HIP events on the launching stream."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# kernel id in the tower driver (include/slime_hip.h: slime_probe) -> (label, rocprof kernel name, N, K)
PROBE_KERNELS = {
    1: ("qkv_proj", "gemm_w4_kernel<BF16, 0, 0, 6, 0>", 3072, 1024),
    3: ("out_proj+residual", "gemm_pp_kernel<BF16, 4, 0, 0, 4>", 1024, 1024),
    5: ("fc1+quick_gelu", "gemm_w4_kernel<BF16, 1, 0, 8, 0>", 4096, 1024),
    6: ("fc2+residual", "gemm_pp_kernel<BF16, 4, 1, 0, 4>", 1024, 4096),
    2: ("attention", "attn64_kernel<BF16>", 0, 0),
}


def kernel_roofline(vision_model, pixels_half, reps=4, layer=11):
    """Per-kernel launch durations at the step's launch shapes: HIP events recorded by the tower driver
    (slime_vit_forward_ex probe) on the launching stream, immediately around one kernel of one layer,
    during a tower pass over ONE of the two half batches with the other stream idle.  This is the
    quantity rocprofv3 --kernel-trace reports as the kernel's average duration (the tool serialises the
    two streams), so the two agree; inside the real step the two streams overlap and a kernel's wall
    duration is longer while it shares the CUs.  Returns (dominant GEMM id, {id: stats})."""
    from slime_amd import ops
    crops = pixels_half.shape[0]
    M = crops * 577
    pt = vision_model.packed(-2, 0)
    per = {}
    for kid, (label, name, N, K) in PROBE_KERNELS.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()                           # force creation of the HIP events
        pt.probe = (layer, kid, e0, e1)
        torch.cuda.synchronize()
        ms = []
        for _ in range(reps):
            ops.tower_forward(pt, pixels_half)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        pt.probe = None
        avg = sum(ms) / len(ms)
        fl = 4.0 * crops * 16 * 577 * 577 * 64 if kid == 2 else 2.0 * M * N * K
        per[kid] = {"label": label, "rocprof_name": name, "ms": round(avg, 4), "min_ms": round(min(ms), 4),
                    "tflops": round(fl / avg / 1e9, 1), "gflop_per_launch": round(fl / 1e9, 2), "M": M, "N": N, "K": K}
    dom = max((k for k in per if k != 2), key=lambda k: per[k]["ms"])
    return dom, per


def cpu_baseline(tower_sd, adapter_sd):
    """Oracle (CPU restatement of the reference) on one 1+4 image; a reported baseline, not a target."""
    from oracle import slime_oracle as O
    from slime_amd import weights as W
    px = W.synthetic_pixels(CROPS_PER_IMAGE, seed=7)
    tsd = W.strip_tower_prefix(tower_sd)
    best = None
    t_all = time.perf_counter()
    for _ in range(2):
        t0 = time.perf_counter()
        O.encode_image(tsd, adapter_sd, W.CLIP_L_336, W.ADAPTER_8B, px, (672, 672))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        if time.perf_counter() - t_all > 20:
            break
    return {"value": round(CROPS_PER_IMAGE / best, 3), "unit": "crops/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"1 image x (1+4) crops, fp32 torch CPU oracle (tower 23 layers + adapter + merge), best of <=2 runs, {best:.2f} s"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    import torch.distributed as dist
    # dry-run knobs for 1-GPU boxes (not used by the driver): SLIME_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and
    # SLIME_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device) -- the numbers mean nothing then, the
    # point is to execute the multi-process code path (rendezvous, gather, barrier, max-reduce, rank-0 print) for real
    if os.environ.get("SLIME_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("SLIME_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SLIME_BENCH_FORCE_COLLECTIVE=1: take the N>1 code path (RCCL init, all-gather, barrier, max-reduce) with a
    # single rank -- a self-test of that path on a 1-GPU box; the timed step then includes the 1-rank all-gather.
    collective = world > 1 or os.environ.get("SLIME_BENCH_FORCE_COLLECTIVE") == "1"
    # stdout carries exactly one line (the JSON): libraries that write banners to fd 1 (RCCL prints its version / library path
    # when a communicator is created) are sent to stderr until the result is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    from slime_amd import weights as W, ops
    from slime_amd.model.llava_arch import SlimeVisualEncoder, default_slime_config
    from slime_amd.dist import sharded_tower_gather

    dt = torch.bfloat16
    tower_sd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
    adapter_sd = W.make_adapter_state_dict(W.ADAPTER_8B, seed=4321)
    enc = SlimeVisualEncoder(default_slime_config("synthetic:1234"))
    enc.load_visual_state(tower_sd, adapter_sd)
    enc.to(dev)
    enc.get_vision_tower().vision_tower.to(dt)          # bf16 tower (training dtype of the reference, BASELINE cfg)
    model = enc.get_model()
    tower = enc.get_vision_tower()

    n_local = IMAGES_PER_GPU * CROPS_PER_IMAGE
    pixels = W.synthetic_pixels(n_local, seed=100 + rank).to(dev).to(dt)     # resident in HBM before timing
    split_sizes = [CROPS_PER_IMAGE] * IMAGES_PER_GPU
    image_sizes = [(672, 672)] * IMAGES_PER_GPU
    g = model.sampler.grid_size
    rows_per_image = 576 + LOCAL_CROPS * g * g
    pg = model.mm_projector.packed(dt)
    post = model.sampler.post_qformer.packed(576, dt)

    # Steps are independent batches, so the tail of step i (all-gather + adapter: small, partly under-filled launches,
    # and for N > 1 an xGMI transfer) is enqueued on its own stream and runs under the tower of step i+1; the timed
    # region ends with a device-wide synchronize, i.e. all K steps are complete inside it.
    tail_stream = None if args.no_pipeline else torch.cuda.Stream(device=dev)

    def tail(feats):
        if collective:
            allf = sharded_tower_gather(feats, world)                        # [40*G,576,1024] on every rank
            feats = allf[rank * n_local:(rank + 1) * n_local]
        # GatedBlock on the global crops + post_qformer / projection MLP / spatial merge on the local crops: one C-ABI
        # call (slime_adapter_forward), tokens [images, 576 + 4*144, 4096] bf16
        return ops.adapter_forward(pg, post, feats, IMAGES_PER_GPU, LOCAL_CROPS, 2, 2, True, -1, dt)

    def step():
        feats = tower(pixels)                                                # [40,576,1024] bf16
        if tail_stream is None:
            return tail(feats)
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(tail_stream):
            tail_stream.wait_event(ready)
            feats.record_stream(tail_stream)
            return tail(feats)

    def barrier():
        if collective:
            dist.barrier()

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(out.float()).all()
    if collective:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        crops_total = n_local * world * args.steps
        value = crops_total / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        step_gf = n_local * GF_VIT_PER_CROP + IMAGES_PER_GPU * GF_GLOBAL_PER_IMAGE + IMAGES_PER_GPU * LOCAL_CROPS * GF_LOCAL_PER_CROP
        path_tflops = step_gf * world / (elapsed / args.steps) / 1e3
        halves = 2 if tower.vision_tower.two_streams else 1
        dom, per = kernel_roofline(tower.vision_tower, pixels[: n_local // halves].contiguous())
        traffic = None
        # HBM bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE x2 correction + WRITE_SIZE,
        # profiles/README.md); null if that kernel is not in the committed summary
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_kernels.json")
        if os.path.isfile(pmc):
            try:
                rec = json.load(open(pmc)).get(per[dom]["rocprof_name"], {})
                if "hbm_read_bytes_corrected" in rec and "hbm_write_bytes" in rec:
                    traffic = int(rec["hbm_read_bytes_corrected"] + rec["hbm_write_bytes"])
            except Exception:
                traffic = None
        res = {
            "metric": "image-crops/sec (ViT+projector) at 336px, 1+4 grid",
            "value": round(value, 1), "unit": "crops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "CLIP-ViT-L/14-336, 8 images x (1 global + 4 local) 336px crops per GPU (672x672 inputs), "
                                   "tower + gated adapter + post_qformer + MLP projector + spatial merge",
                       "crops_per_gpu": n_local, "images_per_gpu": IMAGES_PER_GPU, "grid": "1+4",
                       "parallelism": f"crop-parallel dp{world}" + (" + all-gather of tower features" if world > 1 else ""),
                       "tower_streams": halves,
                       "step_pipelining": "none" if tail_stream is None else "gather+adapter of step i on a second stream, under the tower of step i+1"},
            "path_mfma": {"algorithmic_tflops": round(path_tflops, 1), "frac_of_peak": round(path_tflops / (PEAK_BF16_TFLOPS * world), 4),
                          "gflop_per_step_per_gpu": round(step_gf, 1)},
            "roofline": {"bound": "mfma",
                         "kernel": f"{per[dom]['rocprof_name']} ({per[dom]['label']}, M={per[dom]['M']} N={per[dom]['N']} K={per[dom]['K']})",
                         "achieved": per[dom]["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(per[dom]["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "launch_ms": per[dom]["ms"], "gflop_per_launch": per[dom]["gflop_per_launch"],
                         "method": "HIP events recorded by the tower driver (slime_vit_forward_ex probe) around the kernel of layer 11 on its "
                                   "launching stream, tower pass over one half batch (the step's launch shape), other stream idle; mean of 4",
                         "kernels": {v["label"]: {k: v[k] for k in ("rocprof_name", "ms", "min_ms", "tflops")} for v in per.values()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(tower_sd, adapter_sd)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(res), flush=True)
        os.dup2(2, 1)
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
