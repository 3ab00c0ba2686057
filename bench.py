#!/usr/bin/env python3
"""Contract benchmark: image-crops/sec of the SliME visual hot path (ViT + projector) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 1|2|3|4|5]
    N > 1 works both ways: started plainly (`python bench.py --gpus N ...`, no WORLD_SIZE in the environment) the script
    launches its own N ranks -- it re-runs itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free port>` and passes rank 0's line through --; started under torch.distributed.run
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) it is one of the ranks.

Default (= what the driver runs) is BASELINE.json configs[1] ("config 2"): one STEP = one pass of the hot path over
8 images x (1 global + 4 local) 336x336 crops = 40 crops per GPU (weak scaling: the global batch is 8*N images), bf16 MFMA
operands, inputs already resident in HBM:

    CLIP-ViT-L/14-336 tower over all crops of the rank (23 live layers)  ->  [N>1] RCCL all-gather of the bf16 tower
    features  ->  gated global adapter (576 tokens/image), post_qformer local compression (144 tokens/crop) + MLP
    projector, spatial merge into LLM-ready token rows.

--config 1  BASELINE configs[0] on the HIP path: ONE 336 x 336 crop, global view only, batch 1 (tower + GatedBlock): a latency line.
--config 3  BASELINE configs[2]: 4 images x (1 global + 16 local) = 68 crops, STRONG scaling: the crop list is block
            partitioned over the ranks (ceil(68/N) per rank, zero-padded: slime_amd.dist.sharded_tower), features are
            all-gathered (chunked, under the tower), the adapter (4 x 4 spatial merge) runs on the image-owning rank.
--config 4  BASELINE configs[3], one GPU: config 2's encode + the visual-token splice into 8 text sequences + the
            attention sub-layer of all 32 Llama-3-8B layers over the spliced sequences (slime_llama_attn_forward:
            q/k/v GEMM, RoPE, causal GQA attention, o_proj); reports the prefill-attention kernel's own roofline.

--config 5  BASELINE configs[4], the video path: 8 frames x (1+4) crops = 40 ViT forwards, STRONG scaling like config 3 (the
            frames' crops block partitioned over the ranks, chunked all-gather under the tower), adapter for all frames and the
            prefill of ONE sequence holding the 8 frames' visual tokens (64 text + 8 x 1152 = 9280 positions; 32 attention
            sub-layers) replicated on every rank -- the language model is not sharded on this path.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : the dominant kernel of the step, algorithmic FLOPs / live HIP-event duration, against the dense bf16 MFMA peak;
  cpu_baseline : the CPU oracle (oracle/slime_oracle.py, a restatement validated against the reference) timed on this box's
                 host cores on a bounded sample (N == 1, config 2 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A step uses the caller's stream, the tower's side
# stream, the tail stream and -- with N > 1 -- RCCL's; streams that share a queue serialise (profiles/r04_collective_hw_queues.txt:
# 2129 crops/s with 4 queues against 2447 with 8 on the forced-collective line).  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# dmabuf IPC (this pool's host driver supports nothing else): without it RCCL's hipIpcGetMemHandle fails between ranks.  Already
# exported on the boxes; set here too so that a rank started under a bare torch.distributed.run in a scrubbed environment has it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GF_VIT_PER_CROP = 366.034              # live ViT path, 23 layers, S = 577
GF_GLOBAL_PER_IMAGE_REFERENCE = 54.512 # gated adapter on the global view as the reference computes it: two complete experts, then the gate mix
GF_GLOBAL_PER_IMAGE = 35.185           # EXECUTED since round 4: the gate mixes the hidden rows (slime_gate_premix) and projection[2] runs once per
                                       # token (-2 x 576 x 4096 x 4096 flop per image); rates below are priced on the executed arithmetic
GF_LOCAL_PER_CROP = 9.399              # post_qformer + MLP per local crop
PEAK_BF16_TFLOPS = 2500.0              # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
# committed rocprofv3 --pmc summary the `traffic` figure is read from: the newest round's that exists (the line names file and commit)
PMC_FILE = next((f for f in ("r06_pmc_kernels.json", "r05_pmc_kernels.json") if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f))),
                "r06_pmc_kernels.json")
REPEATS = 3                            # the K-step region is timed this many times back to back; the headline is the FIRST (the driver-visible) one

CONFIGS = {                            # images per step, local crops per image, local grid
    1: dict(images=1, local=0, grid=(1, 1), scaling="weak"),     # BASELINE configs[0] on the HIP path: one 336 x 336 crop, global view only, batch 1 (a latency line)
    2: dict(images=8, local=4, grid=(2, 2), scaling="weak"),
    3: dict(images=4, local=16, grid=(4, 4), scaling="strong"),
    4: dict(images=8, local=4, grid=(2, 2), scaling="weak"),
    5: dict(images=8, local=4, grid=(2, 2), scaling="strong"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run gather + adapter of a step on the tower's stream instead of a second stream")
    ap.add_argument("--gather", choices=["auto", "chunked", "oneshot", "compressed"], default="auto",
                    help="strong-scaling exchange: auto = slime_amd.dist.choose_chunk (measured tower latency curve + transfer model: "
                         "one pass + one all-gather at every per-rank size of configs 2 / 3 / 5), chunked = micro-batches of 3 with "
                         "asynchronous gathers (round 2's default), oneshot, or compressed (post_qformer on the shard first)")
    ap.add_argument("--prefill-shard", choices=["replicated", "heads"], default="replicated",
                    help="config 5 with N > 1: replicated = every rank runs all 32 attention sub-layers (north_star's split: only the ViT "
                         "crops are sharded); heads = kv-head sharding (slime_amd.dist.head_sharded_attention: column-sliced q/k/v, local "
                         "attention, row-parallel o_proj + one all-reduce per layer) -- opt-in until measured on a multi-GPU box")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="config 2 only: strong = the SAME 40 crops block-partitioned over the ranks (what BASELINE's '1/2/4/8 MI355X' "
                         "reads as for a fixed batch); default weak = 40 crops per GPU")
    return ap.parse_args()


# kernel id in the tower driver (include/slime_hip.h: slime_probe) -> (label, N, K); attention has N = K = 0
PROBE_KERNELS = {1: ("qkv_proj", 3072, 1024), 3: ("out_proj+residual", 1024, 1024), 5: ("fc1+quick_gelu", 4096, 1024),
                 6: ("fc2+residual", 1024, 4096), 2: ("attention", 0, 0), 0: ("layernorm1", -1, 0), 4: ("layernorm2", -1, 0)}


def gemm_algorithmic_bytes(kid, M, N, K, split_residual=True):
    """Bytes one launch of a tower GEMM has to move if every operand crossed the fabric exactly once."""
    b = M * K * 2 + N * K * 2 + N * 4                       # A, W (16 bit), bias
    if kid in (1, 5):                                       # q/k/v, fc1: LayerNorm-fold consumer, T output
        return b + M * N * 2 + M * (K // 64) * 8 + N * 4    # + output, row partial sums, colsum
    if split_residual:                                      # out_proj / fc2 on the split residual stream: hi (T) + lo8 (one byte, ABI 7) read and written
        return b + 2 * M * N * (2 + 1) + M * (N // 64) * 8
    return b + 2 * M * N * 4 + M * N * 2 + M * (N // 64) * 8   # rounds 1-4: fp32 residual read + write, T(h), partial sums


class BoxSampler:
    """sclk / mclk / socket power / hot-spot temperature of this rank's GPU, sampled on a thread while the bench runs, so that a
    slow box can be told from a regression (boxes of this pool run the same kernels 8-9 % apart).  Source, in
    order of preference: amdsmi in-process (gpu_metrics table, ~0.1 ms per read), the amdgpu hwmon files in sysfs, `rocm-smi`
    as a subprocess (slow: 0.5 s period).  Every sample carries the phase label that was current when it was taken; `summary()`
    reduces them to mean / min / max per phase.  A box with no readable source yields {"source": None} -- never an exception."""

    FIELDS = ("sclk_mhz", "mclk_mhz", "power_w", "temp_c")

    def __init__(self, device_index=0, period=0.02):
        import threading
        self.period, self.phase, self.samples = period, "init", []
        self._stop, self._thread, self.source, self.note = threading.Event(), None, None, None
        self._read = self._open(device_index)
        if self._read is not None:
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()

    # -- sources ---------------------------------------------------------------------------------------------------------
    def _open(self, idx):
        for fn in (self._open_amdsmi, self._open_sysfs, self._open_rocm_smi):
            try:
                rd = fn(idx)
                if rd is not None and any(v is not None for v in rd().values()):
                    return rd
            except Exception as e:                               # a missing driver / permission: try the next source
                self.note = f"{fn.__name__}: {type(e).__name__}: {e}"[:160]
        return None

    @staticmethod
    def _num(x):
        if isinstance(x, (int, float)) and not isinstance(x, bool):
            return float(x)
        if isinstance(x, (list, tuple)):
            v = [float(y) for y in x if isinstance(y, (int, float)) and not isinstance(y, bool) and 0 < y < 65535]
            return sum(v) / len(v) if v else None
        return None

    def _open_amdsmi(self, idx):
        import amdsmi
        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        h = hs[idx if idx < len(hs) else 0]
        try:                                                     # rank -> GPU by PCI address when torch knows it (HIP_VISIBLE_DEVICES re-orders)
            want = torch.cuda.get_device_properties(idx).pci_bus_id
            for c in hs:
                bdf = amdsmi.amdsmi_get_gpu_device_bdf(c)
                if int(bdf.split(":")[1], 16) == int(want):
                    h = c
        except Exception:
            pass
        num = self._num

        def rd():
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            sclk = num(m.get("current_gfxclk"))
            if sclk is None or not (0 < sclk < 65535):
                sclk = num(m.get("current_gfxclks"))
            pw = num(m.get("current_socket_power"))
            if pw is None or not (0 < pw < 65535):
                pw = num(m.get("average_socket_power"))
            return {"sclk_mhz": sclk, "mclk_mhz": num(m.get("current_uclk")), "power_w": pw, "temp_c": num(m.get("temperature_hotspot"))}
        self.source = "amdsmi gpu_metrics (in-process)"
        return rd

    def _open_sysfs(self, idx):
        import glob
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
        if not cards:
            return None
        hw = cards[idx if idx < len(cards) else 0]

        def f(name, scale):
            try:
                return float(open(os.path.join(hw, name)).read().strip()) * scale
            except (OSError, ValueError):
                return None

        def rd():
            pw = f("power1_input", 1e-6)
            return {"sclk_mhz": f("freq1_input", 1e-6), "mclk_mhz": f("freq2_input", 1e-6),
                    "power_w": pw if pw is not None else f("power1_average", 1e-6), "temp_c": f("temp2_input", 1e-3)}
        self.source = f"sysfs {hw}"
        return rd

    def _open_rocm_smi(self, idx):
        import re
        import shutil
        import subprocess
        if not shutil.which("rocm-smi"):
            return None
        self.period = max(self.period, 0.5)

        def rd():
            o = subprocess.run(["rocm-smi", "-d", str(idx), "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout

            def grab(pat):
                m = re.search(pat, o)
                return float(m.group(1)) if m else None
            return {"sclk_mhz": grab(r"sclk clock level: \d+: \((\d+)Mhz\)"), "mclk_mhz": grab(r"mclk clock level: \d+: \((\d+)Mhz\)"),
                    "power_w": grab(r"Power \(W\): ([0-9.]+)"), "temp_c": grab(r"junction\) \(C\): ([0-9.]+)")}
        self.source = "rocm-smi (subprocess, 0.5 s period)"
        return rd

    # -- sampling --------------------------------------------------------------------------------------------------------
    def _loop(self):
        while not self._stop.is_set():
            try:
                self.samples.append((self.phase, time.perf_counter(), self._read()))
            except Exception:
                pass
            self._stop.wait(self.period)

    def mark(self, phase):
        self.phase = phase

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=5)

    def phase_stats(self, phase):
        rows = [v for ph, _, v in self.samples if ph == phase]
        out = {"samples": len(rows)}
        for k in self.FIELDS:
            xs = [r[k] for r in rows if r.get(k) is not None]
            if xs:
                out[k] = {"mean": round(sum(xs) / len(xs), 1), "min": round(min(xs), 1), "max": round(max(xs), 1)}
        return out

    def summary(self, phases):
        if self._read is None:
            return {"source": None, "note": self.note or "no amdsmi / hwmon / rocm-smi on this box"}
        out = {"source": self.source, "period_s": self.period, **{ph: self.phase_stats(ph) for ph in phases}}
        t = out.get("timed", {})
        for k in ("sclk_mhz", "power_w", "temp_c", "mclk_mhz"):      # the flat keys the VERDICT asks for
            if k in t:
                out[k + "_timed"] = t[k]["mean"]
        return out


def mfma_stream_calibration(dev, dt, box=None, seconds=0.3):
    """THIS box's bare MFMA stream, measured in this run (slime_mfma_stream_probe: register-resident random operands, two waves per
    SIMD, nothing but v_mfma in the loop): the rate the matrix pipes deliver under the 1400 W cap right now, with the clocks / power
    the sampler saw meanwhile.  ~`seconds` of GPU time, HIP events on the launching stream."""
    import ctypes as C
    from slime_amd import ops, _lib
    lib = _lib.load()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    g = torch.Generator(device="cpu").manual_seed(77)
    operands = torch.randn(cus * 8 * 8192, generator=g).to(dt).to(dev)               # 16 KiB of fresh random values per wave
    out = torch.empty(cus * 512, dtype=torch.float32, device=dev)
    fl = C.c_double(0.0)
    st = torch.cuda.current_stream().cuda_stream

    def launch(iters):
        _lib.check(lib.slime_mfma_stream_probe(ops.dtype_code(dt), iters, operands.data_ptr(), operands.numel() * 2, out.data_ptr(), out.numel() * 4,
                                               C.byref(fl), st), "mfma_stream_probe")
    iters = 4000
    launch(iters)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(iters); e1.record()
    torch.cuda.synchronize()
    ms1 = e0.elapsed_time(e1)
    n = max(3, int(seconds * 1e3 / max(ms1, 1e-3)))
    if box:
        box.mark("calibration")
    per = []
    for _ in range(n):                                           # per-launch events: the first launches run before the chip settles at the cap
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); launch(iters); b.record()
        per.append((a, b))
    torch.cuda.synchronize()
    if box:
        box.mark("after_calibration")
    tf = [fl.value / (a.elapsed_time(b) * 1e-3) / 1e12 for a, b in per]
    settled = tf[len(tf) // 2:]                                  # second half: at the power cap
    assert torch.isfinite(out).all()
    return {"tflops": round(sum(settled) / len(settled), 1), "tflops_first_launch": round(tf[0], 1), "tflops_min": round(min(tf), 1), "tflops_max": round(max(tf), 1),
            "launches": n, "ms_per_launch": round(ms1, 3), "gflop_per_launch": round(fl.value / 1e9, 1),
            "what": "slime_mfma_stream_probe: bare v_mfma_f32_16x16x32_bf16 stream, register-resident random operands (16 KiB per wave), 8 waves per CU, no memory "
                    "instruction in the loop; mean of the second half of the launches (the chip has settled at its power cap)",
            **({"box": box.phase_stats("calibration")} if box else {})}


def kernel_roofline(vision_model, pixels_half, reps=1):
    """Per-kernel launch durations at the step's launch shapes: HIP events recorded by the tower driver
    (slime_vit_forward_ex probe) on the launching stream, immediately around one kernel, during a tower pass over ONE of
    the two half batches with the other stream idle -- once per LAYER (all layers that run), so the figures are the mean /
    min / max over the 23 launches of a pass, not one favourable layer.  This is the quantity rocprofv3 --kernel-trace reports
    as the kernel's average duration (with the two streams serialised); inside the real step the two streams overlap and a
    kernel's wall duration is longer while it shares the CUs.  Returns (dominant GEMM id, {id: stats})."""
    from slime_amd import ops
    crops = pixels_half.shape[0]
    M = crops * 577
    pt = vision_model.packed(-2, 0)
    names = ops.tower_kernel_names(pt, crops)
    per = {}
    for kid, (label, N, K) in PROBE_KERNELS.items():
        if kid not in names:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()                           # force creation of the HIP events
        torch.cuda.synchronize()
        ms = []
        for layer in range(pt.layers_run):
            pt.probe = (layer, kid, e0, e1)
            for _ in range(reps):
                ops.tower_forward(pt, pixels_half)
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1))
        pt.probe = None
        avg = sum(ms) / len(ms)
        if kid == 2:
            fl = 4.0 * crops * 16 * 577 * 577 * 64
        elif N < 0:
            fl = 0.0
        else:
            fl = 2.0 * M * N * K
        per[kid] = {"label": label, "rocprof_name": names[kid], "ms": round(avg, 4), "min_ms": round(min(ms), 4), "max_ms": round(max(ms), 4),
                    "launches_timed": len(ms), "tflops": round(fl / avg / 1e9, 1), "gflop_per_launch": round(fl / 1e9, 2), "M": M, "N": N, "K": K}
        if N > 0:
            split = ops._lib.load().slime_vit_residual_epilogue() == ops._lib.EPI_BIAS_RESID_SPLIT_LN
            per[kid]["algorithmic_bytes"] = gemm_algorithmic_bytes(kid, M, N, K, split)
        if N < 0:
            per[kid]["gb_per_s"] = round(M * 1024 * 6 / avg / 1e6, 1)          # fp32 in + 16-bit out
    dom = max((k for k in per if per[k]["N"] > 0), key=lambda k: per[k]["ms"])
    return dom, per


def cpu_baseline(tower_sd, adapter_sd, crops_per_image):
    """Oracle (CPU restatement of the reference) on one 1+4 image; a reported baseline, not a target."""
    from oracle import slime_oracle as O
    from slime_amd import weights as W
    px = W.synthetic_pixels(crops_per_image, seed=7)
    tsd = W.strip_tower_prefix(tower_sd)
    best, ref, best_threads = None, None, None
    t_all = time.perf_counter()
    all_threads = torch.get_num_threads()
    # a 128-thread pool is not the fastest way to run these medium-sized fp32 GEMMs on the host: give the baseline its best thread
    # count (the reference's own CPU path measured 1.35-1.64 crops/s on 8 cores, profiles/r02_reference_cpu_timing.json)
    for threads in [t for t in (all_threads, 32, 16) if t <= all_threads]:
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        ref = O.encode_image(tsd, adapter_sd, W.CLIP_L_336, W.ADAPTER_8B, px, (672, 672))
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, best_threads = dt, threads
        if time.perf_counter() - t_all > 25:
            break
    torch.set_num_threads(all_threads)
    return {"value": round(crops_per_image / best, 3), "unit": "crops/s", "cores": best_threads,
            "kind": "port", "sample": f"1 image x (1+4) crops, fp32 torch CPU oracle (tower 23 layers + adapter + merge), best thread count of "
                                      f"{all_threads} / 32 / 16 (one run each, <= 25 s in total): {best_threads} threads, {best:.2f} s"}, px, ref


def parity_vs_oracle(tower_sd, adapter_sd, px, ref, dev, nw, nh):
    """north_star's accuracy target, measured in this run (outside the timed region) on the cpu_baseline sample: the product
    path (tower + fused adapter, C ABI) against the fp32 oracle's projector outputs, for the bench dtype (bf16) and for the
    reference's inference dtype (fp16, llava/model/builder.py:43)."""
    from slime_amd import ops, weights as W
    A = W.ADAPTER_8B
    tsd = W.strip_tower_prefix(tower_sd)
    out = {"sample": "the cpu_baseline image: 1 global + 4 local crops, ViT-L/14-336 tower + gated adapter + post_qformer + MLP + 2x2 merge; "
                     "rel-L2 of the projector outputs vs the fp32 CPU oracle", "target_rel_l2": 1e-3}

    def rel(a, b):
        a, b = a.double().cpu(), b.double()
        return float((a - b).norm() / b.norm())
    for dt, key in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        pt = ops.pack_tower(tsd, W.CLIP_L_336, dt, dev)
        pg = ops.pack_gated(W.sub_state(adapter_sd, "mm_projector."), A, dt, dev)
        post = ops.pack_resampler(W.sub_state(adapter_sd, "sampler.post_qformer."), 1024, 8, 576, dt, dev, A.ln_eps)
        feats = ops.tower_forward(pt, px.to(dev), out_dtype=dt)
        tok = ops.adapter_forward(pg, post, feats, 1, px.shape[0] - 1, nw, nh, True, -1, torch.float32)[0]
        torch.cuda.synchronize()
        out[key] = {"rel_l2_global": round(rel(tok[:576], ref["global"]), 6), "rel_l2_local": round(rel(tok[576:], ref["merged"]), 6),
                    "rel_l2_tower": round(rel(feats.float(), ref["tower"]), 6)}
        del pt, pg, post
    out["dtype"] = "fp16 meets the 1e-3 target; bf16 (this line's dtype) is bounded by its 2^-9 operand rounding" \
        if max(out["fp16"]["rel_l2_global"], out["fp16"]["rel_l2_local"]) <= 1e-3 else "see figures"
    return out


def pmc_profile_is_current():
    """Do the committed PMC passes describe the kernel sources the running library was built from?  True / False by the digest
    tools/summarize_prof.py stamped into the summary (slime_amd._lib.csrc_digest); None for a summary without one (rounds <= 5)."""
    from slime_amd import _lib
    try:
        sha = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE))).get("_meta", {}).get("csrc_sha")
    except (OSError, ValueError):
        return None
    return None if not sha else sha == _lib.csrc_digest()


def pmc_traffic(rocprof_name, profiled_shape=True):
    """HBM bytes per launch of a kernel from the COMMITTED PMC passes (FETCH_SIZE x2 correction + WRITE_SIZE, profiles/README.md).
    It is a profile of this kernel at this shape, not a measurement of this run (PMC counters need rocprofv3 around the process):
    the bench line says so in `traffic_source` and names the commit the passes ran on in `traffic_head`.  Returns
    (bytes | None, source, head, error | None): a dominant kernel that is missing from the committed summary (another CU count,
    a changed tile rule or template argument) leaves `traffic` null AND sets `traffic_error` -- the line is still printed, and
    tests/test_gpu_bench_contract.py asserts that the default line carries no such error."""
    pmc = os.path.join(ROOT, "profiles", PMC_FILE)
    src = "profiles/" + PMC_FILE
    try:
        table = json.load(open(pmc))
    except (OSError, ValueError) as e:
        return None, src, None, f"cannot read {src}: {e}"
    meta = table.get("_meta", {})
    head = meta.get("git_head")
    rec = table.get(rocprof_name)
    if rec is None:                                       # the summary keys some kernels with their variant / grid ("prefill32_kernel<BF16, 6> [grid 65536]")
        stem = rocprof_name.rstrip(">")
        hits = [k for k in table if k.startswith(stem)]
        rec = table[hits[0]] if len(hits) == 1 else None
    if rec is None or "hbm_read_bytes_corrected" not in rec or "hbm_write_bytes" not in rec:
        if not profiled_shape:  # launch shapes tools/pmc_target.py does not profile (config 3's 34-crop halves, rank shards): say so
            return None, f"not profiled: {src} holds the default step's launch shapes (20-crop half batches) only", head, None
        return None, src, head, (f"{src} has no HBM-traffic record for the dominant kernel {rocprof_name!r}: re-run tools/run_pmc.sh + "
                                 "tools/summarize_prof.py and commit the summary")
    return int(rec["hbm_read_bytes_corrected"] + rec["hbm_write_bytes"]), src, head, None


WORKLOADS = {1: "CLIP-ViT-L/14-336, ONE 336x336 crop (global view only, batch 1: BASELINE configs[0] on the HIP path), tower + GatedBlock on the global view -- a latency line: 117 dependent launches on a chip they cannot fill",
            2: "CLIP-ViT-L/14-336, 8 images x (1 global + 4 local) 336px crops per GPU (672x672 inputs), tower + gated adapter + post_qformer + MLP projector + spatial merge",
            3: "CLIP-ViT-L/14-336, 4 images x (1 global + 16 local) 336px crops = 68 crops in total, crops block-partitioned over the GPUs, all-gather, gated adapter + post_qformer + MLP projector + 4x4 spatial merge on the image-owning rank",
            4: "SliME-8B prefill: config-2 encode (40 crops) + visual-token splice into 8 sequences + the attention sub-layer (q/k/v GEMM, RoPE, causal GQA 32q/8kv dh128, o_proj) of 32 Llama-3-8B layers",
            5: "video path: 8 frames x (1+4) crops = 40 ViT forwards block-partitioned over the GPUs, all-gather, adapter for all frames, visual-token splice into ONE 9280-position sequence + the attention sub-layer of 32 Llama-3-8B layers (replicated per rank)"}


class DeadlineGuard:
    """A hang in the SECOND curve must not cost the first.  The `strong` leg of the default N > 1 line (block-partitioned crops,
    chunked asynchronous all-gathers) has only ever run as a 2-rank rehearsal on one GPU -- no multi-GPU box was available in six
    rounds -- and it runs BEFORE rank 0 prints, because the contract is ONE JSON line.  Every rank arms this timer around the leg;
    if the leg has not returned after `seconds` the timer thread of rank 0 prints the headline it already holds (timed, max over
    ranks, complete) with `strong: {"error": "timeout ..."}` and no probe objects (their GPU work could queue behind the stuck
    collective), and every rank leaves with os._exit(0) -- a process stuck inside a collective cannot be unwound any other way.
    A leg that returns, or raises, cancels the timer: the normal path is untouched."""

    def __init__(self, seconds, on_timeout):
        import threading
        self.seconds, self.fired = seconds, False
        self._on_timeout = on_timeout
        self._timer = threading.Timer(seconds, self._fire)
        self._timer.daemon = True

    def _fire(self):
        self.fired = True
        try:
            self._on_timeout(self.seconds)
        finally:
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)

    def __enter__(self):
        self._timer.start()
        return self

    def __exit__(self, *exc):
        self._timer.cancel()
        return False


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no rank environment: start the N ranks here (one process per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 and a free port -- the container hostname may not resolve) and hand the
    children's stdout (rank 0's JSON line) and exit status through.  The reference has no counterpart: its multi-GPU driver is
    one independent process per GPU (/root/reference/scripts/llama/eval/textvqa.sh:17-26)."""
    import socket
    import subprocess
    single = os.environ.get("SLIME_BENCH_SINGLE_DEVICE") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not single and have < n:
        raise SystemExit(f"bench.py --gpus {n}: this node shows {have} GPU(s) (set SLIME_BENCH_SINGLE_DEVICE=1 SLIME_BENCH_BACKEND=gloo "
                         "to rehearse the multi-process path on one device)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("GPU_MAX_HW_QUEUES", "8")                # tower side stream + tail stream + RCCL's: see the top of this file
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL's hipIpcGetMemHandle fails without it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    env["SLIME_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    C = dict(CONFIGS[args.config])
    if args.scaling is not None:
        if args.config != 2:
            raise SystemExit("--scaling applies to --config 2 (configs 3 / 5 are strong, config 4 is single-GPU)")
        C["scaling"] = args.scaling
    IMAGES, LOCAL, (NW, NH) = C["images"], C["local"], C["grid"]
    CPI = 1 + LOCAL
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))             # one rank per GPU, started by this process
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.config == 4 and world > 1:
        raise SystemExit("--config 4 is the single-GPU prefill configuration (BASELINE configs[3]); the sharded one is --config 5")
    if args.config == 1 and world > 1:
        raise SystemExit("--config 1 is one crop on one GPU (BASELINE configs[0])")
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
    from slime_amd import dist as D

    dt = torch.bfloat16
    tower_sd = W.make_tower_state_dict(W.CLIP_L_336, seed=1234)
    adapter_sd = W.make_adapter_state_dict(W.ADAPTER_8B, seed=4321)
    enc = SlimeVisualEncoder(default_slime_config("synthetic:1234"))
    enc.load_visual_state(tower_sd, adapter_sd)
    enc.to(dev)
    enc.get_vision_tower().vision_tower.to(dt)          # bf16 tower (training dtype of the reference, BASELINE cfg)
    model = enc.get_model()
    tower = enc.get_vision_tower()
    g = model.sampler.grid_size
    pg = model.mm_projector.packed(dt)
    post = model.sampler.post_qformer.packed(576, dt)
    n_step = IMAGES * CPI                                # crops per step per GPU (config 2/4) or in total (config 3)
    strong = C["scaling"] == "strong"
    extra_cfg = {}
    box = BoxSampler(local_rank)                         # clocks / power / temperature of this rank's GPU, sampled from here on
    box.mark("setup")

    # Steps are independent batches, so the tail of step i (all-gather + adapter: small, partly under-filled launches,
    # and for N > 1 an xGMI transfer) is enqueued on its own stream and runs under the tower of step i+1; the timed
    # region ends with a device-wide synchronize, i.e. all K steps are complete inside it.
    tail_stream = None if args.no_pipeline else torch.cuda.Stream(device=dev)
    cur = {}                                             # what the weak step runs on (re-pointed for the fp16 leg)

    def weak_step_fns():
        """Weak scaling: 40 crops per GPU, every rank owns whole images."""
        pixels = W.synthetic_pixels(n_step, seed=100 + rank).to(dev).to(dt)     # resident in HBM before timing
        cur.update({"pixels": pixels, "pg": pg, "post": post, "dt": dt, "tower": tower})

        def tail(feats):
            if collective:
                # weak scaling: every rank owns whole images, so the adapter needs no remote features.  north_star asks for the
                # all-gather that reassembles the visual tokens on every rank (the LLM consumes all of them); it is issued
                # asynchronously and overlaps the adapter, which works on the rank's own block.  Nothing on this path READS the
                # gathered tensor (`gather_consumed: false` in the line): the data flow north_star describes -- gather BEFORE the
                # adapter -- is the strong form, timed in the same run as the `strong` object.
                allf = torch.empty((world * feats.shape[0],) + tuple(feats.shape[1:]), dtype=feats.dtype, device=feats.device)
                work = dist.all_gather_into_tensor(allf, feats.contiguous(), async_op=True)
            out = ops.adapter_forward(cur["pg"], cur["post"] if LOCAL else None, feats, IMAGES, LOCAL, NW, NH, LOCAL > 0, -1, cur["dt"])
            if collective:
                work.wait()
            return out

        def produce():
            return cur["tower"](cur["pixels"])                                   # [40,576,1024] bf16
        cfg = {"gather": "async all-gather of the rank's bf16 tower features, overlapping the adapter (overhead-only in weak scaling: each rank's adapter "
                         "reads its own block)" if collective else "none"}
        if collective:
            cfg["gather_consumed"] = False
        return pixels, produce, tail, cfg

    def strong_step_fns(gather_mode):
        """Strong scaling (config 3 / 5, config 2 with --scaling strong, and the `strong` object of the default N > 1 line): the
        identical crop list on every rank; rank r encodes its block, the exchange reassembles what the adapter needs BEFORE the
        adapter (north_star / SURVEY 8e), the adapter runs for the images this rank owns."""
        pixels = W.synthetic_pixels(n_step, seed=100).to(dev).to(dt)
        vm = tower.vision_tower
        # config 5: every rank needs every frame's tokens for the (replicated) prefill, so the adapter runs for all of them
        my_images = list(range(IMAGES)) if args.config == 5 else D.image_shard(IMAGES, world, rank)
        lo_i, hi_i = (my_images[0], my_images[-1] + 1) if my_images else (0, 0)
        per_rank_crops = -(-n_step // world)
        chunk = {"chunked": 3, "auto": D.choose_chunk(per_rank_crops, world, device_name=D.device_label(dev),
                                                      model="CLIP-ViT-L/14-336", dtype="bf16")}.get(gather_mode, 0)

        def tower_fn(x):
            return vm.encode(x, -2, False, dt)

        def compress_fn(x):
            return ops.resampler_forward(post, x.float(), want_t=True)[1]

        def produce():
            if gather_mode == "compressed":
                return D.sharded_tower_compressed(tower_fn, compress_fn, pixels, CPI, (576, 1024), g * g, dt)
            return D.sharded_tower(tower_fn, pixels, (576, 1024), dt, chunk=chunk)

        def tail(feats):
            if hi_i == lo_i:
                return None
            if gather_mode == "compressed":
                glob, comp = feats
                return ops.adapter_forward_precompressed(pg, glob[lo_i:hi_i], comp[lo_i * LOCAL:hi_i * LOCAL], hi_i - lo_i, LOCAL,
                                                         NW, NH, True, -1, dt)
            return ops.adapter_forward(pg, post, feats[lo_i * CPI:hi_i * CPI], hi_i - lo_i, LOCAL, NW, NH, True, -1, dt)
        full_b, comp_b = D.gather_bytes(n_step, CPI, world)
        cfg = {"gather": gather_mode, "gather_chunk": chunk, "gather_bytes_per_rank": comp_b if gather_mode == "compressed" else full_b,
               "crops_per_rank_padded": -(-n_step // world), "images_owned_by_rank0": len(my_images), "gather_consumed": True}
        return pixels, produce, tail, cfg

    pixels, produce, tail, step_cfg = strong_step_fns(args.gather) if strong else weak_step_fns()
    extra_cfg.update(step_cfg)

    # config 4: splice + 32 Llama-3-8B attention sub-layers over the spliced sequences
    prefill = None
    if args.config == 4:
        prefill = build_prefill(enc, dev, dt, IMAGES, 576 + LOCAL * g * g)
    elif args.config == 5:
        prefill = build_prefill(enc, dev, dt, IMAGES, 576 + LOCAL * g * g, sequences=1,
                                shard=(world, rank) if (args.prefill_shard == "heads" and collective and world > 1) else None)
        extra_cfg["prefill_shard"] = args.prefill_shard if world > 1 else "n/a (one GPU)"

    def make_step(produce_fn, tail_fn, with_prefill=True):
        def step():
            feats = produce_fn()
            if tail_stream is None:
                out = tail_fn(feats)
                return prefill(out) if (prefill and with_prefill) else out
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(tail_stream):
                tail_stream.wait_event(ready)
                for t in (feats if isinstance(feats, tuple) else (feats,)):
                    t.record_stream(tail_stream)
                out = tail_fn(feats)
                return prefill(out) if (prefill and with_prefill) else out
        return step

    step = make_step(produce, tail)

    def barrier():
        if collective:
            dist.barrier()

    def timed_region(step_fn=None, warmup=None, phase="timed"):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides."""
        step_fn = step_fn or step
        out = None
        box.mark("warmup")
        for _ in range(args.warmup if warmup is None else warmup):
            out = step_fn()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        box.mark(phase)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step_fn()
        torch.cuda.synchronize()
        barrier()
        dt_s = time.perf_counter() - t0
        box.mark("between")
        if out is not None:
            assert torch.isfinite(out.float()).all()
        return dt_s

    def max_over_ranks(sec):
        """(max over ranks, every rank's own figure) of a region's wall time."""
        if not collective:
            return sec, [sec]
        t = torch.tensor([sec], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)                            # each rank's own clock around the same K steps
        return max(float(e.item()) for e in every), [float(e.item()) for e in every]

    box.mark("idle")
    time.sleep(0.15)                                         # a few samples of the idle chip (context up, weights resident, nothing running)
    elapsed, every = max_over_ranks(timed_region())          # THE headline: warm-up, then the first K steps
    rank_ms = [e / args.steps * 1e3 for e in every]
    # the same K steps twice more, back to back, no warm-up (the chip is warm): run-to-run spread inside one process on one box
    repeats_s = [elapsed] + [max_over_ranks(timed_region(warmup=0, phase="repeat"))[0] for _ in range(REPEATS - 1)]
    box_ranks = None
    if collective:
        # every rank's clocks / power inside the headline region (a rank on a throttled GPU is the one the MAX over ranks reports)
        mine = box.phase_stats("timed")
        t = torch.tensor([mine.get(k, {}).get("mean", float("nan")) for k in ("sclk_mhz", "power_w", "temp_c")], device=dev, dtype=torch.float64)
        allb = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allb, t)
        box_ranks = [{k: (None if v != v else round(v, 1)) for k, v in zip(("sclk_mhz", "power_w", "temp_c"), x.tolist())} for x in allb]
        # what the communicator itself says (not the environment): backend name and the number of ranks in it
        extra_cfg["collective_backend"] = dist.get_backend()
        extra_cfg["rccl_ranks"] = dist.get_world_size() if dist.get_backend() == "nccl" else None
        extra_cfg["launched_by"] = "bench.py self-launch" if os.environ.get("SLIME_BENCH_SELF_LAUNCHED") == "1" else "torch.distributed.run"

    # The default N > 1 line is weak scaling (40 crops per GPU).  north_star's data flow -- the batch's crops sharded over the GPUs,
    # an RCCL all-gather reassembling the features BEFORE the adapter -- is the strong form: timed here, in the same process group and
    # the same run, on the same 40 crops for every N, so that one driver command returns both curves.
    def time_strong():
        s_pixels, s_produce, s_tail, s_cfg = strong_step_fns(args.gather)
        s_elapsed, s_every = max_over_ranks(timed_region(make_step(s_produce, s_tail), phase="strong"))
        pred_n, pred_1 = D.predicted_step_ms(n_step, CPI, world), D.predicted_step_ms(n_step, CPI, 1)
        return {"what": "the SAME 8 x (1+4) = 40 crops block-partitioned over the ranks (slime_amd.dist.sharded_tower), all-gather of the bf16 tower "
                        "features BEFORE the adapter, adapter on the image-owning rank; same process group, same run, W warm-up + K timed steps, max over ranks",
                "scaling": "strong", "ms_per_step": round(s_elapsed / args.steps * 1e3, 3), "value": round(n_step * args.steps / s_elapsed, 1),
                "unit": "crops/s (whole job)", "ms_per_step_rank_min": round(min(s_every) / args.steps * 1e3, 3),
                "ms_per_step_rank_max": round(max(s_every) / args.steps * 1e3, 3),
                "predicted_ms_per_step": round(pred_n, 2), "predicted_ms_per_step_n1": round(pred_1, 2),
                "speedup_vs_n1_predicted": round(pred_1 / pred_n, 2),
                "prediction": "slime_amd.dist.predicted_step_ms: tower latency profile at ceil(40 / N) crops + all-gather transfer model + adapter for the rank's "
                              "images (DESIGN section 7)",
                **s_cfg}

    def headline_fields():
        """The contract fields of the line: everything the timed regions above produced, no GPU work."""
        crops_total = n_step * (1 if strong else world) * args.steps
        return {
            "metric": {1: "image-crops/sec (ViT+projector) at 336px, single crop (global only, batch 1)",
                       3: "image-crops/sec (ViT+projector) at 336px, 1+16 grid"}.get(args.config, "image-crops/sec (ViT+projector) at 336px, 1+4 grid"),
            "value": round(crops_total / elapsed, 1), "unit": "crops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "ms_per_step_rank_min": round(min(rank_ms), 3), "ms_per_step_rank_max": round(max(rank_ms), 3),
            "ms_per_step_repeats": [round(r / args.steps * 1e3, 3) for r in repeats_s],
            "higher_is_better": True, "scaling": C["scaling"], "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic"}

    def strong_leg_timed_out(seconds):
        """DeadlineGuard's timer thread, every rank: rank 0 prints the headline it holds, without the probe objects."""
        if rank != 0:
            time.sleep(2.0)                                  # rank 0's timer was armed within milliseconds of this one: let it print first
            return
        msg = (f"timeout: the strong leg did not return within {seconds:.0f} s (a stuck collective?); the headline above is complete -- it was "
               "timed before the leg started; roofline / path_mfma / box were not probed because their GPU work could queue behind the stuck leg")
        res = {**headline_fields(),
               "config": {"workload": WORKLOADS[args.config], "baseline_config": args.config, "crops_per_gpu": n_step,
                          "images_per_step": IMAGES * world, "grid": f"1+{LOCAL}", "parallelism": f"crop-parallel dp{world} + all-gather of tower features", **extra_cfg},
               "roofline": None, "cpu_baseline": None, "strong": {"error": msg}}
        os.dup2(saved_stdout, 1)
        print(json.dumps(res), flush=True)

    strong_obj = None
    if collective and args.config == 2 and not strong:
        guard = DeadlineGuard(float(os.environ.get("SLIME_BENCH_STRONG_DEADLINE_S", "240")), strong_leg_timed_out)
        try:
            with guard:
                strong_obj = time_strong()
        except Exception as e:                               # the headline line is printed whatever happens to the second curve
            strong_obj = {"error": f"{type(e).__name__}: {e}"[:400]}
        if guard.fired:                                      # (a peer that left can make the stuck collective raise here) the timer thread
            time.sleep(3600)                                 # is printing the headline and ends this process: stay out of its way

    if rank == 0:
        box.mark("probe")
        step_gf = n_step * GF_VIT_PER_CROP + IMAGES * GF_GLOBAL_PER_IMAGE + IMAGES * LOCAL * GF_LOCAL_PER_CROP
        if prefill:
            step_gf += prefill.gflop
        step_gf_reference = step_gf + IMAGES * (GF_GLOBAL_PER_IMAGE_REFERENCE - GF_GLOBAL_PER_IMAGE)
        path_tflops = step_gf * (1 if strong else world) / (elapsed / args.steps) / 1e3
        halves = 2 if tower.vision_tower.two_streams else 1
        per_rank = -(-n_step // world) if strong else n_step
        n_half = (per_rank + 1) // halves if per_rank >= 8 else per_rank          # encode() splits from 8 crops on
        dom, per = kernel_roofline(tower.vision_tower, pixels[:n_half].contiguous())
        roof_kernel = per[dom]
        if prefill:
            pk = prefill.kernel_times()
            per.update(pk)
            roof_kernel = pk["prefill_attention"]
        # the driver's line (config 2, 40 crops per GPU) must carry a traffic figure; other launch shapes carry one if profiled
        calib = mfma_stream_calibration(dev, dt, box)
        traffic, traffic_src, traffic_head, traffic_err = pmc_traffic(roof_kernel["rocprof_name"], profiled_shape=(args.config == 2 and not strong and world == 1))
        workload = WORKLOADS[args.config]
        res = {
            **headline_fields(),
            "config": {"workload": workload, "baseline_config": args.config,
                       "crops_per_gpu": per_rank, "images_per_step": IMAGES * (1 if strong else world), "grid": f"1+{LOCAL}",
                       "parallelism": f"crop-parallel dp{world}" + (" + all-gather of tower features" if world > 1 else ""),
                       "tower_streams": halves,
                       # the tower's residual stream between the layers: T(h), which is the next GEMM's operand, + one signed byte per element (ABI 7)
                       "residual_stream": {"layout": "hi = T(h) [16 bit] + lo8 [int8]", "bytes_per_element_per_update": 6, "abi": 7},
                       # resident packed weights (each tensor once): since ABI 5 a weight lives in HBM as its fragment-order image only
                       "packed_weight_MB": {"tower": round(ops.packed_weight_bytes(tower.vision_tower.packed(-2, 0)) / 1e6, 1),
                                            "adapter": round(ops.packed_weight_bytes(pg.mlp, pg.attn, post) / 1e6, 1),
                                            "row_major_copies_kept": ops.keep_row_major()},
                       "step_pipelining": "none" if tail_stream is None else "gather+adapter of step i on a second stream, under the tower of step i+1",
                       **extra_cfg},
            "path_mfma": {"algorithmic_tflops": round(path_tflops, 1), "frac_of_peak": round(path_tflops / (PEAK_BF16_TFLOPS * world), 4),
                          "gflop_per_step": round(step_gf, 1),
                          "gflop_per_step_reference_arithmetic": round(step_gf_reference, 1),
                          "gflop_note": "rates are priced on the EXECUTED arithmetic: the gated block's second Linear runs once per global token after the gate "
                                        "mixed the hidden rows (linear map, gates sum to 1/(1+1e-6)); the reference's two complete experts would be the larger figure",
                          # THIS box's bare MFMA stream, measured in this run right after the timed regions (no constant from another box)
                          "power_capped_mfma_stream_tflops": calib["tflops"],
                          "frac_of_power_capped_mfma_stream": round(path_tflops / (calib["tflops"] * world), 4),
                          "mfma_stream_calibration": calib},
            "roofline": {"bound": "mfma",
                         "kernel": f"{roof_kernel['rocprof_name']} ({roof_kernel['label']}, M={roof_kernel['M']} N={roof_kernel['N']} K={roof_kernel['K']})",
                         "achieved": roof_kernel["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(roof_kernel["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src, "traffic_head": traffic_head,
                         # do the committed PMC passes describe the kernel sources this library was built from?
                         "traffic_current": pmc_profile_is_current(),
                         **({"traffic_error": traffic_err} if traffic_err else {}),
                         "traffic_algorithmic": roof_kernel.get("algorithmic_bytes"),
                         "launch_ms": roof_kernel["ms"], "launch_ms_min": roof_kernel["min_ms"], "launch_ms_max": roof_kernel.get("max_ms"),
                         "frac_min": round(roof_kernel["gflop_per_launch"] / roof_kernel.get("max_ms", roof_kernel["ms"]) / PEAK_BF16_TFLOPS, 4),
                         "frac_max": round(roof_kernel["gflop_per_launch"] / roof_kernel["min_ms"] / PEAK_BF16_TFLOPS, 4),
                         "gflop_per_launch": roof_kernel["gflop_per_launch"],
                         "method": "HIP events recorded by the tower driver (slime_vit_forward_ex probe) around this kernel on its launching "
                                   "stream, once per layer over all 23 layers of a tower pass over one half batch (the step's launch shape), other "
                                   "stream idle; achieved / frac = mean over the 23 launches, frac_min / frac_max from the slowest / fastest"
                                   + ("; prefill kernels: HIP events around single launches at the step's shapes" if prefill else ""),
                         "kernels": {v["label"]: {k: v[k] for k in ("rocprof_name", "ms", "min_ms", "max_ms", "tflops") if k in v} for v in per.values()}},
        }
        if prefill:
            res["config"].update(prefill.describe())
        if strong_obj is not None:
            res["strong"] = strong_obj
        if world == 1 and args.config == 2 and not strong:
            # The SAME step in the reference's inference dtype (fp16: llava/model/builder.py:43) -- the dtype whose projector
            # outputs meet north_star's 1e-3 (parity.fp16 below): same region, same K / W, timed right after the bf16 line.
            f16 = torch.float16
            # a second encoder loaded from the fp32 state dicts (NOT the bf16 module cast to fp16 and back: lossy outside fp16's range)
            enc16 = SlimeVisualEncoder(default_slime_config("synthetic:1234"))
            enc16.load_visual_state(tower_sd, adapter_sd)
            enc16.to(dev)
            enc16.get_vision_tower().vision_tower.to(f16)
            tower16, model16 = enc16.get_vision_tower(), enc16.get_model()
            cur.update(pixels=pixels.to(f16), pg=model16.mm_projector.packed(f16), post=model16.sampler.post_qformer.packed(576, f16), dt=f16,
                       tower=tower16)
            e16 = timed_region(phase="fp16")
            res["fp16"] = {"ms_per_step": round(e16 / args.steps * 1e3, 3), "value": round(n_step * args.steps / e16, 1), "unit": "crops/s",
                           "vs_bf16": round(elapsed / e16, 4),
                           "note": "same step, same timed region, fp16 MFMA operands (the reference's inference dtype; meets the 1e-3 target, see parity.fp16)"}
            cur.update(pixels=pixels, pg=pg, post=post, dt=dt, tower=tower)
            del enc16, tower16, model16
        if world == 1 and args.config == 2 and not args.no_cpu_baseline:
            res["cpu_baseline"], px_s, ref_s = cpu_baseline(tower_sd, adapter_sd, CPI)
            res["parity"] = parity_vs_oracle(tower_sd, adapter_sd, px_s, ref_s, dev, NW, NH)
        box.mark("after")
        time.sleep(0.1)
        box.stop()
        res["box"] = {"device": D.device_label(dev), "cus": torch.cuda.get_device_properties(dev).multi_processor_count,
                      **box.summary(("idle", "warmup", "timed", "repeat", "strong", "probe", "calibration", "fp16", "after")),
                      **({"ranks_timed": box_ranks} if box_ranks else {}),
                      "what": "this rank's GPU while the bench ran: idle = context up, nothing running; timed = the headline K steps; repeat = the two repeats; "
                              "probe = per-kernel HIP-event probing; calibration = the bare MFMA stream; after = the last 0.1 s"}
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(res), flush=True)
        os.dup2(2, 1)
    if collective:
        dist.barrier()
        dist.destroy_process_group()


def build_prefill(enc, dev, dt, n_images, visual_rows, text_tokens=64, layers=32, sequences=None, shard=None):
    """Second half of configs 4 / 5: embed table + 32 Llama-3-8B attention sub-layers (random init) and the splice plan.
    Config 4: one sequence per image, [text/2, <image>, text/2].  Config 5 (sequences=1): ONE sequence holding all frames,
    [text chunk, <image>] x n_images + tail.  Returns a callable (visual tokens [n_images, rows, 4096] bf16) -> hidden states."""
    import numpy as np
    from slime_amd import ops
    from slime_amd.model.llava_arch import splice_plan
    from slime_amd.constants import IMAGE_TOKEN_INDEX
    D, HQ, HKV = 4096, 32, 8
    g = torch.Generator().manual_seed(99)
    table = (torch.randn(32000, D, generator=g) * 0.02).to(dt).to(dev)
    images = n_images if sequences is None else sequences                 # batch of the language model
    per_seq = n_images // images
    ids = torch.randint(3, 32000, (images, text_tokens + per_seq), generator=g)
    chunk = text_tokens // (per_seq + 1)
    for j in range(per_seq):
        ids[:, (j + 1) * chunk + j] = IMAGE_TOKEN_INDEX
    src, _, mask, pos = splice_plan(ids.numpy(), None, None, [visual_rows] * n_images)
    S = src.shape[1]
    src_d = torch.from_numpy(src).reshape(-1).to(dev)
    pos_d = torch.from_numpy(pos).to(dev)
    packs = []
    # Random init with the usual scaled residual projection (o_proj x (2 L)^-0.5, GPT-2 / Megatron style): there is no RMSNorm on
    # this path (outside SURVEY section 8), and with plain k^-0.5 weights 32 stacked residual sub-layers blow the hidden state
    # -- and with it the attention logits -- up to 1e5 and more, a regime no trained model is in (SLIME_BENCH_EXPLODING_LOGITS=1
    # restores it; tools/prefill_stack_ab.py times both kernels in both regimes).
    o_scale = 1.0 if os.environ.get("SLIME_BENCH_EXPLODING_LOGITS") == "1" else (2.0 * layers) ** -0.5
    for _ in range(layers):
        w = [torch.randn(n, k, generator=g, dtype=torch.float32) * (k ** -0.5) for n, k in ((HQ * 128, D), (HKV * 128, D), (HKV * 128, D), (D, HQ * 128))]
        wq, wk, wv, wo, hq_r, hkv_r = w[0], w[1], w[2], w[3] * o_scale, HQ, HKV
        if shard is not None:                                   # this rank's kv heads only (slime_amd.dist, --prefill-shard heads)
            from slime_amd import dist as D_
            wq, wk, wv, wo, hq_r, hkv_r = D_.shard_llama_attention_weights(wq, wk, wv, wo, HQ, HKV, shard[0], shard[1])
        packs.append(ops.pack_llama_attention(wq, wk, wv, wo, hq_r, hkv_r, dt, dev, 500000.0))
    M = images * S
    gf_layer = (2.0 * M * (HQ + 2 * HKV) * 128 * D + 2.0 * M * D * HQ * 128 + 4.0 * images * HQ * (S * (S + 1) / 2) * 128) / 1e9

    bufs = [torch.empty((images, S, D), dtype=dt, device=dev) for _ in range(2)]

    def run(tokens):
        # no torch arithmetic in here: the splice writes the 16-bit hidden rows, every layer's residual add rides in its o_proj
        # epilogue (slime_llama_attn_forward_resid: x_next = T(x + self_attn(x)), HF's 16-bit residual stream), ping-pong buffers
        feats = tokens.reshape(-1, tokens.shape[-1])
        x = ops.splice_rows(table, feats, src_d, dt).view(images, S, D)
        for i, p in enumerate(packs):
            if shard is None:
                x = ops.llama_attention_forward_resid(p, x, x, pos_d, None, out=bufs[i & 1])
            else:
                # head-sharded: rank 0's partial carries the residual (its o_proj epilogue adds it), one all-reduce per layer
                from slime_amd import dist as D_
                x = D_.head_sharded_attention(
                    lambda h, r, p=p, i=i: ops.llama_attention_forward_resid(p, h, r, pos_d, None, out=bufs[i & 1]) if r is not None
                    else ops.llama_attention_forward(p, h, pos_d, None, dt), x, x)
        return x

    def kernel_times():
        lib, st = ops._lib.load(), torch.cuda.current_stream().cuda_stream
        p = packs[0]
        HQ, HKV = p.n_heads, p.n_kv_heads                    # this rank's heads (all of them unless --prefill-shard heads)
        N = (HQ + 2 * HKV) * 128
        qkv = (torch.randn(images, S, N, device=dev) * 0.3).to(dt)
        o = torch.empty((images, S, HQ * 128), dtype=dt, device=dev)

        def attn():
            ops._lib.check(lib.slime_prefill_attention(qkv.data_ptr(), S * N, N, qkv.data_ptr() + HQ * 256, S * N, N,
                                                      qkv.data_ptr() + (HQ + HKV) * 256, S * N, N, o.data_ptr(), S * HQ * 128, HQ * 128,
                                                      images, HQ, HKV, 128, S, None, None, ops.dtype_code(dt), st))
        hid = (torch.randn(M, D, device=dev) * 0.3).to(dt)
        out = {}
        for label, name, fn, fl, n_, k_ in (
                ("prefill_attention", "prefill32_kernel<BF16>", attn, 4.0 * images * HQ * (S * (S + 1) / 2) * 128, 0, 0),
                ("llama_qkv_proj", ops.gemm_kernel_name(M, N, D, dt, ops._lib.EPI_BIAS_T, p.tensors["w_qkv_frag"] is not None), lambda: ops.gemm(hid, p.tensors["w_qkv"], None, ops._lib.EPI_BIAS_T, w_frag=p.tensors["w_qkv_frag"]), 2.0 * M * N * D, N, D)):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            out[label] = {"label": label, "rocprof_name": name, "ms": round(ms, 4), "min_ms": round(ms, 4),
                          "tflops": round(fl / ms / 1e9, 1), "gflop_per_launch": round(fl / 1e9, 2), "M": M, "N": n_, "K": k_}
        return out

    run.gflop = gf_layer * layers
    run.kernel_times = kernel_times
    run.describe = lambda: {"prefill_sequences": images, "prefill_seq_len": S, "llama_layers": layers,
                            "prefill_gflop_per_step": round(gf_layer * layers, 1),
                            "prefill_note": "attention sub-layers only (q/k/v projection, RoPE, causal GQA, o_proj with the residual add fused into its epilogue: out = T(x + attn(x)), 16-bit stream as in HF); RMSNorm / MLP / lm_head are outside SURVEY section 8"}
    return run


if __name__ == "__main__":
    main()
