#!/usr/bin/env python3
"""Device slicer timing: uint8 upload + resize/pad/tile/normalise on the GPU vs the host (PIL) slicer + fp32 upload,
for the bench workload (8 images 672x672 -> 8 x (1+4) crops) and a 1344x1344 case (1+6 crops)."""
import os, sys, time
import numpy as np, torch
from PIL import Image
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import mm_utils as M
from slime_amd.image_processor import ClipImageProcessor
dev = torch.device("cuda:0"); proc = ClipImageProcessor()
cfg = SimpleNamespace(image_aspect_ratio="anyres", image_grid_pinpoints="[(336, 672)]")
for (w, h, n) in ((672, 672, 8), (1344, 1344, 8), (1920, 1080, 8)):
    arrs = [np.random.default_rng(i).integers(0, 256, (h, w, 3), dtype=np.uint8) for i in range(n)]
    pil = [Image.fromarray(a, "RGB") for a in arrs]
    host_u8 = [torch.from_numpy(a).pin_memory() for a in arrs]
    dev_u8 = [t.to(dev) for t in host_u8]
    M.process_images_gpu(dev_u8, proc, cfg, dev, torch.bfloat16); torch.cuda.synchronize()
    # (a) device slicer only, pixels already in HBM
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = M.process_images_gpu(dev_u8, proc, cfg, dev, torch.bfloat16)
    e1.record(); torch.cuda.synchronize()
    t_dev = e0.elapsed_time(e1) / 10
    crops = out.shape[0] * out.shape[1]
    # (b) uint8 upload + device slicer (PCIe inclusive)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = M.process_images_gpu(host_u8, proc, cfg, dev, torch.bfloat16)
    torch.cuda.synchronize(); t_up = (time.perf_counter() - t0) / 10 * 1e3
    # (c) host slicer (PIL + numpy, one core) + fp32 upload
    t0 = time.perf_counter()
    ref = M.process_images(pil, proc, cfg); refd = ref.to(dev); torch.cuda.synchronize()
    t_host = (time.perf_counter() - t0) * 1e3
    same = torch.equal(M.process_images_gpu(dev_u8, proc, cfg, dev).cpu(), ref)
    in_b = n * w * h * 3; out_b = crops * 3 * 336 * 336 * 2
    print(f"{n} x {w}x{h} -> {crops} crops: device slicer {t_dev:.3f} ms ({(in_b + out_b) / t_dev / 1e6:.0f} GB/s algorithmic), "
          f"+uint8 H2D {t_up:.3f} ms, host PIL slicer + fp32 H2D {t_host:.1f} ms, bit-identical {same}", flush=True)
