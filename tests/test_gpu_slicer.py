"""GPU parity of the device slicer (slime_resize_bicubic_u8 + slime_tile_normalize) through the C ABI:
bit-exact against Pillow (the third-party library the reference calls), against the host slicer and
against the reference-generated pixel goldens."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
PIN = "[(336, 672), (672, 336), (672, 672), (1008, 336), (336, 1008)]"


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


CASES = [((672, 672), (336, 336)), ((1344, 1344), (336, 336)), ((640, 480), (672, 504)), ((1920, 1080), (1008, 567)),
         ((300, 200), (672, 448)), ((4000, 300), (2352, 177)), ((500, 336), (336, 336)), ((336, 500), (336, 226)),
         ((37, 53), (336, 336)), ((1000, 1000), (672, 672)), ((673, 672), (672, 672)), ((336, 336), (336, 336)),
         ((20000, 9), (64, 9)),        # down-scale 312x: source span of a segment exceeds the LDS stage -> direct path
         ((5, 3), (1, 1)), ((1, 1), (336, 336)), ((5000, 5000), (336, 336))]


@pytest.mark.parametrize("src,dst", CASES)
def test_resize_matches_pillow(dev, src, dst):
    from slime_amd import ops
    (w, h), (ow, oh) = src, dst
    img = np.random.default_rng(w * 7919 + h).integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh)))
    got = ops.resize_bicubic_u8(torch.from_numpy(img).to(dev), ow, oh).cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), int(np.abs(got.astype(int) - ref.astype(int)).max())


def test_resize_into_canvas_view(dev):
    """Destination = interior view of a larger canvas (the centred paste): only the view is written."""
    from slime_amd import ops
    img = np.random.default_rng(5).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    canvas = torch.full((672, 672, 3), 7, dtype=torch.uint8, device=dev)
    ops.resize_bicubic_u8(torch.from_numpy(img).to(dev), 672, 504, out=canvas[84:84 + 504, 0:672])
    ref = np.full((672, 672, 3), 7, dtype=np.uint8)
    ref[84:84 + 504] = np.asarray(Image.fromarray(img, "RGB").resize((672, 504)))
    assert np.array_equal(canvas.cpu().numpy(), ref)


def test_resize_rejects_bad_input(dev):
    from slime_amd import ops
    from slime_amd._lib import SlimeHipError
    with pytest.raises(SlimeHipError):
        ops.resize_bicubic_u8(torch.zeros((8, 8, 3), dtype=torch.uint8), 4, 4)          # CPU tensor: no fallback
    with pytest.raises(ValueError):
        ops.resize_bicubic_u8(torch.zeros((8, 8, 4), dtype=torch.uint8, device=dev), 4, 4)


@pytest.mark.parametrize("size", [(672, 672), (336, 336), (640, 480), (500, 900), (1344, 1344), (300, 200),
                                  (1920, 1080), (4000, 300), (100, 100), (823, 823)])
def test_device_slicer_equals_host_slicer(dev, size):
    """process_anyres_image_gpu (everything after the raw uint8 upload in HIP) == process_anyres_image
    (PIL + numpy), bit for bit in fp32; bf16 output = the rounded fp32 output."""
    from slime_amd import mm_utils as M
    from slime_amd.image_processor import ClipImageProcessor
    w, h = size
    img = Image.fromarray(np.random.default_rng(w * 31 + h).integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB")
    proc = ClipImageProcessor()
    ref = M.process_anyres_image(img, proc, PIN)
    got = M.process_anyres_image_gpu(img, proc, PIN, dev)
    assert got.shape == ref.shape and got.dtype == torch.float32
    assert torch.equal(got.cpu(), ref)
    got16 = M.process_anyres_image_gpu(img, proc, PIN, dev, dtype=torch.bfloat16)
    assert torch.equal(got16.cpu(), ref.to(torch.bfloat16))


def test_device_slicer_matches_reference_goldens(dev):
    """Against the reference's own process_images('anyres') pixels (tests/golden/slicer_pixels.npz)."""
    from slime_amd import mm_utils as M
    from slime_amd.image_processor import ClipImageProcessor
    g = np.load(os.path.join(GOLDEN, "slicer_pixels.npz"))
    proc = ClipImageProcessor()
    for i, (w, h) in enumerate(g["img_specs"]):
        arr = np.random.default_rng(100 + i).integers(0, 256, (int(h), int(w), 3), dtype=np.uint8)
        out = M.process_anyres_image_gpu(torch.from_numpy(arr), proc, PIN, dev).cpu()
        assert tuple(out.shape) == tuple(g[f"img{i}_anyres_shape"])
        flat = out.reshape(out.shape[0], -1).double()
        assert np.array_equal(flat[:, g["sample_idx"]].float().numpy(), g[f"img{i}_anyres_samples"]), i
        assert np.allclose(flat.sum(1).numpy(), g[f"img{i}_anyres_sum"], rtol=0, atol=1e-6), i


@pytest.mark.parametrize("size", [(672, 672), (640, 480), (1344, 1344), (300, 200), (1920, 1080)])
def test_batched_device_slicer_equals_host(dev, size):
    """process_images_gpu on a uniform batch (6 launches for the whole batch) == process_images (PIL), bit for bit."""
    from slime_amd import mm_utils as M
    from slime_amd.image_processor import ClipImageProcessor
    w, h = size
    cfg = SimpleNamespace(image_aspect_ratio="anyres", image_grid_pinpoints=PIN)
    imgs = [Image.fromarray(np.random.default_rng(w + 13 * i).integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB") for i in range(3)]
    proc = ClipImageProcessor()
    ref = M.process_images(imgs, proc, cfg)
    got = M.process_images_gpu(imgs, proc, cfg, dev)
    assert got.shape == ref.shape and torch.equal(got.cpu(), ref)
    got16 = M.process_images_gpu([torch.from_numpy(np.array(im)).to(dev) for im in imgs], proc, cfg, dev, torch.bfloat16)
    assert torch.equal(got16.cpu(), ref.to(torch.bfloat16))


def test_mixed_sizes_and_modes(dev):
    from slime_amd import mm_utils as M
    from slime_amd.image_processor import ClipImageProcessor
    cfg = SimpleNamespace(image_aspect_ratio="anyres", image_grid_pinpoints=PIN)
    proc = ClipImageProcessor()
    imgs = [Image.fromarray(np.random.default_rng(i).integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB")
            for i, (w, h) in enumerate([(672, 672), (1344, 1344)])]
    ref = M.process_images(imgs, proc, cfg)
    got = M.process_images_gpu(imgs, proc, cfg, dev)
    assert isinstance(got, list) and len(got) == 2 and all(torch.equal(g.cpu(), r) for g, r in zip(got, ref))
    with pytest.raises(NotImplementedError):
        M.process_images_gpu(imgs, proc, SimpleNamespace(image_aspect_ratio="pad", image_grid_pinpoints=PIN), dev)
