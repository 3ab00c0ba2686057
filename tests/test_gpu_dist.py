"""the HIP tower driven through slime_amd.dist with world_size 2 and 3 -- every rank on cuda:0 of the
1-GPU test box, gloo between the processes (RCCL refuses two ranks on one device) -- must give the 1-rank tensor bit for bit,
for the one-shot gather, the chunked gather and the compressed-local variant.  (tests/test_dist_gloo.py covers the same
functions on CPU with the oracle as the stand-in tower.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world,crops,geom", [(2, 17, "tiny"), (3, 10, "tiny"), (2, 10, "vitl")])
def test_hip_tower_through_sharded_tower_two_ranks_one_gpu(world, crops, geom):
    """``vitl`` (VERDICT r5 item 1 iii): the shapes an 8-GPU config-2 / config-5 run executes -- 2 ranks x 5 crops of the real
    CLIP-ViT-L/14-336 tower (sub-round GEMM grids: 128x128 ring tiles, attn64r's two-workgroup form) against the 10-crop pass."""
    env = dict(os.environ, SLIME_DIST_CROPS=str(crops), SLIME_DIST_GEOM=geom, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + world + (10 if geom == "vitl" else 0)), os.path.join(HERE, "dist_gpu_worker.py")]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:]
