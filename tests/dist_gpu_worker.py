"""Worker of tests/test_gpu_dist.py: one of G ranks that all sit on cuda:0 (a 1-GPU box) and talk over gloo.
Drives the HIP tower through slime_amd.dist.sharded_tower / sharded_encode and checks bit-equality with the 1-rank tensor."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from slime_amd import weights as W, ops
    from slime_amd import dist as D
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    cfg = W.TINY if os.environ.get("SLIME_DIST_GEOM", "tiny") == "tiny" else W.CLIP_L_336
    n = int(os.environ.get("SLIME_DIST_CROPS", "17"))
    tsd = W.make_tower_state_dict(cfg, seed=11)
    pt = ops.pack_tower(tsd, cfg, torch.bfloat16, dev)
    px = W.synthetic_pixels(n, seed=5).to(dev)

    def tower(x):
        return ops.tower_forward(pt, x, out_dtype=torch.bfloat16)

    gathered = D.sharded_tower(tower, px, (cfg.num_patches, cfg.hidden_size), torch.bfloat16)
    chunked = D.sharded_tower(tower, px, (cfg.num_patches, cfg.hidden_size), torch.bfloat16, chunk=2)
    single = tower(px)
    ok = torch.equal(gathered, single) and torch.equal(chunked, single)
    # compressed-local variant (SURVEY section 8e alternative): post_qformer on the shard, gather [.,144,D] for local crops
    acfg = W.ADAPTER_TINY if cfg is W.TINY else W.ADAPTER_8B
    asd = W.make_adapter_state_dict(acfg, seed=12)
    post = ops.pack_resampler(W.sub_state(asd, "sampler.post_qformer."), acfg.mm_hidden_size, acfg.num_heads, cfg.num_patches,
                              torch.bfloat16, dev, acfg.ln_eps)
    per_image = 1 + (n - 1)                                  # one image: crop 0 global, the rest local

    def compress(x):
        return ops.resampler_forward(post, x.float(), want_t=True)[1]

    glob, comp = D.sharded_tower_compressed(tower, compress, px, per_image, (cfg.num_patches, cfg.hidden_size),
                                            acfg.local_queries, torch.bfloat16)
    ok = ok and torch.equal(glob, single[0::per_image]) and torch.equal(comp, compress(single[1:]))
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 3)


if __name__ == "__main__":
    main()
