import os, sys, hashlib, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from slime_amd import ops, weights as W
dev, dt = torch.device("cuda:0"), torch.bfloat16
sha = lambda t: hashlib.sha1(t.float().cpu().numpy().tobytes()).hexdigest()[:10]
asd = W.make_adapter_state_dict(W.ADAPTER_8B, seed=4321)
pg = ops.pack_gated(W.sub_state(asd, "mm_projector."), W.ADAPTER_8B, dt, dev)
post = ops.pack_resampler(W.sub_state(asd, "sampler.post_qformer."), 1024, 8, 576, dt, dev, W.ADAPTER_8B.ln_eps)
g = torch.Generator().manual_seed(3)
feats = (torch.randn(40, 576, 1024, generator=g) * 0.7).to(dt).to(dev)
loc = feats.view(8, 5, 576, 1024)[:, 1:].reshape(32, 576, 1024)
glob = feats.view(8, 5, 576, 1024)[:, 0].contiguous()
ln = ops.layernorm(loc.reshape(-1, 1024).float().contiguous(), post.tensors["ln_kv_w"], post.tensors["ln_kv_b"], 1e-6, dt)[1]
comp = ops.resampler_forward(post, loc.float(), want_t=True)[1]
fused = ops.adapter_forward(pg, post, feats, 8, 4, 2, 2, True, -1, dt)
pre = ops.adapter_forward_precompressed(pg, glob, comp, 8, 4, 2, 2, True, -1, dt)
torch.cuda.synchronize()
print(os.environ.get("SLIME_HIP_LIBRARY", "product")[-22:], "ln", sha(ln), "comp", sha(comp), "fused", sha(fused), "precompressed", sha(pre),
      "fused==pre", bool(torch.equal(fused, pre)), "local part equal", bool(torch.equal(fused[:, 576:], pre[:, 576:])))
