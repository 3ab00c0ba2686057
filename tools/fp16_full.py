import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import weights as W
from slime_amd.model.multimodal_encoder.clip_encoder import HipCLIPVisionModel
dev = torch.device("cuda:0")
for dt in (torch.float16, torch.bfloat16):
    vm = HipCLIPVisionModel(W.CLIP_L_336); vm.load_state_dict(W.make_tower_state_dict(W.CLIP_L_336, seed=1234)); vm.to(dev).to(dt)
    px = W.synthetic_pixels(40, seed=9).to(dev).to(dt)
    out = vm.encode(px); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): out = vm.encode(px)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
    print(dt, "finite", bool(torch.isfinite(out.float()).all()), "absmax", float(out.float().abs().max()), f"{t*1e3:.2f} ms {40/t:.0f} crops/s")
    if dt == torch.float16: ref16 = out.float()
    else: print("fp16 vs bf16 rel-L2", float((out.float() - ref16).norm() / ref16.norm()))
