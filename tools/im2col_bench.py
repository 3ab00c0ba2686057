import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import _lib
lib = _lib.load_diag(); dev = torch.device("cuda:0")
n = 20
px = torch.randn(n, 3, 336, 336, device=dev).to(torch.bfloat16)
out = torch.empty((n * 576, 640), dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): lib.slime_im2col(px.data_ptr(), _lib.BF16, out.data_ptr(), n, 336, 14, 640, _lib.BF16, st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): lib.slime_im2col(px.data_ptr(), _lib.BF16, out.data_ptr(), n, 336, 14, 640, _lib.BF16, st)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 50
print(f"im2col 20 crops: {t*1e3:.1f} us, {(px.numel()*2 + out.numel()*2)/t/1e6:.0f} GB/s")
