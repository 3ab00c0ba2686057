import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import ops, _lib
dev = torch.device("cuda:0"); lib = _lib.load_diag(); dt = torch.bfloat16
for (M, N, K) in ((256, 256, 64), (256, 256, 128)):
    a = torch.ones(M, K, device=dev).to(dt); w = torch.ones(N, K, device=dev).to(dt); b = torch.zeros(N, device=dev)
    lib.slime_gemm_force_tile(11)
    out = ops.gemm(a, w, b, 3); torch.cuda.synchronize()
    vals, counts = torch.unique(out, return_counts=True)
    print(M, N, K, "unique values:", list(zip(vals.tolist()[:12], counts.tolist()[:12])))
    # which 16x16 tiles are wrong
    wrong = (out != K).view(M // 16, 16, N // 16, 16).any(3).any(1)
    print("wrong 16x16 tiles (rows = m-tile, cols = n-tile):")
    for r in range(M // 16): print("".join("X" if wrong[r, c] else "." for c in range(N // 16)))
    # a with row index pattern to see which k chunk is lost
    a2 = torch.zeros(M, K, device=dev); a2[:, :32] = 1.0
    out2 = ops.gemm(a2.to(dt), w, b, 3); torch.cuda.synchronize()
    print("first-half-k only:", torch.unique(out2, return_counts=True))
lib.slime_gemm_force_tile(0)
