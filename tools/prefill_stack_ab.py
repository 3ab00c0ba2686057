#!/usr/bin/env python3
"""bench.py --config 4 / 5's language-model half (splice + 32 random-weight attention sub-layers) with the two prefill attention
kernels, same call, interleaved (diagnostic build).  Random-weight layers stacked 32 deep blow the logits up to 1e5 and beyond, so
that regime (SLIME_BENCH_EXPLODING_LOGITS=1) is timed next to the default init (scaled residual projections), because there the
speculative softmax of prefill32 is thrown away and recomputed on many steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slime_amd import _lib
import bench
dev = torch.device("cuda:0"); dt = torch.bfloat16
with _lib.diag() as lib:
    for seqs, label, explode in ((None, "config 4: 8 sequences x 1216", "0"), (1, "config 5: 1 sequence x 9280", "0"),
                                 (None, "config 4, exploding logits", "1"), (1, "config 5, exploding logits", "1")):
        os.environ["SLIME_BENCH_EXPLODING_LOGITS"] = explode
        run = bench.build_prefill(None, dev, dt, 8, 1152, sequences=seqs)
        tokens = (torch.randn(8, 1152, 4096, device=dev) * 0.02).to(dt)
        for rnd in range(2):
            for var in (0, 1):
                lib.slime_prefill_set_variant(var)
                for _ in range(2): run(tokens)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5): out = run(tokens)
                torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
                print(f"{label}: {'prefill32 ' if var == 0 else 'eight-wave'} {t*1e3:7.2f} ms per pass, finite={bool(torch.isfinite(out.float()).all())}", flush=True)
        del run
    lib.slime_prefill_set_variant(0)
