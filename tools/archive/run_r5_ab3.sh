#!/bin/bash
# Round 5, GPU call 3: full GPU suite (64 x 64 ring tile, fixed fragment-image test), the full-clock budget table,
# small-batch tower latency with / without the 64 x 64 tile, the tower latency curve (dist.py profile), patch-embed timing.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_contract.py 2>&1 | tail -30 ) > gpurun_out/c_tests.log
( timeout 300 python tools/full_clock_budget.py 2>&1 | tail -14 ) > gpurun_out/c_budget.log
( timeout 600 python tools/small_latency_ab.py --rounds 2 no64 product 2>&1 ) > gpurun_out/c_small.log
( timeout 400 python tools/stream_split_sweep.py 2>gpurun_out/c_sweep.txt ) > gpurun_out/c_sweep.json
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/c_bench.err ) > gpurun_out/c_bench.json
tail -8 gpurun_out/c_tests.log; cat gpurun_out/c_budget.log gpurun_out/c_small.log; cut -c1-300 gpurun_out/c_sweep.json; cut -c1-260 gpurun_out/c_bench.json
