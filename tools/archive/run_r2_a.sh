#!/bin/bash
# GPU call A of round 2: regression tests, MFMA shape probe, attention A/B, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/a_tests.log
( timeout 120 tools/mfma_shape_mix 2>&1 ) > gpurun_out/a_mfma_shape.log
( timeout 200 python tools/attn_bench.py 2>&1 ) > gpurun_out/a_attn.log
( timeout 200 python tools/tower_ab.py attn 2>&1 ) > gpurun_out/a_tower_ab.log
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/a_bench.err ) > gpurun_out/a_bench.json
tail -5 gpurun_out/a_tests.log; cat gpurun_out/a_mfma_shape.log gpurun_out/a_attn.log gpurun_out/a_tower_ab.log gpurun_out/a_bench.json
