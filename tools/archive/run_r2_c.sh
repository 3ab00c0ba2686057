#!/bin/bash
# GPU call C of round 2: full GPU suite with the LayerNorm fold, bench, in-tower marginals
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -40 ) > gpurun_out/c_tests.log
( timeout 300 python bench.py --steps 20 --warmup 5 2>gpurun_out/c_bench2.err ) > gpurun_out/c_bench2.json
( timeout 200 python tools/marginal_bench.py 2>&1 ) > gpurun_out/c_marginal.log
cat gpurun_out/c_tests.log; cat gpurun_out/c_bench2.json; tail -3 gpurun_out/c_bench2.err; cat gpurun_out/c_marginal.log
