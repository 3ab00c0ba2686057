#!/bin/bash
# round 6, GPU call H: last check of the final tree -- bench contract tests (incl. the 2-rank rehearsal with the strong object and --config 1), the default bench line twice
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_dist.py tests/test_checkpoints.py -m gpu -q -rf 2>&1 | grep -v "^$" | tail -12 ) > gpurun_out/r6_h_tests.txt
( timeout 300 python bench.py 2>gpurun_out/r6_h_bench1.err ) > gpurun_out/r6_h_bench1.json
( timeout 300 python bench.py --gpus 1 --steps 20 --warmup 3 2>gpurun_out/r6_h_bench2.err ) > gpurun_out/r6_h_bench2.json
tail -4 gpurun_out/r6_h_tests.txt; cut -c1-300 gpurun_out/r6_h_bench1.json; cut -c1-300 gpurun_out/r6_h_bench2.json
