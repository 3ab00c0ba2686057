#!/bin/bash
# round 4: q/k/v + context share the MLP intermediate's region of the tower workspace (165 MB per 20-crop stream instead of 260): A/B + parity
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python tools/lib_variant_ab.py --rounds 3 noalias product > gpurun_out/alias_ab.txt 2>&1; cat gpurun_out/alias_ab.txt
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_gpu_modules.py -q -x > gpurun_out/alias_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/alias_tests.log; tail -3 gpurun_out/alias_tests.log
