#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "gate" > gpurun_out/premix_tests_a.log 2>&1; echo "pytest rc $?" >> gpurun_out/premix_tests_a.log; tail -25 gpurun_out/premix_tests_a.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_gpu_modules.py tests/test_gpu_dist.py tests/test_checkpoints.py -q -m gpu > gpurun_out/premix_tests_b.log 2>&1; echo "pytest rc $?" >> gpurun_out/premix_tests_b.log; tail -12 gpurun_out/premix_tests_b.log | cut -c1-250
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/premix_smoke.log; cat gpurun_out/premix_smoke.log
