#!/bin/bash
# round 4: (1) the persistent-GEMM diagnostics through their own translation unit, (2) what a byte costs (tools/power_probe.py --hbm)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "persistent or direct_b" > gpurun_out/ps_tu.log 2>&1; echo "pytest rc $?" >> gpurun_out/ps_tu.log
timeout 300 python tools/power_probe.py --hbm > gpurun_out/power_hbm.txt 2>&1; echo "rc $?" >> gpurun_out/power_hbm.txt
tail -3 gpurun_out/ps_tu.log; cat gpurun_out/power_hbm.txt
