#!/bin/bash
# round 4: the gate mix on the projector's hidden rows (slime_gate_premix) -- parity tests, A/B against the library without it
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py tests/test_gpu_modules.py tests/test_gpu_dist.py -q -x -k "gate or gated or adapter or fused or encode or chain or path or modules or premix or north or bench_shape or images" > gpurun_out/premix_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/premix_tests.log; tail -5 gpurun_out/premix_tests.log
AB_ADAPTER=1 timeout 600 python tools/lib_variant_ab.py --rounds 3 nopremix product > gpurun_out/premix_ab.txt 2>&1; cat gpurun_out/premix_ab.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/premix_bench.err > gpurun_out/premix_bench.json; cut -c1-400 gpurun_out/premix_bench.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/premix_bench.json').read().strip().splitlines()[-1]); print(d.get('parity'))
PY
