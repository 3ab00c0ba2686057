#!/bin/bash
# Round 5, GPU call: projection[2] storing straight into the token buffer (slime_gemm_args.row_map) vs the fp32 rows + merge passes
# (variant nodirect = SLIME_OPT_ADAPTER_DIRECT=0): adapter / module parity tests, row-map GEMM test, tower + adapter A/B, bench.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_path.py tests/test_gpu_dist.py -m gpu -q 2>&1 | tail -5 ) > gpurun_out/e_tests.log
( timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "row_map or mix or premix or production_shapes" 2>&1 | tail -5 ) >> gpurun_out/e_tests.log
( AB_ADAPTER=1 timeout 900 python tools/lib_variant_ab.py --rounds 2 nodirect product 2>&1 ) > gpurun_out/e_ab.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/e_bench.err ) > gpurun_out/e_bench.json
( SLIME_HIP_LIBRARY=$R/slime_amd/variants/libslime_hip_nodirect.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/e_bench_nodirect.err ) > gpurun_out/e_bench_nodirect.json
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/e_bench2.err ) > gpurun_out/e_bench2.json
cat gpurun_out/e_tests.log gpurun_out/e_ab.log; for f in e_bench e_bench_nodirect e_bench2; do cut -c1-180 gpurun_out/$f.json; done
