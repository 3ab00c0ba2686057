#!/bin/bash
# round 6, GPU call F: the SHIPPED form of the 96-row rule (fc2 only, 8-10 crops) against the library without it; then the tests that walk it
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
( AB_SIZES=8,9,12,16,17,18,19,20,21,24,40 timeout 900 python tools/small_latency_ab.py --rounds 3 nodb96 product 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r6_f_latency_ab.txt
rm -f gpurun_out/r6_f_bench_ab.txt
for v in nodb96 product nodb96 product; do
  if [ $v = product ]; then unset SLIME_HIP_LIBRARY; else export SLIME_HIP_LIBRARY="$R/slime_amd/variants/libslime_hip_$v.so"; fi
  ( timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['ms_per_step_repeats'], {k: v['ms'] for k, v in d['roofline']['kernels'].items()}, d['box'].get('sclk_mhz_timed'), d['box'].get('power_w_timed'))" ) >> gpurun_out/r6_f_bench_ab.txt
done
unset SLIME_HIP_LIBRARY
( timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py tests/test_gpu_modules.py tests/test_checkpoints.py -m gpu -q -rf -k "gemm or shard or config1 or production or tile or encode_images or verify" 2>&1 | grep -v "^$" | tail -15 ) > gpurun_out/r6_f_tests.txt
cat gpurun_out/r6_f_latency_ab.txt gpurun_out/r6_f_bench_ab.txt; tail -5 gpurun_out/r6_f_tests.txt
