#!/bin/bash
# round 6, GPU call D: the 96-row direct-B dispatch rule (product = SLIME_OPT_DB96=1) against the same library without it (variant
# nodb96), one process per library, interleaved rounds: tower latency at every crop count the rule fires on, the 40-crop tower, the bench
# step; then the bit-equality / parity tests that walk the new tile.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
( AB_SIZES=4,5,6,8,9,10,12,13,14,16,17,20,21,24,34,40 timeout 900 python tools/small_latency_ab.py --rounds 2 nodb96 product 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r6_d_latency_ab.txt
( timeout 600 python tools/lib_variant_ab.py --rounds 2 nodb96 product 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r6_d_tower_ab.txt
for v in nodb96 product nodb96 product; do
  if [ $v = product ]; then unset SLIME_HIP_LIBRARY; else export SLIME_HIP_LIBRARY="$R/slime_amd/variants/libslime_hip_$v.so"; fi
  ( timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['ms_per_step_repeats'], {k: v['ms'] for k, v in d['roofline']['kernels'].items()}, d['box'].get('sclk_mhz_timed'), d['box'].get('power_w_timed'))" ) >> gpurun_out/r6_d_bench_ab.txt
done
unset SLIME_HIP_LIBRARY
( timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -m gpu -q -rf -k "gemm or shard or config1 or production or tile" 2>&1 | grep -v "^$" | tail -15 ) > gpurun_out/r6_d_tests.txt
cat gpurun_out/r6_d_latency_ab.txt gpurun_out/r6_d_tower_ab.txt gpurun_out/r6_d_bench_ab.txt; tail -5 gpurun_out/r6_d_tests.txt
