#!/bin/bash
# round 6, GPU call G: fc2 on 192-row ping-pong tiles at the 20-crop half batch (variant pp192 = -DSLIME_OPT_PP192=1) against the product
# (256 rows), judged on the BENCH STEP (tower + adapter, three streams), alternating processes; plus the two-stream tower alone.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/r6_g_bench_ab.txt
for v in product pp192 product pp192 product pp192; do
  if [ $v = product ]; then unset SLIME_HIP_LIBRARY; else export SLIME_HIP_LIBRARY="$R/slime_amd/variants/libslime_hip_$v.so"; fi
  ( timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['ms_per_step_repeats'], d['roofline']['frac'], {k: v['ms'] for k, v in d['roofline']['kernels'].items()}, d['box'].get('sclk_mhz_timed'), d['box'].get('power_w_timed'), d['fp16']['ms_per_step'])" ) >> gpurun_out/r6_g_bench_ab.txt
done
unset SLIME_HIP_LIBRARY
( timeout 600 python tools/lib_variant_ab.py --rounds 3 product pp192 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r6_g_tower_ab.txt
cat gpurun_out/r6_g_bench_ab.txt gpurun_out/r6_g_tower_ab.txt
