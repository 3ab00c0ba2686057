#!/bin/bash
# Round 6, same-box A/B of the residual stream's lower part: ABI 6 (16-bit lower part; commit 6e713ed, exported to .old_tree with its own
# product library) against ABI 7 (one signed byte per element): the 40-crop tower in alternating processes, then bench.py of the new tree.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
# .old_tree: mkdir .old_tree && git archive 6e713ed slime_amd include tools/lib_variant_ab.py | tar -x -C .old_tree && make -C .old_tree/slime_amd/csrc product
for r in 1 2 3; do
  ( cd .old_tree && timeout 300 python tools/lib_variant_ab.py --rounds 1 product 2>&1 | grep "^product" | sed 's/^product/lo16 (ABI 6)/' )
  timeout 300 python tools/lib_variant_ab.py --rounds 1 product 2>&1 | grep "^product" | sed 's/^product/lo8  (ABI 7)/'
done | tee gpurun_out/z_lo8_ab.txt
( timeout 400 python bench.py 2>gpurun_out/z_bench_lo8.err ) > gpurun_out/z_bench_lo8.json; cut -c1-330 gpurun_out/z_bench_lo8.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/z_bench_lo8.json'))
print({k:d[k] for k in ('value','ms_per_step','ms_per_step_repeats')}, d['box'].get('sclk_mhz_timed'), d['roofline']['frac'], d['roofline']['launch_ms'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()}, d['fp16']['ms_per_step'], d['parity'])
PY
