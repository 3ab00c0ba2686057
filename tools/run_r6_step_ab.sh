#!/bin/bash
# Round 6: same-box A/B of the BENCH STEP (tower + adapter, the driver's command): the tree before the one-byte lower part (commit 6e713ed,
# ABI 6, exported to .old_tree with its own product library: mkdir .old_tree && git archive 6e713ed | tar -x -C .old_tree && make -C
# .old_tree/slime_amd/csrc product) against the final tree (ABI 7), alternating processes, 3 rounds.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; : > gpurun_out/z_step_ab.txt
for r in 1 2 3; do
  for t in old new; do
    if [ $t = old ]; then d=.old_tree; else d=.; fi
    ( cd $d && timeout 300 python bench.py --no-cpu-baseline 2>/dev/null ) > gpurun_out/z_step_ab_$t$r.json
    python - "$t" gpurun_out/z_step_ab_$t$r.json <<'PY' | tee -a gpurun_out/z_step_ab.txt
import json, sys
d = json.load(open(sys.argv[2])); b = d["box"]
print(f"{'ABI 6 (16-bit lower part)' if sys.argv[1] == 'old' else 'ABI 7 (one-byte lower part)':28s}: {d['value']:7.1f} crops/s  {d['ms_per_step']:7.3f} ms/step  repeats {d['ms_per_step_repeats'][1]:7.3f} {d['ms_per_step_repeats'][2]:7.3f}  sclk {b.get('sclk_mhz_timed')} MHz  {b.get('power_w_timed')} W  fc2 probe {d['roofline']['launch_ms']:.4f} ms  fp16 step {d['fp16']['ms_per_step']:.3f} ms")
PY
  done
done
