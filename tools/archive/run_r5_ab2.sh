#!/bin/bash
# Round 5, GPU call 2: full GPU suite (fragment-layout staging in the LDS kernels, prefetching patch-embed loop), A/B of the
# balanced XCD row ownership in the direct-B kernel and of fragment-only weights vs the row-major copies, patch-embed timing, bench.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/prof_*
( timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_contract.py 2>&1 | tail -40 ) > gpurun_out/b_tests.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/b_smoke.log
( timeout 120 python tools/patch_embed_bench.py 2>&1 | tail -4 ) > gpurun_out/b_pe.log
( timeout 900 python tools/lib_variant_ab.py --rounds 2 base product product@SLIME_KEEP_ROW_MAJOR=1 xcdb 2>&1 ) > gpurun_out/b_ab.log
( timeout 300 python bench.py --steps 20 --warmup 5 2>gpurun_out/b_bench.err ) > gpurun_out/b_bench.json
cat gpurun_out/b_tests.log | tail -30; cat gpurun_out/b_smoke.log gpurun_out/b_pe.log gpurun_out/b_ab.log; cut -c1-330 gpurun_out/b_bench.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/b_bench.json"))
print({k:(v["ms"],v["tflops"]) for k,v in d["roofline"]["kernels"].items()}, d.get("fp16",{}).get("ms_per_step"), d.get("parity"))
PY
