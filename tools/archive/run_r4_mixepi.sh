#!/bin/bash
# round 4, last GPU call: the gate mix inside projection[0]'s epilogue (SLIME_EPI_BIAS_GELU_MIX_T) -- full GPU suite, smoke, A/B against the
# library that mixes with a separate pass (slime_gate_premix), the bench line
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1; mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -15 ) > gpurun_out/y_tests.log; cat gpurun_out/y_tests.log | cut -c1-250
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > gpurun_out/y_smoke.log; cat gpurun_out/y_smoke.log
AB_ADAPTER=1 timeout 400 python tools/lib_variant_ab.py --rounds 2 premixkernel product > gpurun_out/y_ab.txt 2>&1; cat gpurun_out/y_ab.txt
( timeout 300 python bench.py --steps 20 --warmup 5 2>gpurun_out/y_bench2.err ) > gpurun_out/y_bench2.json; cut -c1-200 gpurun_out/y_bench2.json
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/y_bench2.json').read().splitlines() if l.startswith('{')][-1]); print(d['parity']['bf16'], d['parity']['fp16'], d['fp16']['value'])
PY
