#!/bin/bash
# Round 4 checkpoint: GPU suite (tightened bounds, new tests), smoke, bench lines: plain / forced collective, default and 4 HW queues
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/r4a; rm -rf $O; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | grep -v "^$" | tail -40 ) > $O/tests.log; tail -25 $O/tests.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/smoke.log; cat $O/smoke.log
B="python bench.py --steps 20 --warmup 5"
run() { ( timeout 400 env "$@" $B 2>$O/$TAG.err ) > $O/$TAG.json; cut -c1-150 $O/$TAG.json | sed "s/^/$TAG /"; python - <<PY
import json
try:
    d=json.load(open("$O/$TAG.json")); print("   fp16:", d.get("fp16"), " parity:", {k:d.get("parity",{}).get(k) for k in ("bf16","fp16")})
except Exception as e: print("   (no json)", e)
PY
}
TAG=plain run X=1
TAG=coll run SLIME_BENCH_FORCE_COLLECTIVE=1
TAG=plain_q4 run GPU_MAX_HW_QUEUES=4
TAG=coll_q4 run SLIME_BENCH_FORCE_COLLECTIVE=1 GPU_MAX_HW_QUEUES=4
