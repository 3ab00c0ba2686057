#!/bin/bash
# Round 4, final check at HEAD: GPU suite, smoke, bench line (refreshes profiles/r04_bench_line.json with the power-context fields)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/r4d; rm -rf $O; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -6 ) > $O/tests.log; cat $O/tests.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log; cat $O/smoke.log
( timeout 300 python bench.py --steps 20 --warmup 5 2>$O/bench2.err ) > $O/bench2.json; cut -c1-260 $O/bench2.json
