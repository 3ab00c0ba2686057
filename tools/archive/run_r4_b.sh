#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/r4b; rm -rf $O; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_modules.py -m gpu -q -x 2>&1 | tail -5 ) > $O/tests.log; cat $O/tests.log
( timeout 300 python tools/graph_small.py 2>&1 | grep -v amdgpu.ids ) > $O/graph_small.txt; cat $O/graph_small.txt
