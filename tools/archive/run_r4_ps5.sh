#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/r4p5; rm -rf $O; mkdir -p $O
( timeout 300 python tools/ps_gemm_check.py time 2>&1 | tail -20 ) > $O/time.txt; cat $O/time.txt
