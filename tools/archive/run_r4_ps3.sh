#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/r4p4; rm -rf $O; mkdir -p $O
( timeout 600 python tools/ps_gemm_check.py check 2>&1 | cut -c1-400 ) > $O/check.txt; grep -v "^bfloat16\|^float16" $O/check.txt | tail -30; grep -c "equal=True" $O/check.txt
( timeout 300 python tools/ps_gemm_check.py time 2>&1 | tail -20 ) > $O/time.txt; cat $O/time.txt
( timeout 400 python tools/ps_gemm_check.py tower 2>&1 | tail -20 ) > $O/tower.txt; cat $O/tower.txt
