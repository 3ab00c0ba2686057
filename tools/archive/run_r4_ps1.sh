#!/bin/bash
# Round 4: first run of the persistent direct-B GEMM (bit-equality, stand-alone timing, tower A/B) + the collective path with more
# hardware queues (is the 15 % loss of the forced-collective line two tower streams aliasing onto one HW queue?)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/r4p1; rm -rf $O; mkdir -p $O
( timeout 600 python tools/ps_gemm_check.py check 2>&1 | tail -40 ) > $O/check.txt; tail -3 $O/check.txt
( timeout 300 python tools/ps_gemm_check.py time 2>&1 | tail -20 ) > $O/time.txt; cat $O/time.txt
( timeout 400 python tools/ps_gemm_check.py tower 2>&1 | tail -20 ) > $O/tower.txt; cat $O/tower.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
run() { ( timeout 300 env "$@" $B 2>$O/$TAG.err ) > $O/$TAG.json; cut -c1-120 $O/$TAG.json | sed "s/^/$TAG /"; }
TAG=coll_q8 run SLIME_BENCH_FORCE_COLLECTIVE=1 GPU_MAX_HW_QUEUES=8
TAG=plain_q8 run GPU_MAX_HW_QUEUES=8
TAG=coll_q4 run SLIME_BENCH_FORCE_COLLECTIVE=1
