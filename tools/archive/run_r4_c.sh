#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/r4c2; rm -rf $O; mkdir -p $O
( timeout 600 python tools/tile_rule_ab.py "1024,4096,9" "1024,4096,10" "1024,4096,12" "1024,1024,9" "1024,1024,4" "1024,4096,9;1024,1024,9" "3072,1024,10" "4096,1024,10" 2>&1 | grep -v amdgpu.ids ) > $O/tile_ab.txt; cat $O/tile_ab.txt
( timeout 300 python bench.py --config 3 --steps 10 --warmup 3 2>$O/bench3.err ) > $O/bench3.json; cut -c1-200 $O/bench3.json
( SLIME_BENCH_SINGLE_DEVICE=1 SLIME_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29627 bench.py --gpus 2 --config 2 --scaling strong --steps 2 --warmup 1 --no-cpu-baseline 2>$O/dry2s.err ) > $O/dry2s.json; cut -c1-160 $O/dry2s.json
