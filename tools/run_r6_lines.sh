#!/bin/bash
# Round 6: the bench lines of tools/run_r6_final.sh once more (bench.py's traffic_algorithmic now prices the residual epilogues at
# 6 B per element; the library is the one the PMC passes / GPU suite / rocprofv3 stats of that call ran on -- same kernel-source digest),
# and the tower latency sweep behind slime_amd/data/tower_latency_*.json on the ABI-7 tower.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 400 python bench.py 2>gpurun_out/z_bench2_default.err ) > gpurun_out/z_bench2_default.json
( timeout 400 python bench.py --steps 20 --warmup 5 2>gpurun_out/z_bench2.err ) > gpurun_out/z_bench2.json
( SLIME_BENCH_FORCE_COLLECTIVE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/z_bench2c.err ) > gpurun_out/z_bench2c.json
( timeout 300 python bench.py --config 1 --steps 50 --warmup 5 2>gpurun_out/z_bench1.err ) > gpurun_out/z_bench1.json
( timeout 300 python bench.py --config 3 --steps 10 --warmup 3 2>gpurun_out/z_bench3.err ) > gpurun_out/z_bench3.json
( timeout 400 python bench.py --config 4 --steps 5 --warmup 2 2>gpurun_out/z_bench4.err ) > gpurun_out/z_bench4.json
( timeout 400 python bench.py --config 5 --steps 5 --warmup 2 2>gpurun_out/z_bench5.err ) > gpurun_out/z_bench5.json
export SLIME_BENCH_SINGLE_DEVICE=1 SLIME_BENCH_BACKEND=gloo
( timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/z_dry2.err ) > gpurun_out/z_dry2.json
unset SLIME_BENCH_SINGLE_DEVICE SLIME_BENCH_BACKEND
( timeout 400 python tools/stream_split_sweep.py 2>gpurun_out/z_stream_split_sweep.txt ) > gpurun_out/z_stream_split_sweep.json
( AB_SIZES=1,2,3,5,9 timeout 300 python tools/small_latency_ab.py product 2>&1 | grep -v amdgpu.ids | tail -12 ) > gpurun_out/z_small_latency.txt
for f in 2_default 2 2c 1 3 4 5; do cut -c1-200 gpurun_out/z_bench$f.json; done; cut -c1-160 gpurun_out/z_dry2.json; grep -v amdgpu gpurun_out/z_stream_split_sweep.txt | head -45; cat gpurun_out/z_small_latency.txt
