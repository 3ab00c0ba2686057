#!/bin/bash
# Round-3 evidence after the persistent prefill32: full GPU suite, smoke, bench lines of configs 2 / 4 / 5, the 1-rank RCCL self-test of
# the N > 1 code path, rocprofv3 kernel stats of config 4 and 5 -> gpurun_out/ (copied to profiles/ afterwards)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/prof_*
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -6 ) > gpurun_out/y_tests.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/y_smoke.log
( timeout 300 python bench.py --steps 20 --warmup 5 2>gpurun_out/y_bench2.err ) > gpurun_out/y_bench2.json
( timeout 400 python bench.py --config 4 --steps 5 --warmup 2 2>gpurun_out/y_bench4.err ) > gpurun_out/y_bench4.json
( timeout 400 python bench.py --config 5 --steps 5 --warmup 2 2>gpurun_out/y_bench5.err ) > gpurun_out/y_bench5.json
( SLIME_BENCH_FORCE_COLLECTIVE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/y_bench_rccl1.err ) > gpurun_out/y_bench_rccl1.json
cd /tmp; export TMPDIR=/tmp
for CFG in 4 5; do
( AMD_SERIALIZE_KERNEL=3 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_cfg$CFG" -- python "$R/bench.py" --config $CFG --steps 4 --warmup 1 --no-cpu-baseline 2>"$R/gpurun_out/y_bench${CFG}_prof.err" ) > "$R/gpurun_out/y_bench${CFG}_prof.json"
find "$R/gpurun_out/prof_cfg$CFG" -name "*kernel_stats.csv" -exec cp {} "$R/gpurun_out/y_kernel_stats_config$CFG.csv" \;
done
cd "$R"
find gpurun_out -name "*kernel_trace.csv" -delete
rm -rf gpurun_out/prof_*
cat gpurun_out/y_tests.log gpurun_out/y_smoke.log; for f in 2 4 5 _rccl1; do cut -c1-200 gpurun_out/y_bench$f.json; done; tail -2 gpurun_out/y_bench_rccl1.err | cut -c1-200
for CFG in 4 5; do head -6 gpurun_out/y_kernel_stats_config$CFG.csv | cut -c1-140; done
