#!/bin/bash
# Round-6 evidence run, final build: the 3 PMC passes FIRST (bench.py's `traffic` is this build's, stamped with SLIME_GIT_HEAD), the
# full GPU suite, smoke, the bench lines (config 2 = the driver's command, configs 1 / 3 / 4 / 5, forced collective), the self-launched
# 2-rank rehearsals (the default one now carries the `strong` object), rocprofv3 kernel stats (serialised and as-run)
# -> gpurun_out/ (copied to profiles/r06_* afterwards).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/prof_*
cd /tmp; export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$R/gpurun_out/prof_pmc_sq" -- python "$R/tools/pmc_target.py" ) > "$R/gpurun_out/z_pmc_sq.log" 2>&1
( timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/prof_pmc_fetch" -- python "$R/tools/pmc_target.py" ) > "$R/gpurun_out/z_pmc_fetch.log" 2>&1
( timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$R/gpurun_out/prof_pmc_write" -- python "$R/tools/pmc_target.py" ) > "$R/gpurun_out/z_pmc_write.log" 2>&1
cd "$R"
python tools/summarize_prof.py gpurun_out r06 > gpurun_out/z_pmc_summary.log 2>&1
cp profiles/r06_pmc_kernels.json gpurun_out/z_r06_pmc_kernels.json 2>/dev/null
cp profiles/r06_pmc_summary.json gpurun_out/z_r06_pmc_summary.json 2>/dev/null
find gpurun_out -name "*counter_collection.csv" -delete; find gpurun_out/prof_pmc_* -name "*kernel_trace.csv" -delete 2>/dev/null
( timeout 2000 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -12 ) > gpurun_out/z_tests.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/z_smoke.log
( timeout 400 python bench.py 2>gpurun_out/z_bench2_default.err ) > gpurun_out/z_bench2_default.json
( timeout 400 python bench.py --steps 20 --warmup 5 2>gpurun_out/z_bench2.err ) > gpurun_out/z_bench2.json
( SLIME_BENCH_FORCE_COLLECTIVE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/z_bench2c.err ) > gpurun_out/z_bench2c.json
( timeout 300 python bench.py --config 1 --steps 50 --warmup 5 2>gpurun_out/z_bench1.err ) > gpurun_out/z_bench1.json
( timeout 300 python bench.py --config 3 --steps 10 --warmup 3 2>gpurun_out/z_bench3.err ) > gpurun_out/z_bench3.json
( timeout 400 python bench.py --config 4 --steps 5 --warmup 2 2>gpurun_out/z_bench4.err ) > gpurun_out/z_bench4.json
( timeout 400 python bench.py --config 5 --steps 5 --warmup 2 2>gpurun_out/z_bench5.err ) > gpurun_out/z_bench5.json
# 2-rank rehearsals on the one GPU, started the way the driver starts N > 1: plain `python bench.py --gpus 2` (bench.py launches its ranks)
export SLIME_BENCH_SINGLE_DEVICE=1 SLIME_BENCH_BACKEND=gloo
( timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/z_dry2.err ) > gpurun_out/z_dry2.json
( timeout 400 python bench.py --gpus 2 --config 2 --scaling strong --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/z_dry2s.err ) > gpurun_out/z_dry2s.json
( timeout 400 python bench.py --gpus 2 --config 3 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/z_dry3.err ) > gpurun_out/z_dry3.json
( timeout 400 python bench.py --gpus 2 --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/z_dry5.err ) > gpurun_out/z_dry5.json
( timeout 400 python bench.py --gpus 2 --config 5 --prefill-shard heads --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/z_dry5h.err ) > gpurun_out/z_dry5h.json
unset SLIME_BENCH_SINGLE_DEVICE SLIME_BENCH_BACKEND
cd /tmp
( AMD_SERIALIZE_KERNEL=3 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench_serial" -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline 2>"$R/gpurun_out/z_bench_serial.err" ) > "$R/gpurun_out/z_bench_serial.json"
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench" -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline 2>"$R/gpurun_out/z_bench_prof.err" ) > "$R/gpurun_out/z_bench_prof.json"
cd "$R"
find gpurun_out/prof_bench_serial -name "*kernel_stats.csv" -exec cp {} gpurun_out/z_kernel_stats_serialized.csv \;
find gpurun_out/prof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/z_kernel_stats_concurrent.csv \;
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
rm -rf gpurun_out/prof_*
cat gpurun_out/z_tests.log gpurun_out/z_smoke.log; for f in 2_default 2 2c 1 3 4 5; do cut -c1-260 gpurun_out/z_bench$f.json; done; for f in 2 2s 3 5 5h; do cut -c1-160 gpurun_out/z_dry$f.json; tail -1 gpurun_out/z_dry$f.err | cut -c1-200; done; head -9 gpurun_out/z_kernel_stats_serialized.csv | cut -c1-150
