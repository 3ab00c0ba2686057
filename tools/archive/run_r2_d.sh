#!/bin/bash
# GPU call D of round 2: full GPU suite, bench, rocprofv3 kernel stats + 3 PMC passes -> profiles/r02_*
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/prof_bench gpurun_out/prof_pmc_sq gpurun_out/prof_pmc_fetch gpurun_out/prof_pmc_write
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -30 ) > gpurun_out/d_tests.log
( timeout 300 python bench.py --steps 20 --warmup 5 2>gpurun_out/d_bench2.err ) > gpurun_out/d_bench2.json
cd /tmp; export TMPDIR=/tmp
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench" -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline 2>"$R/gpurun_out/d_bench_prof.err" ) > "$R/gpurun_out/d_bench_prof.json"
( timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$R/gpurun_out/prof_pmc_sq" -- python "$R/tools/pmc_target.py" ) > "$R/gpurun_out/d_pmc_sq.log" 2>&1
( timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/prof_pmc_fetch" -- python "$R/tools/pmc_target.py" ) > "$R/gpurun_out/d_pmc_fetch.log" 2>&1
( timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$R/gpurun_out/prof_pmc_write" -- python "$R/tools/pmc_target.py" ) > "$R/gpurun_out/d_pmc_write.log" 2>&1
cd "$R"
python tools/summarize_prof.py gpurun_out r02 > gpurun_out/d_pmc_summary.log 2>&1
cp profiles/r02_pmc_kernels.json gpurun_out/r02_pmc_kernels.json 2>/dev/null
find gpurun_out/prof_bench -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_bench_kernel_stats.csv \;
# keep the merged output small: drop the raw traces
find gpurun_out/prof_bench gpurun_out/prof_pmc_sq gpurun_out/prof_pmc_fetch gpurun_out/prof_pmc_write -name "*kernel_trace.csv" -delete
cat gpurun_out/d_tests.log; cat gpurun_out/d_bench2.json | cut -c1-400; head -20 gpurun_out/r02_bench_kernel_stats.csv | cut -c1-200; tail -5 gpurun_out/d_pmc_sq.log | cut -c1-200
