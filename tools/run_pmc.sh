#!/bin/bash
# The three PMC passes over tools/pmc_target.py (separate rocprofv3 runs, no other tracing), reduced by tools/summarize_prof.py.
R=$(pwd); mkdir -p gpurun_out; rm -rf gpurun_out/prof_pmc_sq gpurun_out/prof_pmc_fetch gpurun_out/prof_pmc_write
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$R/gpurun_out/prof_pmc_sq" -- python "$R/tools/pmc_target.py" ) > "$R/gpurun_out/p_pmc_sq.log" 2>&1
( timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/prof_pmc_fetch" -- python "$R/tools/pmc_target.py" ) > "$R/gpurun_out/p_pmc_fetch.log" 2>&1
( timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$R/gpurun_out/prof_pmc_write" -- python "$R/tools/pmc_target.py" ) > "$R/gpurun_out/p_pmc_write.log" 2>&1
cd "$R"
python tools/summarize_prof.py gpurun_out r03 > gpurun_out/p_pmc_summary.log 2>&1
tail -3 gpurun_out/p_pmc_sq.log | cut -c1-200; tail -5 gpurun_out/p_pmc_summary.log | cut -c1-300
find gpurun_out/prof_pmc_sq gpurun_out/prof_pmc_fetch gpurun_out/prof_pmc_write -name "*kernel_trace.csv" -delete
cp profiles/r03_pmc_kernels.json gpurun_out/z_r03_pmc_kernels.json; cp profiles/r03_pmc_summary.json gpurun_out/z_r03_pmc_summary.json; rm -rf gpurun_out/prof_pmc_*
