#!/bin/bash
# Round 5, GPU call 1: full GPU suite on the new build (fused patch-embed front end, split residual stream, short attention tail,
# fragment-image-only weights), then the same-box A/B of each change as a product-library variant (tools/build_variants.sh), the
# bench line and a serialised rocprofv3 kernel-stats pass.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/prof_*
( timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bench_contract.py 2>&1 | tail -25 ) > gpurun_out/a_tests.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/a_smoke.log
( timeout 900 python tools/lib_variant_ab.py --rounds 2 base split tail product product@SLIME_KEEP_ROW_MAJOR=1 2>&1 ) > gpurun_out/a_ab.log
( timeout 300 python bench.py --steps 20 --warmup 5 2>gpurun_out/a_bench.err ) > gpurun_out/a_bench.json
( SLIME_HIP_LIBRARY=$R/slime_amd/variants/libslime_hip_base.so timeout 300 python bench.py --steps 20 --warmup 5 2>gpurun_out/a_bench_base.err ) > gpurun_out/a_bench_base.json
cd /tmp; export TMPDIR=/tmp
( AMD_SERIALIZE_KERNEL=3 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench_serial" -- python "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline 2>"$R/gpurun_out/a_bench_serial.err" ) > "$R/gpurun_out/a_bench_serial.json"
cd "$R"
find gpurun_out/prof_bench_serial -name "*kernel_stats.csv" -exec cp {} gpurun_out/a_kernel_stats_serialized.csv \;
rm -rf gpurun_out/prof_*
cat gpurun_out/a_tests.log | tail -12; cat gpurun_out/a_smoke.log; cat gpurun_out/a_ab.log; cut -c1-400 gpurun_out/a_bench.json; cut -c1-300 gpurun_out/a_bench_base.json; head -14 gpurun_out/a_kernel_stats_serialized.csv | cut -c1-160
