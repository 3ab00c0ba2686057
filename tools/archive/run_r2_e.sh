#!/bin/bash
# GPU call E: attention K-prefetch A/B + correctness subset, bench, serialized rocprof kernel stats
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -m gpu -x -q -k "attention or tower or production or config3 or golden" 2>&1 | tail -6 ) > gpurun_out/e_tests.log
( timeout 200 python tools/attn_bench.py 2>&1 | head -24 ) > gpurun_out/e_attn.log
( timeout 200 python tools/tower_ab.py attn 2>&1 ) > gpurun_out/e_tower_ab.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/e_bench2.err ) > gpurun_out/e_bench2.json
cd /tmp; export TMPDIR=/tmp
rm -rf "$R/gpurun_out/prof_bench_serial"
( AMD_SERIALIZE_KERNEL=3 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench_serial" -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline 2>"$R/gpurun_out/e_bench_serial.err" ) > "$R/gpurun_out/e_bench_serial.json"
cd "$R"
find gpurun_out/prof_bench_serial -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_bench_kernel_stats_serialized.csv \;
find gpurun_out/prof_bench_serial -name "*kernel_trace.csv" -delete
cat gpurun_out/e_tests.log gpurun_out/e_attn.log gpurun_out/e_tower_ab.log; cut -c1-200 gpurun_out/e_bench2.json; head -8 gpurun_out/r02_bench_kernel_stats_serialized.csv | cut -c1-160; cut -c1-300 gpurun_out/e_bench_serial.json
