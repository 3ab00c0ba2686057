#!/bin/bash
# round 4: fabric-read traffic options (streaming stores, attention partner re-deal, XCD-owned rows) -- parity tests on the default build,
# A/B timings per option, FETCH_SIZE per kernel for 'off' and the default build
cd "$GRAFT_REPO_ROOT"; R=$(pwd); mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -q -x > gpurun_out/traffic_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/traffic_tests.log
tail -3 gpurun_out/traffic_tests.log
timeout 900 python tools/lib_variant_ab.py --rounds 3 off nt pair pp db product > gpurun_out/traffic_ab.txt 2>&1
cat gpurun_out/traffic_ab.txt
cd /tmp && export TMPDIR=/tmp
for v in off product; do
  rm -rf "$R/gpurun_out/prof_fetch_$v"
  ( VARIANT=$v timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/prof_fetch_$v" -- python "$R/tools/fetch_target.py" ) > "$R/gpurun_out/fetch_$v.log" 2>&1
done
cd "$R"
python tools/fetch_summary.py off=gpurun_out/prof_fetch_off all_on=gpurun_out/prof_fetch_product > gpurun_out/traffic_fetch.txt 2>&1
cat gpurun_out/traffic_fetch.txt
rm -rf gpurun_out/prof_fetch_*
