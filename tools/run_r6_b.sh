#!/bin/bash
# round 6, GPU call B (kill-rule experiments, diagnostic build): 96-row direct-B tiles / 192-row ping-pong fc2 stand-alone and inside
# the two-stream tower; attention with conflict-free V reads (the price of the LDS bank conflict); where a 5- / 9-crop pass's time is
# (wall per pass vs the sum of its kernels' own durations under rocprofv3).
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; rm -rf gpurun_out/prof_*
( timeout 900 python tools/r6_tile_ab.py tiles attn --rounds 3 2>&1 | grep -v "amdgpu.ids" ) > gpurun_out/r6_b_tile_ab.txt
cd /tmp; export TMPDIR=/tmp
for n in 5 9; do
  ( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_gap$n" -- python "$R/tools/small_pass_gaps.py" $n 2>&1 | grep "^crops" ) > "$R/gpurun_out/r6_b_gap$n.txt"
  find "$R/gpurun_out/prof_gap$n" -name "*kernel_stats.csv" -exec cp {} "$R/gpurun_out/r6_b_gap${n}_kernel_stats.csv" \;
done
cd "$R"; rm -rf gpurun_out/prof_*
cat gpurun_out/r6_b_tile_ab.txt; cat gpurun_out/r6_b_gap5.txt gpurun_out/r6_b_gap9.txt; head -8 gpurun_out/r6_b_gap5_kernel_stats.csv | cut -c1-160
( timeout 600 python -m pytest "tests/test_gpu_modules.py::test_encode_images_vs_oracle" "tests/test_gpu_modules.py::test_config1_plugin_api_one_crop_full_size" "tests/test_gpu_modules.py::test_encode_images_flags_and_mask" tests/test_checkpoints.py -m gpu -q -rf -s 2>&1 | grep -v "^$" | tail -40 ) > gpurun_out/r6_b_retests.txt
tail -12 gpurun_out/r6_b_retests.txt
