cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -8 ) > gpurun_out/z_tests.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > gpurun_out/z_smoke.log
( timeout 400 python bench.py 2>gpurun_out/z_bench2_default.err ) > gpurun_out/z_bench2_default.json
cat gpurun_out/z_tests.log gpurun_out/z_smoke.log; cut -c1-400 gpurun_out/z_bench2_default.json
