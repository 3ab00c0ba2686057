#!/bin/bash
# GPU call B of round 2: full GPU suite (f-2, f-4, router, production shapes, cfg3, outliers, 2-rank), bench configs 2/3/4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -40 ) > gpurun_out/b_tests.log
( timeout 200 python tools/attn_bench.py 2>&1 | head -12 ) > gpurun_out/b_attn.log
( timeout 300 python bench.py --steps 10 --warmup 3 2>gpurun_out/b_bench2.err ) > gpurun_out/b_bench2.json
( timeout 300 python bench.py --config 3 --steps 10 --warmup 3 2>gpurun_out/b_bench3.err ) > gpurun_out/b_bench3.json
( timeout 300 python bench.py --config 3 --gather compressed --steps 10 --warmup 3 2>gpurun_out/b_bench3c.err ) > gpurun_out/b_bench3c.json
( timeout 400 python bench.py --config 4 --steps 5 --warmup 2 2>gpurun_out/b_bench4.err ) > gpurun_out/b_bench4.json
cat gpurun_out/b_tests.log gpurun_out/b_attn.log; for f in 2 3 3c 4; do echo "== bench $f"; cat gpurun_out/b_bench$f.json; tail -3 gpurun_out/b_bench$f.err; done
