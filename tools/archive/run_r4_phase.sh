#!/bin/bash
# round 4: the shipped defaults (attention partner re-deal + XCD-owned rows in the ping-pong GEMM) against 'off', and the stream phase offset probe
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python tools/lib_variant_ab.py --rounds 3 off product > gpurun_out/defaults_ab.txt 2>&1; cat gpurun_out/defaults_ab.txt
timeout 300 python tools/phase_offset_ab.py > gpurun_out/phase_offset.txt 2>&1; cat gpurun_out/phase_offset.txt
