#!/bin/bash
# Evidence run for the attn32 alternative (diagnostic build).  Usage: gpurun -- 'bash tools/run_r2_attn32.sh'
mkdir -p gpurun_out
# the probes are built on the build machine (hipcc cross-compiles) and travel with the snapshot:
#   hipcc --offload-arch=gfx950 -O3 -w -o tools/attn32_slot_probe tools/attn32_slot_probe.hip
#   python tools/attn32_iter_probe.py tools/attn32_iter/*.txt
python tools/attn32_check.py > gpurun_out/r02_attn32_check.txt 2>&1
{ tools/attn32_slot_probe; python tools/attn32_iter_probe.py --run; } > gpurun_out/r02_attn32_probes.txt 2>&1
python tools/attn32_stamps.py 5 2>&1 | grep -v "first-wave\|launch span\|amdgpu.ids" > gpurun_out/r02_attn32_stamps.txt
python tools/attn_cold_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_attention_cold_inputs.txt
python tools/tower_ab.py attn32 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_tower_attn32_two_streams.txt
NSTREAMS=1 python tools/tower_ab.py attn32 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_tower_attn32_one_stream.txt
python tools/tower_streams.py 40 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_tower_stream_count.txt
tail -n 30 gpurun_out/r02_attn32_check.txt gpurun_out/r02_attn32_probes.txt
