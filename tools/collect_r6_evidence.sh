#!/bin/bash
# gpurun_out/z_* of tools/run_r6_final.sh -> profiles/r06_* (the names profiles/README.md indexes).  Run in the build container after the call.
cd "$(dirname "$0")/.." || exit 1
G=gpurun_out; P=profiles
cp $G/z_r06_pmc_kernels.json $P/r06_pmc_kernels.json; cp $G/z_r06_pmc_summary.json $P/r06_pmc_summary.json
{ cat $G/z_tests.log; cat $G/z_smoke.log; } > $P/r06_gpu_suite.txt
cp $G/z_bench2_default.json $P/r06_bench_line.json
cp $G/z_bench2.json $P/r06_bench_line_steps20_warmup5.json
cp $G/z_bench2c.json $P/r06_bench_line_rccl_1rank_selftest.json
for c in 1 3 4 5; do cp $G/z_bench$c.json $P/r06_bench_line_config$c.json; done
cp $G/z_dry2.json $P/r06_dryrun_2ranks_1gpu_config2.json; cp $G/z_dry2s.json $P/r06_dryrun_2ranks_1gpu_config2_strong.json
cp $G/z_dry3.json $P/r06_dryrun_2ranks_1gpu_config3.json; cp $G/z_dry5.json $P/r06_dryrun_2ranks_1gpu_config5.json
cp $G/z_dry5h.json $P/r06_dryrun_2ranks_1gpu_config5_heads.json
cp $G/z_kernel_stats_serialized.csv $P/r06_bench_kernel_stats.csv; cp $G/z_bench_serial.json $P/r06_bench_line_under_rocprof_serialized.json
cp $G/z_kernel_stats_concurrent.csv $P/r06_bench_kernel_stats_concurrent.csv; cp $G/z_bench_prof.json $P/r06_bench_line_under_rocprof.json
python - <<'PY'
import json
for f in ("r06_bench_line.json", "r06_bench_line_steps20_warmup5.json"):
    d = json.load(open("profiles/" + f))
    print(f, d["value"], d["ms_per_step"], d["ms_per_step_repeats"], d["box"].get("sclk_mhz_timed"), d["box"].get("power_w_timed"), d["roofline"]["frac"], d["roofline"]["launch_ms"],
          d["roofline"]["traffic"], d["roofline"]["traffic_current"], d["path_mfma"]["frac_of_peak"], d["path_mfma"]["power_capped_mfma_stream_tflops"], d["path_mfma"]["frac_of_power_capped_mfma_stream"], d.get("fp16", {}).get("ms_per_step"))
    print({k: v["ms"] for k, v in d["roofline"]["kernels"].items()}, d.get("parity", {}).get("bf16"), d.get("parity", {}).get("fp16"), d.get("cpu_baseline", {}).get("value"))
PY
