"""bench.py keeps the driver's contract: one JSON line on stdout with the required keys, a roofline object measured
with HIP events, and (at N=1) the CPU-oracle baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "3"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 3 and d["unit"] == "crops/s" and d["dtype"] == "bf16"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["value"] - 40 * 1e3 / d["ms_per_step"]) / d["value"] < 0.01          # 40 crops per step
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 1.0
    assert r["traffic"] is not None and r["traffic"] > 0 and r["traffic_source"].startswith("profiles/")
    # the fp16 instantiation (the reference's inference dtype, the one that meets the 1e-3 accuracy target) is timed in the same
    # run over the same region and runs at the bf16 line's speed.  Measured: fp16 is 2.3 % SLOWER than bf16 -- the
    # step is power-capped and 11-bit significands switch more multiplier bits than 8-bit ones (profiles/r04_power_cap.txt) -- so the
    # bound is 5 %: 3 % would sit 0.7 % from the measured value on an 8-step timing
    f = d["fp16"]
    assert f["unit"] == "crops/s" and abs(f["value"] - 40 * 1e3 / f["ms_per_step"]) / f["value"] < 0.01
    assert abs(f["value"] / d["value"] - 1.0) < 0.05, (f["value"], d["value"])
    assert max(d["parity"]["fp16"]["rel_l2_global"], d["parity"]["fp16"]["rel_l2_local"]) <= 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "crops/s" and c["cores"] >= 1 and 0 < c["value"] < d["value"]
