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
    # bound is loose (12 %): the measured 3-6 % on an 8-step timing must not flake
    f = d["fp16"]
    assert f["unit"] == "crops/s" and abs(f["value"] - 40 * 1e3 / f["ms_per_step"]) / f["value"] < 0.01
    assert abs(f["value"] / d["value"] - 1.0) < 0.12, (f["value"], d["value"])      # rounds 5-6: 0.942-0.971 of bf16 over six leases (the fp16 leg runs last, on a hot chip)
    assert max(d["parity"]["fp16"]["rel_l2_global"], d["parity"]["fp16"]["rel_l2_local"]) <= 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "crops/s" and c["cores"] >= 1 and 0 < c["value"] < d["value"]
    assert "traffic_error" not in r, r.get("traffic_error")       # the committed PMC summary knows the dominant kernel of this build
    assert d["ms_per_step_rank_min"] <= d["ms_per_step"] + 1e-3 and d["ms_per_step_rank_max"] >= d["ms_per_step_rank_min"]
    # round 6: the line says what box it ran on and how repeatable it was
    rep = d["ms_per_step_repeats"]
    assert len(rep) == 3 and rep[0] == d["ms_per_step"] and max(rep) / min(rep) < 1.10, rep      # headline = the FIRST K steps; repeats within 10 %
    box = d["box"]
    assert "source" in box and ("MI3" in box["device"] or "gfx950" in box["device"])    # boxes without amdgpu.ids call themselves 'AMD Radeon Graphics'
    from slime_amd import dist as D
    assert D.profile_applies(D.tower_latency_profile(), box["device"], "CLIP-ViT-L/14-336", "bf16")    # ... and the latency profile still knows them
    if box["source"] is not None:                                 # a box with a readable telemetry source: the timed region was sampled
        assert box["timed"]["samples"] >= 2 and 500 < box["sclk_mhz_timed"] < 3000 and 50 < box["power_w_timed"] < 2000, box
        assert box["idle"]["samples"] >= 1
    pm = d["path_mfma"]
    cal = pm["mfma_stream_calibration"]                           # THIS box's bare MFMA stream, measured in this run (no constant)
    assert pm["power_capped_mfma_stream_tflops"] == cal["tflops"] and 800 < cal["tflops"] <= 2600 and cal["launches"] >= 3, cal
    assert abs(pm["frac_of_power_capped_mfma_stream"] - pm["algorithmic_tflops"] / cal["tflops"]) < 2e-3
    assert pm["frac_of_peak"] < pm["frac_of_power_capped_mfma_stream"] < 1.0


def _bench(args, env=None, timeout=900):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)                                            # the driver's plain `python bench.py ...` environment
    e.update(env or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=timeout, cwd=ROOT, env=e)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


ONE_GPU_REHEARSAL = {"SLIME_BENCH_SINGLE_DEVICE": "1", "SLIME_BENCH_BACKEND": "gloo"}


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` exactly as the driver types it (no torch.distributed.run around it, no rank environment):
    the script starts its two ranks itself.  One GPU here, so both ranks sit on cuda:0 and the collective backend is gloo (RCCL
    refuses two ranks on one device) -- the numbers mean nothing, the launch / rendezvous / gather / max-reduce / rank-0 line do."""
    d = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], ONE_GPU_REHEARSAL)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["launched_by"] == "bench.py self-launch" and d["config"]["collective_backend"] == "gloo"
    assert d["config"]["rccl_ranks"] is None                      # gloo rehearsal: no RCCL communicator to report
    assert d["config"]["crops_per_gpu"] == 40 and d["config"]["images_per_step"] == 16
    assert abs(d["value"] - 80 * 1e3 / d["ms_per_step"]) / d["value"] < 0.01          # whole-job rate: 2 x 40 crops per step
    assert d["ms_per_step_rank_min"] <= d["ms_per_step_rank_max"] and abs(d["ms_per_step_rank_max"] - d["ms_per_step"]) < 0.05 * d["ms_per_step"]
    # round 6: ONE driver command returns both curves -- the weak headline says its gather feeds nobody, and the
    # same run times north_star's data flow (the same 40 crops sharded over the ranks, all-gather BEFORE the adapter) as `strong`
    assert d["config"]["gather_consumed"] is False
    s = d["strong"]
    assert s["scaling"] == "strong" and s["gather_consumed"] is True and s["crops_per_rank_padded"] == 20 and s["images_owned_by_rank0"] == 4
    assert s["ms_per_step"] > 0 and abs(s["value"] - 40 * 1e3 / s["ms_per_step"]) / s["value"] < 0.01      # whole job: the 40 crops, once
    assert s["ms_per_step_rank_min"] <= s["ms_per_step_rank_max"]
    assert s["predicted_ms_per_step"] > 0 and 1.0 < s["speedup_vs_n1_predicted"] < 2.0                      # two ranks: below 2x by the model
    assert len(d["ms_per_step_repeats"]) == 3 and len(d["box"].get("ranks_timed", [0, 0])) == 2


def test_bench_gpus_2_strong_leg_deadline_keeps_the_headline():
    """A strong leg that overruns its deadline (here: a deadline no leg can meet) must not cost the weak headline: the two ranks
    leave with status 0 and rank 0's line carries the complete headline fields plus `strong: {"error": "timeout ..."}`
    (bench.DeadlineGuard; the leg has never run on a multi-GPU node, and it runs before the ONE line is printed)."""
    d = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {**ONE_GPU_REHEARSAL, "SLIME_BENCH_STRONG_DEADLINE_S": "0.001"})
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["unit"] == "crops/s" and d["dtype"] == "bf16"
    assert abs(d["value"] - 80 * 1e3 / d["ms_per_step"]) / d["value"] < 0.01 and len(d["ms_per_step_repeats"]) == 3
    assert "timeout" in d["strong"]["error"] and d["roofline"] is None and d["config"]["gather_consumed"] is False


def test_bench_refuses_more_gpus_than_the_node_has():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SLIME_BENCH_SINGLE_DEVICE")}
    import torch
    n = torch.cuda.device_count() + 1
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT, env=e)
    assert res.returncode != 0 and "GPU(s)" in res.stderr and not res.stdout.strip()


@pytest.mark.parametrize("config,extra", [(1, []), (3, []), (4, []), (5, []), (3, ["--gpus", "2"])])
def test_bench_other_configs_print_their_line(config, extra):
    """BASELINE configs 3 / 4 / 5 through bench.py, 2 steps each (and the strong-scaling ones once more as a self-launched 2-rank
    rehearsal on the one GPU): those modes are not what the driver runs, so nothing else would notice them rotting."""
    d = _bench(["--config", str(config), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra, ONE_GPU_REHEARSAL if extra else None)
    world = 2 if extra else 1
    assert d["n_gpus"] == world and d["steps"] == 2 and d["config"]["baseline_config"] == config
    assert d["scaling"] == {1: "weak", 3: "strong", 4: "weak", 5: "strong"}[config]
    crops = {1: 1, 3: 68, 4: 40, 5: 40}[config]
    assert abs(d["value"] - crops * 1e3 / d["ms_per_step"]) / d["value"] < 0.01      # strong scaling / one GPU: the step's crops, once
    assert d["config"]["crops_per_gpu"] == -(-crops // world) if config != 4 else d["config"]["crops_per_gpu"] == 40
    if config == 1:                                               # one crop: the 64 x 64 ring tile is what runs (auto_tile's smallest-grid rule)
        assert "single crop" in d["metric"] and "gemm_kernel<BF16, 64, 64" in d["roofline"]["kernel"] and d["ms_per_step"] < 5.0
    r = d["roofline"]
    assert 0.02 < r["frac"] < 1.0 and r["achieved"] > 0 and "traffic_error" not in r
    if config in (4, 5):
        assert d["config"]["prefill_seq_len"] == {4: 1216, 5: 9280}[config] and d["config"]["llama_layers"] == 32
        assert "prefill32_kernel" in r["kernel"]
        assert d["path_mfma"]["gflop_per_step_reference_arithmetic"] > d["path_mfma"]["gflop_per_step"] > d["config"]["prefill_gflop_per_step"]
