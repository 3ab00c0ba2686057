"""bench.py's own rank launcher (CPU part): `python bench.py --gpus N` with no rank environment re-runs the script under
torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1 -- checked here on the command / environment it builds;
tests/test_gpu_bench_contract.py runs it for real on the GPU box."""
import importlib.util
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("slime_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_self_launch_builds_one_rank_per_gpu(monkeypatch):
    bench = _load_bench()
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("SLIME_BENCH_SINGLE_DEVICE", "1")
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert bench.self_launch(4) == 7                           # the children's exit status is handed through
    cmd, env = seen["cmd"], seen["env"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]      # the ranks get the caller's arguments unchanged
    assert env["GPU_MAX_HW_QUEUES"] == "8" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["SLIME_BENCH_SELF_LAUNCHED"] == "1"


def test_self_launch_refuses_without_enough_gpus(monkeypatch):
    bench = _load_bench()
    monkeypatch.delenv("SLIME_BENCH_SINGLE_DEVICE", raising=False)
    import torch
    with pytest.raises(SystemExit) as e:
        bench.self_launch(torch.cuda.device_count() + 1)
    assert "GPU(s)" in str(e.value)


def test_main_self_launches_only_without_a_rank_environment(monkeypatch):
    bench = _load_bench()
    calls = []
    monkeypatch.setattr(bench, "self_launch", lambda n: calls.append(n) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert calls == [2] and e.value.code == 0
    # under torch.distributed.run (WORLD_SIZE set) the script is a rank: a mismatch is an error, not another launch
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert calls == [2] and "WORLD_SIZE=4" in str(e.value)


def test_pmc_traffic_reports_a_missing_kernel_without_aborting():
    bench = _load_bench()
    t, src, head, err = bench.pmc_traffic("no_such_kernel<BF16>")
    assert t is None and src.startswith("profiles/") and err and "no_such_kernel" in err
    t, src, head, err = bench.pmc_traffic("no_such_kernel<BF16>", profiled_shape=False)
    assert t is None and err is None and src.startswith("not profiled")


def test_box_sampler_reduces_samples_per_phase(monkeypatch):
    """bench.py's BoxSampler (round 6): samples carry the phase that was current, summary() reduces per phase and lifts the timed
    region's means to the flat keys the bench line promises; a box without any telemetry source yields source None, no exception."""
    import time
    bench = _load_bench()
    vals = iter(range(10_000))

    def fake_open(self, idx):
        self.source = "fake"
        return lambda: {"sclk_mhz": 2000.0 + next(vals) % 3, "mclk_mhz": 1900.0, "power_w": 1000.0, "temp_c": None}
    monkeypatch.setattr(bench.BoxSampler, "_open", fake_open)
    b = bench.BoxSampler(0, period=0.002)
    b.mark("idle"); time.sleep(0.03)
    b.mark("timed"); time.sleep(0.05)
    b.mark("after"); time.sleep(0.01)
    b.stop()
    s = b.summary(("idle", "timed", "after", "never"))
    assert s["source"] == "fake" and s["timed"]["samples"] >= 5 and s["never"]["samples"] == 0
    assert 2000.0 <= s["sclk_mhz_timed"] <= 2002.0 and s["power_w_timed"] == 1000.0 and "temp_c_timed" not in s
    assert s["timed"]["sclk_mhz"]["min"] >= 2000.0 and s["timed"]["sclk_mhz"]["max"] <= 2002.0
    monkeypatch.setattr(bench.BoxSampler, "_open", lambda self, idx: None)
    none = bench.BoxSampler(0)
    none.mark("timed"); none.stop()
    assert none.summary(("timed",))["source"] is None
    # _num: amdsmi hands back ints, "N/A" strings and per-XCD lists (0xFFFF = not populated)
    assert bench.BoxSampler._num(1950) == 1950.0 and bench.BoxSampler._num("N/A") is None
    assert bench.BoxSampler._num([2000, 1900, 65535, "N/A"]) == 1950.0 and bench.BoxSampler._num([]) is None


def test_box_sampler_real_sources_never_raise():
    """On this GPU-less container every source is missing or refuses (amdsmi: driver not loaded): the constructor falls through."""
    bench = _load_bench()
    b = bench.BoxSampler(0)
    b.mark("timed"); b.stop()
    s = b.summary(("timed",))
    assert "source" in s


def test_deadline_guard_prints_the_headline_and_leaves_when_the_second_leg_hangs():
    """bench.py's DeadlineGuard (round 6): the `strong` leg of the N > 1 line runs before rank 0 prints (one JSON line), so a hang
    inside it -- a stuck collective on hardware the leg has never run on -- would take the finished weak headline with it.  A leg
    that overruns its deadline: the callback prints from the timer thread and the process leaves with status 0 while the main
    thread is still stuck; a leg that returns, or raises, in time: nothing fires."""
    code = r'''
import importlib.util, json, sys, time
spec = importlib.util.spec_from_file_location("b", sys.argv[1]); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
mode = sys.argv[2]
def cb(seconds): print(json.dumps({"value": 1.0, "strong": {"error": f"timeout after {seconds} s"}}), flush=True)
if mode == "hang":
    with b.DeadlineGuard(0.3, cb):
        time.sleep(30)                      # the stuck collective
    print("not reached")
elif mode == "ok":
    with b.DeadlineGuard(5.0, cb) as g:
        time.sleep(0.05)
    time.sleep(0.2); print(json.dumps({"fired": g.fired}))
else:
    try:
        with b.DeadlineGuard(5.0, cb) as g:
            raise RuntimeError("leg failed")
    except RuntimeError:
        time.sleep(0.2); print(json.dumps({"fired": g.fired, "raised": True}))
'''
    import json
    import time
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "bench.py"), "hang"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and time.perf_counter() - t0 < 25, (r.returncode, r.stderr[-400:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and "timeout" in json.loads(lines[0])["strong"]["error"]
    r = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "bench.py"), "ok"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip()) == {"fired": False}
    r = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "bench.py"), "raise"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip()) == {"fired": False, "raised": True}
