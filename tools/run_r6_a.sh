#!/bin/bash
# round 6, GPU call A: the new / tightened tests, what telemetry this box offers, the bench line with box / repeats / calibration
mkdir -p gpurun_out
python - > gpurun_out/r6_a_telemetry_probe.txt 2>&1 <<'PY'
import json, glob, os
try:
    import amdsmi
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    print("amdsmi handles:", len(hs))
    m = amdsmi.amdsmi_get_gpu_metrics_info(hs[0])
    print(json.dumps({k: (v if not isinstance(v, (bytes,)) else str(v)) for k, v in m.items()}, default=str)[:6000])
    print("bdf:", amdsmi.amdsmi_get_gpu_device_bdf(hs[0]))
    try: print("power_info:", amdsmi.amdsmi_get_power_info(hs[0]))
    except Exception as e: print("power_info failed", e)
    try: print("power_cap:", amdsmi.amdsmi_get_power_cap_info(hs[0]))
    except Exception as e: print("power_cap failed", e)
except Exception as e:
    print("amdsmi failed:", type(e).__name__, e)
for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
    print(hw, sorted(os.listdir(hw)))
    for f in ("power1_input", "power1_average", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "power1_cap"):
        try: print("  ", f, open(os.path.join(hw, f)).read().strip())
        except Exception as e: print("  ", f, "unreadable:", type(e).__name__)
import torch
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else "no pci_bus_id")
PY
python -m pytest tests/test_gpu_path.py tests/test_gpu_modules.py tests/test_gpu_dist.py tests/test_checkpoints.py -m gpu -q -rf -s 2>&1 | grep -v "^$" | tail -80 > gpurun_out/r6_a_tests.txt
python bench.py > gpurun_out/r6_a_bench.json 2> gpurun_out/r6_a_bench.err
tail -5 gpurun_out/r6_a_bench.err
python -m pytest tests/test_gpu_bench_contract.py -m gpu -q -rf 2>&1 | tail -30 > gpurun_out/r6_a_contract.txt
tail -3 gpurun_out/r6_a_tests.txt gpurun_out/r6_a_contract.txt
