"""Build invariants read back from the code objects inside the shared libraries (no GPU needed).

The hand-scheduled kernels keep their MFMA accumulators in AGPRs through inline asm, count their own vmcnt / lgkmcnt and pad MFMA
latencies by hand: a register spill (scratch traffic is VMEM: it would break the counted waits) must fail the build, not a run."""
import os
import re
import shutil
import subprocess

import pytest

from slime_amd import _lib

LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_notes(so_path, tmp_path):
    objcopy, bundler, readelf = (os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))
    if not all(os.path.exists(t) for t in (objcopy, bundler, readelf)) or not os.path.exists(so_path):
        pytest.skip("ROCm LLVM tools or the library are not available")
    fat = str(tmp_path / "fat.bin")
    subprocess.check_call([objcopy, f"--dump-section=.hip_fatbin={fat}", so_path, str(tmp_path / "discard.o")])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]          # one bundle per translation unit
    kernels = {}
    for i, st in enumerate(starts):
        part, co = str(tmp_path / f"bundle{i}.bin"), str(tmp_path / f"dev{i}.co")
        open(part, "wb").write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        subprocess.check_call([bundler, "--type=o", "--unbundle", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        text = subprocess.check_output([readelf, "--notes", co], text=True)
        for block in text.split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block).group(1)
            kernels[name] = {k: int(v) for k, v in
                             re.findall(r"\.(private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count):\s+(\d+)", block)}
    return kernels


@pytest.mark.parametrize("which", ["product", "diag"])
def test_hand_scheduled_kernels_do_not_spill(which, tmp_path):
    kernels = _kernel_notes(_lib.LIB_PATH if which == "product" else _lib.DIAG_LIB_PATH, tmp_path)
    assert kernels, "no kernels found in the code object"
    watched = [n for n in kernels if re.search(r"attn32_kernel|prefill32_kernel|attn64r_kernel|gemm_w4_kernel|gemm_db_kernel|prefill_attn_kernel", n)]
    if which == "diag":
        assert any("attn32_kernel" in n for n in watched), "the diagnostic library lost the attn32 alternative"
    assert any("gemm_w4_kernel" in n for n in watched) and any("prefill32_kernel" in n for n in watched)
    assert any("gemm_db_kernel" in n for n in watched)
    # two workgroups per CU is the point of the direct-B kernel: at most 256 registers per lane (VGPR + AGPR)
    assert all(kernels[n]["vgpr_count"] <= 256 for n in watched if "gemm_db_kernel" in n)
    for n in watched:
        if "attn32_kernel" in n or "prefill32_kernel" in n:
            assert kernels[n]["private_segment_fixed_size"] == 0 and kernels[n]["vgpr_spill_count"] == 0, (n, kernels[n])
    # round 4's persistent GEMMs count every VMEM request by hand AND use all 256 accumulator registers: any spill is a bug
    ps = [n for n in kernels if re.search(r"gemm_ps(32)?_kernel", n)]
    if which == "diag":
        assert ps, "the diagnostic library lost the persistent GEMM alternatives"
    for n in ps:
        assert kernels[n]["private_segment_fixed_size"] == 0 and kernels[n]["vgpr_spill_count"] == 0, (n, kernels[n])
    spilled = {n: kernels[n]["vgpr_spill_count"] for n in watched if kernels[n]["vgpr_spill_count"] > 40}
    assert not spilled, f"register spills grew: {spilled}"
