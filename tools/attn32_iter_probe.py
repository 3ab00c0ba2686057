#!/usr/bin/env python3
"""Generates and runs a micro-kernel that loops over ONE main-loop iteration of attn32 copied from the compiler's output
(physical registers and all; file of instructions as argument), one wave per SIMD, and prints cycles per iteration; variants of
the text (nops stripped, ...) show what the iteration's cycles are made of.  Usage: attn32_iter_probe.py iter.txt [more.txt ...]"""
import os, subprocess, sys, tempfile
SRC = r'''
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 512
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc) {
    extern __shared__ char smem[];
    if (smem[threadIdx.x] == 77) out[0] = 1.f;
    // defined values in every register the text reads (finite, small)
    asm volatile(INIT ::: CLOBBERS);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) asm volatile(BODY ::: CLOBBERS);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 4096); hipMalloc(&cyc, 256 * 4 * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int i = 0; i < 2; ++i) { hipLaunchKernelGGL(k, dim3(256), dim3(256), 150 * 1024, 0, out, cyc); hipDeviceSynchronize(); }
    unsigned long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += (double)h[i];
    printf("%s: %.1f cycles per iteration\n", NAME, s / 1024 / REP);
    return 0;
}
'''
def build(path, name, outdir):
    text = [l.strip() for l in open(path) if l.strip() and not l.strip().startswith("#")]
    import re
    regs = set()
    for l in text:
        for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]', l):
            for i in range(int(m.group(2)), int(m.group(3)) + 1): regs.add(f"{m.group(1)}{i}")
        for m in re.finditer(r'\b([va])(\d+)\b', l): regs.add(f"{m.group(1)}{m.group(2)}")
        for m in re.finditer(r'\bs\[(\d+):(\d+)\]', l):
            for i in range(int(m.group(1)), int(m.group(2)) + 1): regs.add(f"s{i}")
        for m in re.finditer(r'(?<![\w\[])s(\d+)\b', l): regs.add(f"s{m.group(1)}")
    vregs = sorted(r for r in regs if r[0] == 'v'); aregs = sorted(r for r in regs if r[0] == 'a'); sregs = sorted(r for r in regs if r[0] == 's')
    init = "".join(f"v_mov_b32 {r}, 0x3c003c00\\n\\t" for r in vregs) + "".join(f"v_accvgpr_write_b32 {r}, 0\\n\\t" for r in aregs)
    body = "".join(l.replace('"', '') + "\\n\\t" for l in text)
    clob = ", ".join(f'"{r}"' for r in vregs + aregs + sregs) + ', "memory", "vcc", "scc"'
    src = SRC.replace("INIT", '"' + init + '"').replace("BODY", '"' + body + '"').replace("CLOBBERS", clob).replace("NAME", '"' + name + '"')
    f = os.path.join(outdir, name + ".hip"); open(f, "w").write(src)
    exe = os.path.join(outdir, name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-w", "-o", exe, f])
    return exe
if __name__ == "__main__":
    outdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_iter_probe"); os.makedirs(outdir, exist_ok=True)
    if sys.argv[1] == "--run":
        for e in sorted(os.listdir(outdir)):
            p = os.path.join(outdir, e)
            if os.access(p, os.X_OK) and not e.endswith(".hip"): subprocess.call([p])
    else:
        for path in sys.argv[1:]: build(path, os.path.splitext(os.path.basename(path))[0], outdir)
