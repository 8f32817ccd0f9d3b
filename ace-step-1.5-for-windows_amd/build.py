"""Build the C-ABI shared library (csrc/*.hip -> csrc/libace355.so) with hipcc for gfx950.

In-tree, incremental (per-file mtime), no cmake/ninja.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libace355.so")
SOURCES = ["gemm.hip", "attn.hip", "elementwise.hip", "conv.hip", "dit.hip", "vae.hip", "cond.hip", "audio_out.hip", "api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]
# gemm.hip: MFMA results stay in architectural VGPRs.  Left to itself the register allocator moves the 16x16x32 accumulators of the
# one-wave-per-SIMD kernels (3-4 LDS stages: 512 registers available) to AGPRs and then ROTATES them through v_accvgpr_read/write every
# K step (104 moves and 31 s_nops in a 32-MFMA loop: 1435 instead of 857 cycles per K step, profiles/r04/r04_m16_gemm_clk_inpass.txt).
FILE_FLAGS = {"gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the ace355 native library cannot be built")
    return exe


def _newer(a: str, b: str) -> bool:
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "_build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "ace355.h"),
               os.path.abspath(__file__)]   # (the flags live in this file)
    objs, relink = [], force or not os.path.exists(LIB)
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(hd, o) for hd in headers):
            cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(src, []), "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            relink = True
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    if relink:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
