"""Builds a variant of the kernel library for same-box A/B runs: one source recompiled with extra
defines, linked with the other objects of the regular build into `ops/_C_<tag>.so`. Select it at run
time with DLIA_OPS_LIB=<path> (see ops/__init__.py).

    python bench/ab_build.py fw gemm_pingpong.hip -DPP_FINE_WAIT=1
"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPS = os.path.join(ROOT, "distributed_llm_inference_amd", "ops")
sys.path.insert(0, ROOT)
from distributed_llm_inference_amd.ops import build as b  # noqa: E402


def main():
    tag, src, defines = sys.argv[1], sys.argv[2], sys.argv[3:]
    b.build(verbose=False)
    hipcc = "/opt/rocm/bin/hipcc"
    obj = os.path.join(b.OBJ, f"{src}.{tag}.o")
    subprocess.check_call([hipcc, *b.HIP_FLAGS, *defines, "-c", os.path.join(b.CSRC, src), "-o", obj])
    objs = [os.path.join(b.OBJ, f + ".o") for f in sorted(os.listdir(b.CSRC)) if f.endswith(".hip") and f != src]
    objs += [obj, os.path.join(b.OBJ, "bindings.o")]
    out = os.path.join(OPS, f"_C_{tag}.so")
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    subprocess.check_call([hipcc, "-shared", "-fPIC", f"--offload-arch={b.ARCH}", "-fno-gpu-rdc", *objs, "-o", out,
                           f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip", "-lc10_hip", "-lamdhip64",
                           f"-Wl,-rpath,{tlib}"])
    print(out)


if __name__ == "__main__":
    main()
