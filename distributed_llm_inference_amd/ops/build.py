"""Build the HIP kernel library in-tree for gfx950 (MI355X).

`python -m distributed_llm_inference_amd.ops.build` compiles every `csrc/*.hip` with hipcc
(`--offload-arch=gfx950`; cross-compiles without a GPU), compiles `bindings.cpp` against the
PyTorch headers and links `ops/_C.so`, which registers the `torch.ops.dlia.*` custom ops.
Only sources newer than their object are recompiled. The native runtime pieces
(`csrc/runtime/*.cpp`: shared-memory control ring, KV block allocator) go to `ops/_rt.so`.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
ARCH = "gfx950"
HIP_FLAGS = ["-O3", f"--offload-arch={ARCH}", "-std=c++17", "-fPIC", "-fno-math-errno", "-fno-gpu-rdc",
             "-Wno-unused-result"]
# kernels with hand-counted vmcnt waits: {source: kernel-name regex}; their ISA is linted after compiling
LINTED = {"gemm_skinny.hip": "gemm_skinny_kernel", "gemm_a8w8.hip": "gemm_a8w8_kernel"}


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: " + " ".join(cmd) + "\n" + r.stdout)
    return r.stdout


def _stale(src, obj, extra_deps=()):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in (src, *extra_deps))


def _lint(hipcc, src, regex):
    """Compile `src` to ISA text and make sure no in-flight load destination is read early."""
    from .isa_lint import lint_asm
    asm = os.path.join(OBJ, os.path.basename(src) + ".s")
    _run([hipcc, *HIP_FLAGS, "-S", "--cuda-device-only", src, "-o", asm])
    with open(asm) as f:
        bad = lint_asm(f.read(), regex)
    if bad:
        raise RuntimeError(f"{src}: {len(bad)} instruction(s) read a register whose load is still in flight, "
                           f"first: {bad[0][0]}: {bad[0][1]}")


def build(verbose: bool = True, force: bool = False) -> str:
    import torch
    from torch.utils import cpp_extension as ce

    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.path.join(os.environ.get("ROCM_HOME", "/opt/rocm"), "bin", "hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hip_srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    jobs = []
    objs = []
    lint = []
    for f in hip_srcs:
        src, obj = os.path.join(CSRC, f), os.path.join(OBJ, f + ".o")
        objs.append(obj)
        if force or _stale(src, obj, headers):
            jobs.append([hipcc, *HIP_FLAGS, "-c", src, "-o", obj])
            if f in LINTED:
                lint.append((src, LINTED[f]))
    tinc = []
    for p in ce.include_paths("cuda"):
        tinc += ["-isystem", p]
    bsrc, bobj = os.path.join(CSRC, "bindings.cpp"), os.path.join(OBJ, "bindings.o")
    objs.append(bobj)
    if force or _stale(bsrc, bobj, headers):
        jobs.append([hipcc, "-O2", "-std=c++17", "-fPIC", "-x", "c++", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
                     "-DHIPBLAS_V2", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
                     "-Wno-deprecated-declarations", *tinc, "-c", bsrc, "-o", bobj])
    if jobs:
        if verbose:
            print(f"[dlia.build] compiling {len(jobs)} translation unit(s) for {ARCH}", flush=True)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(_run, jobs))
        for src, regex in lint:
            _lint(hipcc, src, regex)
    out = os.path.join(HERE, "_C.so")
    if jobs or not os.path.exists(out):
        tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
        _run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", *objs, "-o", out,
              f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip", "-lc10_hip", "-lamdhip64",
              f"-Wl,-rpath,{tlib}"])
        if verbose:
            print(f"[dlia.build] linked {out}", flush=True)
    build_runtime(verbose=verbose, force=force)
    build_comm(verbose=verbose, force=force)
    return out


def build_comm(verbose: bool = True, force: bool = False) -> str:
    """Native communication + stage-loop library (`csrc/comm/*.cpp` -> `ops/_comm.so`): host C++ on
    the HIP runtime API; RCCL itself is resolved at run time from the library PyTorch loads. Linked
    against the HIP runtime PyTorch ships so that a process never holds two of them."""
    import torch
    cdir = os.path.join(CSRC, "comm")
    out = os.path.join(HERE, "_comm.so")
    if not os.path.isdir(cdir):
        return out
    srcs = sorted(os.path.join(cdir, f) for f in os.listdir(cdir) if f.endswith(".cpp"))
    deps = srcs + [os.path.join(cdir, f) for f in os.listdir(cdir) if f.endswith(".h")]
    if srcs and (force or not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)):
        rocm = os.environ.get("ROCM_HOME", "/opt/rocm")
        tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-D__HIP_PLATFORM_AMD__=1",
              f"-I{rocm}/include", *srcs, "-o", out, f"-L{tlib}", "-lamdhip64", "-ldl", "-lrt",
              f"-Wl,-rpath,{tlib}"])
        if verbose:
            print(f"[dlia.build] linked {out}", flush=True)
    return out


def build_runtime(verbose: bool = True, force: bool = False) -> str:
    """Native host runtime (no GPU code, no torch): plain C ABI, loaded with ctypes."""
    rdir = os.path.join(CSRC, "runtime")
    out = os.path.join(HERE, "_rt.so")
    if not os.path.isdir(rdir):
        return out
    srcs = sorted(os.path.join(rdir, f) for f in os.listdir(rdir) if f.endswith(".cpp"))
    deps = srcs + [os.path.join(rdir, f) for f in os.listdir(rdir) if f.endswith(".h")]
    if srcs and (force or not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", *srcs, "-o", out, "-lrt"])
        if verbose:
            print(f"[dlia.build] linked {out}", flush=True)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
