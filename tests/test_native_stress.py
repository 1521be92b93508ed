"""The native host runtime (shared-memory ring, KV block allocator, status board) under
ThreadSanitizer: tests/native/stress.cpp is compiled together with the runtime sources with
`-fsanitize=thread` and run; a data race or a broken invariant fails the test."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT = os.path.join(ROOT, "distributed_llm_inference_amd", "ops", "csrc", "runtime")


def _build(tmp_path, sanitize: bool):
    srcs = sorted(os.path.join(RT, f) for f in os.listdir(RT) if f.endswith(".cpp"))
    exe = str(tmp_path / ("stress_tsan" if sanitize else "stress"))
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "native", "stress.cpp"), *srcs,
           "-o", exe, "-lrt"]
    if sanitize:
        cmd[1:1] = ["-fsanitize=thread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return exe, r


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_native_runtime_stress_plain(tmp_path):
    exe, r = _build(tmp_path, sanitize=False)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "native stress ok" in out.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_native_runtime_stress_under_thread_sanitizer(tmp_path):
    exe, r = _build(tmp_path, sanitize=True)
    if r.returncode != 0 and ("tsan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("ThreadSanitizer runtime not installed with this g++")
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    env.pop("LD_PRELOAD", None)          # TSan must be the first runtime loaded
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    if "FATAL: ThreadSanitizer" in out.stderr and "memory layout" in out.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this sandbox")
    assert out.returncode == 0, out.stdout + out.stderr[-4000:]
