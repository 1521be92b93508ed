"""bench.py's contract with the benchmark driver, exercised on CPU (`--cpu`: torch backend, gloo):
one JSON line from rank 0 with the agreed keys, whole-job throughput consistent with the timed
steps, and the same command line under `torch.distributed.run` for N > 1."""
import importlib.util
import json
import os
import socket
import subprocess
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
KEYS = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
        "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _check(line: str, n: int, steps: int, warmup: int):
    r = json.loads(line)
    for k, t in KEYS.items():
        assert k in r, k
        assert isinstance(r[k], t) or (t is float and isinstance(r[k], int)), (k, r[k])
    assert "vs_baseline" in r
    assert r["n_gpus"] == n and r["steps"] == steps and r["warmup"] == warmup
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["unit"] == "tokens/s"
    cfg = r["config"]
    assert cfg["parallelism"] == f"pp{n}" and cfg["seq_len"] == 128
    assert cfg["global_batch"] == cfg["micro_batches"] * cfg["micro_batch_size"]
    # value is the whole job: every in-flight sequence emits one token per step
    want = cfg["global_batch"] / (r["ms_per_step"] * 1e-3)
    assert abs(r["value"] - want) <= 0.01 * want, (r["value"], want)
    return r


def _json_lines(out: str):
    return [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]


def test_single_rank_prints_one_json_line():
    p = subprocess.run([sys.executable, BENCH, "--cpu", "--model", "tiny-llama-test", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    r = _check(lines[0], 1, 3, 1)
    assert r["config"]["micro_batches"] == 2           # 2 x gpus


def test_gpus_n_without_a_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the command starts two rank
    processes itself and prints the one line with n_gpus = 2 — stdout carries that line and nothing
    else — with what a reader needs to check a scaling run: every rank, its transport, the hops timed."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "2"
    p = subprocess.run([sys.executable, BENCH, "--cpu", "--gpus", "2", "--model", "tiny-llama-test", "--steps", "3",
                        "--warmup", "1", "--max-seconds", "240"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    assert len(p.stdout.strip().splitlines()) == 1, p.stdout
    r = _check(p.stdout.strip(), 2, 3, 1)
    assert r["config"]["launcher"] == "self" and r["config"]["transport"] == "gloo"
    assert [x["rank"] for x in r["ranks"]] == [0, 1] and all(x["transport"] == "gloo" and x["ms_per_step"] > 0 for x in r["ranks"])
    assert {h["hop"] for h in r["hops"]} == {"0->1", "1->0 (tokens)"} and all(h["us_min"] > 0 for h in r["hops"])
    # the low-latency regions ride in the same line, timed the same way
    for key, rows in (("latency_point", 8), ("single_request", 1)):
        pt = r[key]
        assert pt["batch"] == rows and pt["micro_batches"] == 1 and pt["ms_per_token"] > 0
        assert abs(pt["tok_s"] - rows / (pt["ms_per_token"] * 1e-3)) <= 0.01 * pt["tok_s"]


def test_a_launcher_with_another_world_size_is_an_error_not_a_smaller_run():
    """--gpus 4 under a launcher that started 2 ranks (or 1) must fail, never quietly run PP=2."""
    for ws in ("2", "1"):
        env = dict(os.environ, WORLD_SIZE=ws, RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        p = subprocess.run([sys.executable, BENCH, "--cpu", "--gpus", "4", "--model", "tiny-llama-test"],
                           capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
        assert p.returncode == 2 and "WORLD_SIZE" in p.stderr and not _json_lines(p.stdout), (p.returncode, p.stderr[-500:])


def test_a_failing_rank_fails_the_self_launched_job():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, BENCH, "--cpu", "--gpus", "2", "--model", "no-such-model", "--max-seconds", "60"],
                       capture_output=True, text=True, timeout=200, cwd=ROOT, env=env)
    assert p.returncode != 0 and not _json_lines(p.stdout)


def test_preflight_opens_times_and_closes_every_hop():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, BENCH, "--cpu", "--gpus", "3", "--preflight"], capture_output=True, text=True,
                       timeout=120, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(p.stdout.strip())
    assert r["n_gpus"] == 3 and r["transport"] == "gloo" and r["seconds"] < 20
    assert {h["hop"] for h in r["hops"]} == {"0->1", "1->2", "2->0 (tokens)"}
    assert "hop 1->2" in p.stderr


def test_two_ranks_under_torch_distributed_run():
    """The driver's N > 1 launch line: one rank per stage, rank 0 prints the line, the others none."""
    env = dict(os.environ, OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--cpu", "--gpus", "2",
                        "--model", "tiny-llama-test", "--steps", "3", "--warmup", "1", "--max-seconds", "240"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    r = _check(lines[0], 2, 3, 1)
    assert r["config"]["micro_batches"] == 6 and len(r["config"]["layer_ranges"]) == 2      # 2 per stage + 2 spares


def test_eight_ranks_under_torch_distributed_run():
    """The driver's N = 8 launch line, rehearsed on CPU: eight stage processes, seven forward hops
    and the token feedback hop over gloo, 16 micro-batches in flight, rank 0 prints the one line."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--cpu", "--gpus", "8",
                        "--model", "tiny-llama-8l-test", "--steps", "3", "--warmup", "1", "--batch", "4",
                        "--max-seconds", "400"],
                       capture_output=True, text=True, timeout=450, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    r = _check(lines[0], 8, 3, 1)
    assert r["config"]["micro_batches"] == 18 and len(r["config"]["layer_ranges"]) == 8


def test_micro_batch_shrinks_when_the_kv_cache_would_not_fit(monkeypatch):
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    b = importlib.util.module_from_spec(spec)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    spec.loader.exec_module(b)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: types.SimpleNamespace(total_memory=288 * 2 ** 30))

    def fit(steps, world, model="llama-3-8b", dtype="bf16", batch=1024):
        a = types.SimpleNamespace(model=model, partition="balanced", batch=batch, prompt_len=128, steps=steps, warmup=4,
                                  kv_dtype="bf16", dtype=dtype)
        max_seq = ((a.prompt_len + a.steps + a.warmup + 2 + 31) // 32) * 32
        return b._fit_batch(a, 2 * world, max_seq, world)

    assert fit(32, 1) == 1024 and fit(32, 8) == 1024
    assert fit(32, 1, batch=2048) == 2048 and fit(32, 8, batch=2048) == 2048        # the driver's default fits at every N
    assert fit(1000, 1) == 512                              # 2 x 1024 x 1152 tokens x 128 KiB = 309 GB does not
    assert fit(1000, 8) < 1024
    # 70B on one GPU: 141 GB of bf16 weights leave room for 2 x 1024 x 192 tokens x 320 KiB, not for 2 x 1024 x 256;
    # 70 GB of e4m3 weights do leave room for that
    assert fit(32, 1, "llama-3-70b") == 1024 and fit(100, 1, "llama-3-70b") == 512
    assert fit(100, 1, "llama-3-70b", "fp8") == 1024

    # several stages: the two spare micro-batches go before the micro-batch is halved
    def schedule(steps, world=8, batch=2048):
        a = types.SimpleNamespace(model="llama-3-8b", partition="balanced", batch=batch, prompt_len=128, steps=steps, warmup=10,
                                  kv_dtype="bf16", dtype="bf16")
        max_seq = ((a.prompt_len + a.steps + a.warmup + 54 + 2 + 31) // 32) * 32
        return b._fit_schedule(a, 2 * world + 2, max_seq, world, 2)

    assert schedule(50) == (2048, 18)                       # the usual run: everything fits
    assert schedule(150) == (2048, 16)                      # 18 x 2048 x 352 tokens x 20 KiB = 247 GiB do not fit, 16 x ... do
    assert schedule(400)[0] < 2048                          # nothing but a smaller micro-batch helps


def test_overlap_analysis_reads_a_two_stream_kernel_trace(tmp_path, capsys):
    """bench/overlap_analysis.py on a synthetic rocprofv3 kernel trace: kernel classes from names and grids,
    O told from down by what precedes it on its stream, and the overlap shares of two streams."""
    spec = importlib.util.spec_from_file_location("overlap_analysis", os.path.join(ROOT, "bench", "overlap_analysis.py"))
    oa = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(oa)
    assert oa.klass("void dlia::gemm_pingpong_kernel<false, false>(dlia::GemmParams)", 458752, 512) == "gate_up"
    assert oa.klass("void dlia::gemm_pingpong_kernel<false, true>(dlia::GemmParams)", 2064384, 512) == "lm_head"
    assert oa.klass("void dlia::paged_attn_lean_kernel<128, false>(dlia::AttnParams)", 1048576, 64) == "attn"
    head = ("Kind,Agent_Id,Queue_Id,Stream_Id,Thread_Id,Dispatch_Id,Kernel_Id,Kernel_Name,Correlation_Id,Start_Timestamp,"
            "End_Timestamp,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Workgroup_Size_X,"
            "Workgroup_Size_Y,Workgroup_Size_Z,Grid_Size_X,Grid_Size_Y,Grid_Size_Z")
    gemm, attn = "void dlia::gemm_pingpong_kernel<false, false>(dlia::GemmParams)", "void dlia::paged_attn_lean_kernel<128, false>(dlia::AttnParams)"

    def row(i, stream, name, t0, t1, grid, wg):
        return f'"KERNEL_DISPATCH","Agent 1",{stream},{stream},1,{i},1,"{name}",{i},{t0},{t1},0,0,64,0,32,{wg},1,1,{grid},1,1'

    lines = [head,
             row(1, 1, attn, 1_000_000, 1_300_000, 1048576, 64),        # stream 1: attention, then O (128 workgroups)
             row(2, 1, gemm, 1_300_000, 1_400_000, 65536, 512),
             row(3, 2, gemm, 1_000_000, 1_400_000, 458752, 512),        # stream 2: gate/up beside both, then down
             row(4, 2, gemm, 1_400_000, 1_600_000, 65536, 512)]
    d = tmp_path / "trace"
    d.mkdir()
    (d / "1_kernel_trace.csv").write_text("\n".join(lines) + "\n")
    sys.argv = ["overlap_analysis.py", str(d), "10"]
    oa.main()
    out = capsys.readouterr().out
    stats = {l.split()[0]: l for l in out.splitlines() if l and l.split()[0] in ("gate_up", "attn", "o", "down")}
    assert set(stats) == {"gate_up", "attn", "o", "down"}
    assert "attn:75%" in stats["gate_up"] and "o:25%" in stats["gate_up"]
    assert "gate_up:100%" in stats["attn"] and "gate_up:100%" in stats["o"]
    assert float(stats["down"].split()[3]) == 100.0                       # down ran alone
