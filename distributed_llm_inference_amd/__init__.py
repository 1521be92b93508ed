"""MI355X-native layer-split (pipeline-parallel) LLM inference engine.

Sub-packages: `models` (specs, partitioning, weights, stage executors), `ops` (HIP kernels for
gfx950 and their torch references), `parallel` (stage-to-stage transports, pipeline runtime,
launchers), `runtime` (scheduler/engine, KV blocks, sampler, tokenizer), `serving` (orchestrator
and worker HTTP APIs, client, CLI), `utils` (native host runtime bindings, tracing).
"""
__version__ = "0.1.0"

import os as _os

# Process-wide GPU runtime settings; both are read when HIP initialises (the first GPU call, not
# `import torch`).
#  * RCCL and GPU tensors shared between processes need dmabuf IPC on hosts whose driver has no
#    legacy IPC path; without it the first cross-process handle fails with "invalid argument".
#  * HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (4 by default, and 4 is what
#    these machines export). How many are right depends on what the streams do, measured on MI355X:
#      - one stage in the process (a single-GPU server, bench.py at one GPU): 4, 8 and 16 give the same;
#      - a stage process of a multi-stage pipeline over RCCL hops drives its decode streams plus one stream
#        per hop, and a receive kernel that spins until its peer sends holds back whatever compute stream
#        shares its queue: two TinyLlama stage processes decode a token in 3.8 ms with 4 queues, 1.7 ms with 8;
#      - stage processes that SHARE one card (rehearsals: DLIA_SHARE_GPU=1) also share its queue slots:
#        4 ranks x 8 queues oversubscribe them (116 -> 399 ms per step), 2 x 16 too (1.7 -> 26 ms per token);
#      - several stages as threads of ONE process over in-process hops (tests, rehearsals) exchange nothing
#        but events between streams, and those resolve faster inside a shared queue than across queues:
#        1.5 ms per TinyLlama token with 4 queues, 6.2 ms with 8.
#    So a smaller exported value is raised to 8 exactly for the second case — a rank of a multi-rank job
#    (WORLD_SIZE > 1, or a stage process spawned for a pipeline of several stages: DLIA_PP_RANKS) that has its
#    GPU to itself — and left alone otherwise; DLIA_KEEP_HW_QUEUES=1 leaves it alone always.


def ensure_hw_queues(minimum: int = 8) -> None:
    env = _os.environ

    def num(name):
        try:
            return int(env.get(name, "0") or 0)
        except ValueError:
            return 0

    if env.get("DLIA_KEEP_HW_QUEUES") == "1" or env.get("DLIA_SHARE_GPU") == "1":
        return
    if max(num("WORLD_SIZE"), num("DLIA_PP_RANKS")) <= 1:
        return
    if num("GPU_MAX_HW_QUEUES") < minimum:
        env["GPU_MAX_HW_QUEUES"] = str(minimum)


_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ensure_hw_queues()
