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
#  * A stage of a multi-GPU pipeline drives its decode streams plus one RCCL stream per pipeline
#    edge. HIP multiplexes streams onto 4 hardware queues by default, and a receive kernel that
#    spins until its peer sends would hold back whatever compute stream shares its queue; 8
#    queues give every stream of a stage its own (no effect on a single-GPU run, measured). A
#    smaller exported value is raised to 8 (HIP's own default of 4, exported as such, is the common
#    case; two TinyLlama stage processes: 3.8 ms per token with 4 queues, 1.7 ms with 8), a larger one
#    is kept, DLIA_KEEP_HW_QUEUES=1 keeps whatever is there — and so does DLIA_SHARE_GPU=1: stage
#    processes that share one card also share its queue slots, and 4 ranks x 8 queues oversubscribe
#    them (116 -> 399 ms per step in a 4-rank rehearsal on one GPU).


def ensure_hw_queues(minimum: int = 8) -> None:
    env = _os.environ
    if env.get("DLIA_KEEP_HW_QUEUES") == "1" or env.get("DLIA_SHARE_GPU") == "1":
        return
    try:
        cur = int(env.get("GPU_MAX_HW_QUEUES", "0") or 0)
    except ValueError:
        cur = 0
    if cur < minimum:
        env["GPU_MAX_HW_QUEUES"] = str(minimum)


_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ensure_hw_queues()
