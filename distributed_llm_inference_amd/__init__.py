"""MI355X-native layer-split (pipeline-parallel) LLM inference engine.

Sub-packages: `models` (specs, partitioning, weights, stage executors), `ops` (HIP kernels for
gfx950 and their torch references), `parallel` (stage-to-stage transports, pipeline runtime,
launchers), `runtime` (scheduler/engine, KV blocks, sampler, tokenizer), `serving` (orchestrator
and worker HTTP APIs, client, CLI), `utils` (native host runtime bindings, tracing).
"""
__version__ = "0.1.0"

import os as _os

# Process-wide GPU runtime settings; both are read when HIP initialises (the first GPU call, not
# `import torch`), and a value the user exported wins.
#  * RCCL and GPU tensors shared between processes need dmabuf IPC on hosts whose driver has no
#    legacy IPC path; without it the first cross-process handle fails with "invalid argument".
#  * A stage of a multi-GPU pipeline drives its decode streams plus one RCCL stream per pipeline
#    edge. HIP multiplexes streams onto 4 hardware queues by default, and a receive kernel that
#    spins until its peer sends would hold back whatever compute stream shares its queue; 8
#    queues give every stream of a stage its own (no effect on a single-GPU run, measured).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
