"""MI355X-native layer-split (pipeline-parallel) LLM inference engine.

Sub-packages: `models` (specs, partitioning, weights, stage executors), `ops` (HIP kernels for
gfx950 and their torch references), `parallel` (stage-to-stage transports, pipeline runtime,
launchers), `runtime` (scheduler/engine, KV blocks, sampler, tokenizer), `serving` (orchestrator
and worker HTTP APIs, client, CLI), `utils` (native host runtime bindings, tracing).
"""
__version__ = "0.1.0"
