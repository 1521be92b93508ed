#!/usr/bin/env python3
"""Chrome trace written by `--trace`: mean GPU compute span of a decode step, the idle gap between
consecutive GPU spans, and the host's enqueue span (steady state: second half of the trace)."""
import json
import sys

ev = json.load(open(sys.argv[1]))
ev = ev["traceEvents"] if isinstance(ev, dict) else ev
gpu = sorted((e["ts"], e["dur"]) for e in ev if e.get("ph") == "X" and e.get("name") == "compute")
host = sorted((e["ts"], e["dur"]) for e in ev if e.get("ph") == "X" and e.get("name") == "enqueue step")
gpu, host = gpu[len(gpu) // 2:], host[len(host) // 2:]
mean = lambda v: sum(v) / max(1, len(v))
gaps = [gpu[i + 1][0] - (gpu[i][0] + gpu[i][1]) for i in range(len(gpu) - 1)]
period = [gpu[i + 1][0] - gpu[i][0] for i in range(len(gpu) - 1)]
hper = [host[i + 1][0] - host[i][0] for i in range(len(host) - 1)]
print(f"gpu spans {len(gpu)}: mean {mean([d for _, d in gpu]):.1f} us, gap to next {mean(gaps):.1f} us, period {mean(period):.1f} us")
print(f"host enqueue spans {len(host)}: mean {mean([d for _, d in host]):.1f} us, period {mean(hper):.1f} us")
# how far ahead of the GPU the host enqueues: start of GPU span k minus end of its host span
lead = [g[0] - (h[0] + h[1]) for g, h in zip(gpu, host)]
print(f"gpu start minus host enqueue end: mean {mean(lead):.1f} us, min {min(lead):.1f} us")
