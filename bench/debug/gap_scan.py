#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV and reports the idle gaps between consecutive kernels:
how much of the wall time of the busiest stretch the GPU spent with no kernel running."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
tail = rows[len(rows) // 2:]                       # steady state: second half
busy = sum(e - s for s, e, _ in tail)
wall = tail[-1][1] - tail[0][0]
gaps = [(tail[i + 1][0] - tail[i][1], tail[i][2][:50], tail[i + 1][2][:50]) for i in range(len(tail) - 1)]
big = sorted((g for g in gaps if g[0] > 15000), reverse=True)
print(f"kernels {len(tail)}  wall {wall / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  ({100 * busy / wall:.1f}%)")
print(f"gaps > 15 us: {len(big)}, total {sum(g[0] for g in big) / 1e6:.2f} ms")
from collections import Counter
c = Counter((a, b) for _, a, b in big)
for (a, b), n in c.most_common(8):
    tot = sum(g[0] for g in big if (g[1], g[2]) == (a, b))
    print(f"  {n:5d} x  avg {tot / n / 1e3:7.1f} us   after {a}  ->  before {b}")
