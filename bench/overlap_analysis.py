"""Two-stream overlap analysis of a rocprofv3 kernel trace of `bench.py` (2 decode streams).

For every kernel class: how long it runs when another stream's kernel of each class is on the chip at the
same time, against how long it runs alone. Usage: python bench/overlap_analysis.py TRACE_DIR [last_n_ms]
"""
import collections
import csv
import glob
import sys


def klass(name, grid, wg):
    if "gemm_pingpong" in name or "gemm_quad" in name:
        wgs = grid // wg
        if "true>" in name:
            return "lm_head"
        return {896: "gate_up", 192: "qkv", 128: "o_or_down"}.get(wgs, f"gemm{wgs}")
    for key, val in (("paged_attn", "attn"), ("rope_kvwrite", "rope"), ("norm_kernel", "norm"), ("sample_kernel", "sample"),
                     ("embed", "embed")):
        if key in name:
            return val
    return "other"


def main():
    d = sys.argv[1]
    last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=lambda p: -len(open(p).read()))[0]
    rows = []
    for r in csv.DictReader(open(f)):
        g, w = int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], klass(r["Kernel_Name"], g, w)))
    tmax = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= tmax - last_ms * 1e6]
    rows.sort()
    # o vs down: in a stream's order, the gemm of 128 workgroups right after attention is O, the other is down
    prev = {}
    fixed = []
    for s, e, st, k in rows:
        if k == "o_or_down":
            k = "o" if prev.get(st) == "attn" else "down"
        prev[st] = k
        fixed.append((s, e, st, k))
    rows = fixed
    streams = collections.Counter(r[2] for r in rows)
    print("kernels per stream:", dict(streams), " window %.1f ms" % ((tmax - rows[0][0]) / 1e6))
    # per kernel: overlap with other streams' kernels by class
    stats = collections.defaultdict(lambda: {"n": 0, "dur": 0.0, "alone": 0.0, "with": collections.Counter()})
    evs = rows
    j0 = 0
    for i, (s, e, st, k) in enumerate(evs):
        a = stats[k]
        a["n"] += 1
        a["dur"] += (e - s)
        covered = 0
        while j0 < len(evs) and evs[j0][1] < s - 5_000_000:
            j0 += 1
        for (s2, e2, st2, k2) in evs[j0:]:
            if s2 >= e:
                break
            if st2 == st:
                continue
            ov = min(e, e2) - max(s, s2)
            if ov > 0:
                a["with"][k2] += ov
                covered += ov
        a["alone"] += max(0, (e - s) - covered)
    print(f"{'class':10s} {'n':>5s} {'mean us':>9s} {'alone %':>8s}  overlap share by the other stream's class")
    for k, a in sorted(stats.items(), key=lambda kv: -kv[1]["dur"]):
        tot = a["dur"]
        w = "  ".join(f"{k2}:{100 * v / tot:.0f}%" for k2, v in a["with"].most_common(6))
        print(f"{k:10s} {a['n']:5d} {tot / a['n'] / 1e3:9.1f} {100 * a['alone'] / tot:8.1f}  {w}")
    busy_any = 0
    cur_s, cur_e = None, None
    for s, e, _, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy_any += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy_any += cur_e - cur_s
    print("chip busy with at least one kernel: %.1f %% of the window; sum of kernel time / window = %.2f"
          % (100 * busy_any / (tmax - rows[0][0]), sum(e - s for s, e, _, _ in rows) / (tmax - rows[0][0])))


if __name__ == "__main__":
    main()


def rates(d, last_ms=300.0, alone=None):
    """Least-squares progress rates: every instance of a class does one unit of work, sum over partner
    classes of rate[partner] x (time overlapped with that partner) = 1. Printed as the equivalent
    duration 1 / rate (us) the kernel would take if it ran beside that partner class only."""
    import numpy as np
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=lambda p: -len(open(p).read()))[0]
    rows = []
    for r in csv.DictReader(open(f)):
        g, w = int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], klass(r["Kernel_Name"], g, w)))
    tmax = max(r[1] for r in rows)
    rows = sorted(r for r in rows if r[0] >= tmax - last_ms * 1e6)
    prev, fixed = {}, []
    for s, e, st, k in rows:
        if k == "o_or_down":
            k = "o" if prev.get(st) == "attn" else "down"
        prev[st] = k
        fixed.append((s, e, st, k))
    rows = fixed
    groups = {"gate_up": "gemm", "down": "gemm", "o": "gemm", "qkv": "gemm", "lm_head": "gemm", "attn": "attn"}
    partners = ["gemm", "attn", "small", "alone"]
    for target in ("gate_up", "down", "o", "qkv", "attn"):
        A, b = [], []
        for i, (s, e, st, k) in enumerate(rows):
            if k != target:
                continue
            ov = dict.fromkeys(partners, 0.0)
            covered = 0
            for (s2, e2, st2, k2) in rows:
                if st2 == st or e2 <= s or s2 >= e:
                    continue
                o = min(e, e2) - max(s, s2)
                ov[groups.get(k2, "small")] += o
                covered += o
            ov["alone"] = max(0, (e - s) - covered)
            A.append([ov[p] / 1e3 for p in partners])
            b.append(1.0)
        A, b = np.array(A), np.array(b)
        x, *_ = np.linalg.lstsq(A, b, rcond=None)
        print(f"{target:8s} n={len(b):4d} equivalent duration beside " +
              "  ".join(f"{p}: {1 / v:7.1f} us" if v > 1e-6 else f"{p}:     n/a" for p, v in zip(partners, x)))


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "rates":
    rates(sys.argv[1], float(sys.argv[2]))
