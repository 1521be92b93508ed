"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel (and per-grid) time of the steady state."""
import collections
import csv
import glob
import sys


def main(d, last=0):
    f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if last:
        rows = rows[-last:]
    agg = collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        n = n.replace("void ", "").replace("dlia::", "").split("(")[0][:44]
        key = (n, r["Grid_Size_X"] + "x" + r["Grid_Size_Y"], r["Workgroup_Size_X"])
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
    print(f"kernels {len(rows)}  busy {tot:.0f} us  span {span:.0f} us  idle {100 * (1 - tot / span):.1f}%")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        print(f"{k[0]:44s} grid {k[1]:>12s} wg {k[2]:>4s} n={len(v):5d} med {v2[len(v2) // 2]:8.1f} us  total {sum(v):9.0f} us {100 * sum(v) / tot:5.1f}%")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
