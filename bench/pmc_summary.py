"""Aggregate a rocprofv3 --pmc counter_collection CSV per (kernel, grid): mean counter values."""
import collections
import csv
import glob
import sys


def main(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").replace("dlia::", "").split("(")[0][:40]
        key = (n, r.get("Grid_Size", r.get("Grid_Size_X", "")))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for v in agg.values() for c in v})
    print("kernel,grid,n," + ",".join(names))
    for k, v in sorted(agg.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
        cnt = len(next(iter(v.values())))
        print(f"{k[0]},{k[1]},{cnt}," + ",".join(f"{sum(v[c]) / max(1, len(v[c])):.4g}" if c in v else "" for c in names))


if __name__ == "__main__":
    main(sys.argv[1])
