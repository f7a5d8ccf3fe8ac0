#!/usr/bin/env python
"""Fold a `rocprofv3 --kernel-trace` CSV (one row per dispatch, tens of MB) into a small JSON: per (kernel name, grid size)
the number of dispatches and the total / average / min / max duration in ns, plus the share of GPU time of every name.

    python tools/trace_summary.py DIR_OR_CSV OUT.json

Column names differ a little between rocprofv3 builds (Grid_Size vs Grid_Size_X/Y/Z): both are read.
"""
import csv
import glob
import json
import os
import sys


def rows_of(path):
    if os.path.isdir(path):
        files = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    else:
        files = [path]
    for f in files:
        with open(f, newline="") as fh:
            yield from csv.DictReader(fh)


def grid_of(row):
    if row.get("Grid_Size"):
        return int(row["Grid_Size"])
    g = 1
    for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"):
        g *= int(row.get(k) or 1)
    return g


def main():
    src, out = sys.argv[1], sys.argv[2]
    acc, by_name, total = {}, {}, 0
    for r in rows_of(src):
        name = r["Kernel_Name"]
        ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        key = (name, grid_of(r))
        a = acc.setdefault(key, [0, 0, None, 0, -1, 0])
        a[0] += 1
        a[1] += ns
        a[2] = ns if a[2] is None else min(a[2], ns)
        a[3] = max(a[3], ns)
        if int(r["Start_Timestamp"]) > a[4]:          # the class's LAST dispatch in time order
            a[4], a[5] = int(r["Start_Timestamp"]), ns
        b = by_name.setdefault(name, [0, 0, -1, 0, []])    # total ns, dispatches, start of the last dispatch, its duration, (start, ns) of all
        b[0] += ns
        b[1] += 1
        b[4].append((int(r["Start_Timestamp"]), ns))
        if int(r["Start_Timestamp"]) > b[2]:
            b[2], b[3] = int(r["Start_Timestamp"]), ns
        total += ns
    kernels = [{"name": k[0], "grid_threads": k[1], "dispatches": v[0], "total_ns": v[1], "avg_ns": v[1] / v[0], "min_ns": v[2], "max_ns": v[3],
                "last_ns": v[5]}
               for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])]
    # (persistent kernels launch one workgroup per CU whatever the layer: all their shapes share ONE grid size, so a class is also
    #  reported per NAME -- the template arguments separate the instances -- with the duration of its last dispatch in time order)
    names = [{"name": n, "total_ns": b[0], "dispatches": b[1], "share": b[0] / max(total, 1), "last_ns": b[3],
              "last12_ns": [d for _, d in sorted(b[4])[-12:]]}
             for n, b in sorted(by_name.items(), key=lambda kv: -kv[1][0])]
    with open(out, "w") as f:
        json.dump({"what": "rocprofv3 --kernel-trace folded by (kernel name, grid size); durations = End - Start timestamps (ns); last_ns = the "
                           "class's last dispatch in time order (bench.py's instrumented frame runs last: its launches are the last of their class)",
                   "gpu_time_ns": total, "names": names, "kernels": kernels[:400]}, f, indent=1)
    print(f"{len(acc)} (kernel, grid) classes, {sum(v[0] for v in acc.values())} dispatches, {total / 1e9:.3f} s of GPU time -> {out}")


if __name__ == "__main__":
    main()
