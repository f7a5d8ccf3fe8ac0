#!/usr/bin/env python
"""Fold rocprofv3 --pmc passes over tools/conv_bench.py into profiles/<round>/traffic.json.

    python tools/pmc_traffic.py OUT.json "kernel label"=FETCH_DIR,WRITE_DIR [...]
Each directory holds the csv output of ONE pass (`rocprofv3 --kernel-trace --pmc FETCH_SIZE ...` or `... WRITE_SIZE`,
never combined with other traces).  FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE is doubled for gfx950 as
/opt/skills/guides/MI355X_MICROARCH.md prescribes.  Only convolution kernels are averaged (the benchmark's fill
kernels are ignored)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mean_counter(directory, counter):
    vals = []
    for path in glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter and "conv_" in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for convolution kernels under {directory}")
    return sum(vals) / len(vals), len(vals)


def main():
    out, kernels = sys.argv[1], {}
    for spec in sys.argv[2:]:
        label, dirs = spec.split("=")
        fdir, wdir = dirs.split(",")
        fetch, nf = mean_counter(fdir, "FETCH_SIZE")
        write, nw = mean_counter(wdir, "WRITE_SIZE")
        kernels[label] = {"FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write, "launches_averaged": [nf, nw],
                          "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0}
    from bench import kernel_source_stamp       # bench.py refuses this file once the kernel sources have changed
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on tools/conv_bench.py (VC_AUTOTUNE=0, tile "
                       "configuration pinned to the one bench.py's autotuner settles on); FETCH_SIZE doubled per the gfx950 correction",
               "kernel_source_stamp": kernel_source_stamp(),
               "kernels": kernels}, open(out, "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
