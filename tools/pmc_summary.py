#!/usr/bin/env python
"""Average the rocprofv3 --pmc counters of the convolution kernels found under one or more output directories.

    python tools/pmc_summary.py OUT.json LABEL=DIR[,DIR...] ...
Each DIR holds the csv output of ONE `rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/conv_bench.py ...`
pass (counter passes are never combined with other trace domains).  Adds the derived matrix-pipe utilisation
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) when both counters are present."""
import csv
import glob
import json
import sys


def main():
    out, result = sys.argv[1], {}
    for spec in sys.argv[2:]:
        label, dirs = spec.split("=")
        sums, counts, per = {}, {}, {}
        for d in dirs.split(","):
            for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(path)):
                    if "conv_" not in row["Kernel_Name"]:
                        continue
                    k = row["Counter_Name"]
                    sums[k] = sums.get(k, 0.0) + float(row["Counter_Value"])
                    counts[k] = counts.get(k, 0) + 1
                    # the same kernel instance serves several shapes: split by grid size as well
                    name = f'{row["Kernel_Name"]} grid={row.get("Grid_Size", "?")}'
                    ks, kc = per.setdefault(name, ({}, {}))
                    ks[k] = ks.get(k, 0.0) + float(row["Counter_Value"])
                    kc[k] = kc.get(k, 0) + 1

        def derive(a):
            if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "GRBM_GUI_ACTIVE" in a:
                a["mfma_busy_fraction"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            return a
        avg = derive({k: sums[k] / counts[k] for k in sums})
        avg["launches_averaged"] = max(counts.values()) if counts else 0
        avg["per_kernel"] = {name: derive({k: ks[k] / kc[k] for k in ks}) for name, (ks, kc) in per.items()}
        result[label] = avg
    json.dump({"note": "rocprofv3 --pmc on tools/conv_bench.py (VC_AUTOTUNE=0), per-launch averages over the convolution "
                       "kernel dispatches; GRBM_GUI_ACTIVE is summed over the 8 XCDs", "kernels": result}, open(out, "w"), indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
