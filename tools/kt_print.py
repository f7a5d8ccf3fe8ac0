#!/usr/bin/env python
"""Print the top rows of bench.py --kernel-table files: python tools/kt_print.py FILE [FILE ...] [--top N]"""
import json
import sys

top = 22
files = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--top" in sys.argv:
    top = int(sys.argv[sys.argv.index("--top") + 1])
    files = [f for f in files if f != str(top)]
for f in files:
    d = json.load(open(f))
    print(f, "conv ms per frame", round(d["conv_ms_per_frame"], 2))
    tot = 0.0
    for k, v in sorted(d["convolutions"].items(), key=lambda kv: -kv[1]["ms"])[:top]:
        tot += v["ms"]
        print("%-46s x%3d %7.3f ms %6.1f TF  cum %6.2f" % (k, v["launches"], v["ms"], v["tflops"], tot))
    for k, v in d.get("hbm_kernels", {}).items():
        print("   hbm %-44s x%2d %7.1f us %5.3f of peak" % (k, v["launches"], v["avg_us"], v["frac_of_hbm_peak"]))
