#!/usr/bin/env python3
"""Per-kernel, per-grid-size totals from a rocprofv3 --kernel-trace CSV directory: parse_kernel_trace.py OUT [rows]."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void pnode::lg::", "").replace("pnode::", "")
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[key][0] += 1; agg[key][1] += d
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]
for k, (n, t) in rows:
    print(f"{t/1e3:9.2f} ms  n={n:4d}  avg={t/n:9.1f} us  {k}")
