#!/usr/bin/env python3
"""How much of a rocprofv3 --kernel-trace run had two kernels in flight (look-ahead of the blocked LQ): overlap_from_trace.py OUTDIR [kernel substring]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void pnode::lg::", "").replace("pnode::", ""), r.get("Queue_Id", "?"))
        for r in csv.DictReader(open(f))]
rows.sort()
t0 = rows[0][0]
busy = overlap = 0
ev = sorted([(s, 1) for s, e, *_ in rows] + [(e, -1) for s, e, *_ in rows])
depth, last = 0, ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: overlap += t - last
    depth += d; last = t
print(f"kernels {len(rows)}, span {(rows[-1][1] - t0) / 1e6:.2f} ms, >= 1 kernel in flight {busy / 1e6:.2f} ms, >= 2 in flight {overlap / 1e6:.2f} ms; queues {sorted(set(r[3] for r in rows))}")
pat = sys.argv[2] if len(sys.argv) > 2 else "lq_panel"
sel = [r for r in rows if pat in r[2]]
for s, e, n, q in sel[40:52]:
    others = [(s2, e2, n2, q2) for s2, e2, n2, q2 in rows if s2 < e and e2 > s and (s2, e2, n2) != (s, e, n)]
    print(f"  {n[:40]:40s} q{q} start {(s - t0) / 1e3:10.1f} us dur {(e - s) / 1e3:7.1f} us | concurrent: " + ", ".join(f"{n2[:28]} q{q2} {(min(e, e2) - max(s, s2)) / 1e3:.0f}us" for s2, e2, n2, q2 in others[:3]))
