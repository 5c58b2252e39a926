#!/usr/bin/env python3
"""Summarise a scripts/collect_profile.sh run: per-kernel average duration (rocprofv3 --kernel-trace --stats)
and PMC-derived quantities per launch. HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream, so it is
doubled; the passes are separate --pmc runs. Writes <dir>/summary.json next to the text summary on stdout."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").replace("pnode::", "")


def main(out):
    res = collections.OrderedDict()
    stats = sorted(glob.glob(f"{out}/trace/*/*kernel_stats.csv"), key=os.path.getmtime, reverse=True)  # newest run
    if stats:
        for r in csv.DictReader(open(stats[0])):
            k = short(r["Name"])
            if k.startswith("dense") or k.startswith("iso"):
                res.setdefault(k, {})["avg_us"] = float(r["AverageNs"]) / 1e3
                res[k]["calls"] = int(r["Calls"])
                res[k]["pct"] = float(r["Percentage"])
    for tag in ("sq", "fetch", "write"):
        files = sorted(glob.glob(f"{out}/{tag}/*/*counter_collection.csv"), key=os.path.getmtime, reverse=True)
        if not files:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(files[0])):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if k in res:
                for c, v in cs.items():
                    tail = v[-10:]  # steady-state launches
                    res[k][c] = sum(tail) / len(tail)
    for k, v in res.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_read_bytes"] = 2.0 * v["FETCH_SIZE"] * 1024.0
            v["hbm_write_bytes"] = v["WRITE_SIZE"] * 1024.0
            v["hbm_bytes_per_launch"] = v["hbm_read_bytes"] + v["hbm_write_bytes"]
            v["hbm_GBps"] = v["hbm_bytes_per_launch"] / (v["avg_us"] * 1e-6) / 1e9 if "avg_us" in v else None
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            v["mfma_util"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (v["GRBM_GUI_ACTIVE"] / 8.0)
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in v:
            v["mfma_flops_per_launch"] = v["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
    json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
    print(f"{'kernel':44s} {'avg us':>9s} {'%':>6s} {'HBM MB/launch':>14s} {'GB/s':>8s} {'MFMA util':>10s} {'MFMA GF':>8s}")
    for k, v in res.items():
        print(f"{k[:44]:44s} {v.get('avg_us', 0):9.1f} {v.get('pct', 0):6.2f} {v.get('hbm_bytes_per_launch', 0) / 1e6:14.1f} "
              f"{(v.get('hbm_GBps') or 0):8.0f} {v.get('mfma_util', 0):10.3f} {v.get('mfma_flops_per_launch', 0) / 1e9:8.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
