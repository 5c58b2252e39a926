#!/bin/bash
# Run ON THE GPU BOX: kernel trace + FETCH / WRITE passes of the one-launch LQ (path 3) at 1024 and 128 trajectories.
set -e
OUT=gpurun_out/prof_r02_onelq
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT; mkdir -p $OUT
for B in 1024 128; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$B -- python3 scripts/time_dense_paths.py 32 $B 3 > $OUT/trace$B.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch$B -- python3 scripts/time_dense_paths.py 32 $B 3 > $OUT/fetch$B.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write$B -- python3 scripts/time_dense_paths.py 32 $B 3 > $OUT/write$B.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for B in (1024, 128):
    res = collections.defaultdict(dict)
    for r in csv.DictReader(open(sorted(glob.glob(f"gpurun_out/prof_r02_onelq/trace{B}/*/*kernel_stats.csv"))[-1])):
        res[r["Name"].split("(")[0]]["avg_us"] = float(r["AverageNs"]) / 1e3
    for tag, key, mul in (("fetch", "FETCH_SIZE", 2048.0), ("write", "WRITE_SIZE", 1024.0)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(sorted(glob.glob(f"gpurun_out/prof_r02_onelq/{tag}{B}/*/*counter_collection.csv"))[-1])):
            if r["Counter_Name"] == key:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * mul)
        for k, v in agg.items():
            res[k][key + "_bytes"] = sum(v[-3:]) / len(v[-3:])
    for k, v in res.items():
        if "lq3" in k or "lq2" in k:
            print(f"B={B} {k}: avg {v.get('avg_us', 0):.1f} us, read {v.get('FETCH_SIZE_bytes', 0) / 1e6:.1f} MB, write {v.get('WRITE_SIZE_bytes', 0) / 1e6:.1f} MB per launch "
                  f"({(v.get('FETCH_SIZE_bytes', 0) + v.get('WRITE_SIZE_bytes', 0)) / B / 1e6:.2f} MB per trajectory)")
PY
