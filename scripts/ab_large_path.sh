#!/bin/bash
# A/B of the large-D path's r04 changes through pnode_set_generic_path: 2 = as shipped, 4 = 64 x 64 GEMM tile only, 5 = blocked LQ without look-ahead;
# config 5's size (filter and smoother), the D = 256 smoother, D = 512.
set -e
out=gpurun_out/${1:-ab_large}
mkdir -p $out
for cfg in "128 64 5 filter --phases" "128 64 2 filter --phases" "128 64 5 fixedpoint" "128 64 2 fixedpoint" "32 1024 5 fixedpoint" "32 1024 2 fixedpoint" "64 256 5 filter" "64 256 2 filter" "96 64 5 filter" "96 64 2 filter"; do
  timeout -k 10 300 python scripts/time_dense_paths.py $cfg >> $out/ab.log 2>&1
done
cat $out/ab.log
