#!/bin/bash
# A/B of the batched GEMM tiles of the large-D path (pnode_set_generic_path 2: 128 x 128 double-buffered tile where the shape allows,
# 4: the 64 x 64 tile everywhere) at config 5's size, the D = 256 smoother and D = 512.
set -e
out=gpurun_out/${1:-ab_gemm}
mkdir -p $out
for cfg in "128 64 4 filter --phases" "128 64 2 filter --phases" "32 1024 4 fixedpoint" "32 1024 2 fixedpoint" "128 64 4 fixedpoint" "128 64 2 fixedpoint" "64 256 4 filter" "64 256 2 filter"; do
  timeout -k 10 300 python scripts/time_dense_paths.py $cfg >> $out/ab.log 2>&1
done
cat $out/ab.log
