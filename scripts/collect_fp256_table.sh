#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace of the dense fixed-point smoother at the headline shape (Brusselator N=32, D=256, B=1024; large-D kernels),
# summarised per kernel -> gpurun_out/fp256_<tag>/table.txt (copy to profiles/<tag>_dense_fixedpoint_d256_kernel_table*.txt).
set -e
TAG=${1:-r04}
OUT=gpurun_out/fp256_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 scripts/time_dense_paths.py 32 1024 0 fixedpoint > $OUT/run.log 2> $OUT/run.err
{
  echo "rocprofv3 --kernel-trace of scripts/time_dense_paths.py 32 1024 0 fixedpoint (Brusselator N=32, D=256, B=1024; 1 warm-up + 4 timed smoother attempts + 1 interpolation; large-D kernels): $(grep 'ms/step' $OUT/run.log | tail -1)"
  python3 scripts/parse_kernel_trace.py $OUT/tr 60
} > $OUT/table.txt
rm -rf $OUT/tr
head -12 $OUT/table.txt
