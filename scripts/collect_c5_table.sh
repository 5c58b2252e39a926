#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace of one config-5-sized dense step (Brusselator N=128, D=1024, B=64), filter and
# fixed-point smoother, summarised per kernel -> gpurun_out/c5_<tag>/table.txt (copy to profiles/<tag>_config5_kernel_table.txt).
set -e
TAG=${1:-r03}
OUT=gpurun_out/c5_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT
mkdir -p $OUT
{
  echo "rocprofv3 --kernel-trace of scripts/time_dense_paths.py 128 64 0 {filter,fixedpoint}; summed over grid sizes (top 60 entries)"
  for strat in filter fixedpoint; do
    rocprofv3 --kernel-trace --output-format csv -d $OUT/$strat -- python3 scripts/time_dense_paths.py 128 64 0 $strat > $OUT/$strat.log 2> $OUT/$strat.err
    echo "== config 5 size (Brusselator N=128, D=1024, B=64), $strat: $(grep 'ms/step' $OUT/$strat.log | tail -1)"
    python3 scripts/parse_kernel_trace.py $OUT/$strat 60
  done
} > $OUT/table.txt
cat $OUT/table.txt | head -30
