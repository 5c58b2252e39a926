#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace + three separate PMC passes of the default bench.
# Output under gpurun_out/prof_<tag>/ ; summarise with scripts/summarize_profile.py and copy to profiles/.
set -e
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT
mkdir -p $OUT
ARGS="bench.py --steps 20 --warmup 15 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.json 2> $OUT/sq.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.json 2> $OUT/write.err
python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
