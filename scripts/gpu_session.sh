#!/bin/bash
# Run ON THE GPU BOX through gpurun: a sequence of steps, each with its own timeout; a step that is killed or times out ends the
# session (no further GPU step is started after one), an ordinary failure (non-zero exit) is logged and the session goes on.
#   bash scripts/gpu_session.sh <tag> "<step name>|<timeout s>|<command>" ...
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; tmo=${rest%%|*}; cmd=${rest#*|}
  echo "=== $name (timeout ${tmo}s): $cmd"
  start=$(date +%s)
  timeout -k 10 $tmo bash -c "$cmd" > $OUT/$name.log 2>&1
  rc=$?
  echo "=== $name rc=$rc in $(( $(date +%s) - start ))s"; tail -n 6 $OUT/$name.log
  echo "$name rc=$rc" >> $OUT/status.txt
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "step killed: stopping the session"; exit $rc; fi
done
exit 0
