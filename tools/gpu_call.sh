#!/bin/bash
# Generic "one GPU call" runner: every argument is "label::timeout::command"; output -> gpurun_out/<name>.log (first argument = name).
#   gpurun --timeout 1200 -- 'bash tools/gpu_call.sh call2 "tests::900::python -m pytest tests -q -m gpu" "bench::400::python bench.py"'
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
NAME=$1; shift
LOG=$ROOT/gpurun_out/$NAME.log
: > $LOG
for spec in "$@"; do
  label=${spec%%::*}; rest=${spec#*::}; t=${rest%%::*}; cmd=${rest#*::}
  echo "=== $label" | tee -a $LOG
  timeout $t bash -c "$cmd" >> $LOG 2>&1
  echo "    exit code $?" | tee -a $LOG
done
grep -E "^===|exit code|passed|failed|FAILED|Error" $LOG | tail -80
