#!/bin/bash
# PMC passes for kernel #1 at the training shape (run on the MI355X box; counters in separate passes, no tracing domains
# besides --kernel-trace, as the profiling guide prescribes).  Output: gpurun_out/pmc/<pass>/*counter_collection.csv
set -u
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
IMPL=${ETM_ATTENTION:-folded}
export ETM_ATTENTION=$IMPL
OUT=$ROOT/gpurun_out/pmc_$IMPL
mkdir -p $OUT
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmc_${IMPL}_$tag -o pmc -- python $ROOT/tools/mha_shape_run.py 3 > /tmp/pmc_${IMPL}_$tag.log 2>&1
  mkdir -p $OUT/$tag
  for f in $(find /tmp/pmc_${IMPL}_$tag -name "*counter_collection.csv"); do cp $f $OUT/$tag/; done
  tail -1 /tmp/pmc_${IMPL}_$tag.log
done
ls -R $OUT | head -30
