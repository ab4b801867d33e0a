#!/bin/bash
# Builds diagnostic variants of libetm_hip.so (ablation flags / s_memtime trace) into tools/diag_build/.  Tools only:
# the product library is always built by csrc/Makefile without any ETM_DIAG_* flag.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
SRC=$REPO/episodic-transformer-memory-ppo_amd/csrc
OUT=$REPO/tools/diag_build
mkdir -p $OUT
make -C $SRC -j8 >/dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable"
OTHERS=$(ls $SRC/build/*.o | grep -v -e mha_fwd.o -e mha_bwd.o)
build() {  # name, flags...
  name=$1; shift
  ( /opt/rocm/bin/hipcc $FLAGS "$@" -c $SRC/mha_fwd.hip -o $OUT/fwd_$name.o &
    /opt/rocm/bin/hipcc $FLAGS "$@" -c $SRC/mha_bwd.hip -o $OUT/bwd_$name.o & wait )
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/fwd_$name.o $OUT/bwd_$name.o $OTHERS -o $OUT/libetm_$name.so
  rm -f $OUT/fwd_$name.o $OUT/bwd_$name.o
  echo built $name
}
if [ "$1" = "window" ]; then
  for v in base loadonly trace; do
    fl=""; [ $v = loadonly ] && fl="-DETM_DIAG_LOAD_ONLY"; [ $v = trace ] && fl="-DETM_DIAG_TRACE"
    /opt/rocm/bin/hipcc $FLAGS $fl -c $SRC/window_attn.hip -o $OUT/win_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/win_$v.o $(ls $SRC/build/*.o | grep -v window_attn.o) -o $OUT/libetm_win_$v.so
    rm -f $OUT/win_$v.o; echo built win_$v
  done
  exit 0
fi
if [ "$1" = "prio" ]; then
  build base &
  build stage3 -DETM_PRIO_OTHER=3 -DETM_PRIO_MFMA=0 &
  build stage1 -DETM_PRIO_OTHER=1 -DETM_PRIO_MFMA=0 &
  build mfma3 -DETM_PRIO_OTHER=0 -DETM_PRIO_MFMA=3 &
  build mfma1 -DETM_PRIO_OTHER=0 -DETM_PRIO_MFMA=1 &
  wait
  build trace_stage3 -DETM_DIAG_TRACE -DETM_PRIO_OTHER=3 -DETM_PRIO_MFMA=0
  exit 0
fi
build base &
build trace -DETM_DIAG_TRACE &
build noloads -DETM_DIAG_NO_STAGE_LOADS &
build nostores -DETM_DIAG_NO_STAGE_STORES &
wait
build nostage -DETM_DIAG_NO_STAGE_LOADS -DETM_DIAG_NO_STAGE_STORES &
build nostage_nobar -DETM_DIAG_NO_STAGE_LOADS -DETM_DIAG_NO_STAGE_STORES -DETM_DIAG_NO_BARRIERS &
build nobar -DETM_DIAG_NO_BARRIERS &
build nomfma -DETM_DIAG_SKIP_MFMA &
wait
