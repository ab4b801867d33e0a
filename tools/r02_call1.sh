#!/bin/bash
# Round 2, GPU call 1: un-gated parity suite + measurements of everything prepared at the end of round 1.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
LOG=$ROOT/gpurun_out/r02_call1.log
: > $LOG
run() {  # label, timeout, command...
  local label=$1 t=$2; shift 2
  echo "=== $label" | tee -a $LOG
  timeout $t "$@" >> $LOG 2>&1
  echo "    exit code $?" | tee -a $LOG
}
export ETM_TUNABLE_GEMM=0
run "full gpu parity suite (nothing gated)" 900 python -m pytest tests -q -m gpu -rA --durations=15
for rep in 1 2 3; do
  run "small-group stress + candidate paths, repetition $rep" 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "small_worker_groups or candidate_rollout or baseline_config_shapes"
done
unset ETM_TUNABLE_GEMM
V2=$ROOT/tools/diag_build/libetm_win_v2.so
for pattern in random shuffled sorted; do
  run "window pass, shipped kernel, samples=$pattern" 200 env ETM_WIN_SAMPLES=$pattern python tools/window_time.py
  run "window pass, v2 candidate, samples=$pattern" 200 env ETM_WIN_SAMPLES=$pattern ETM_DIAG_LIB=$V2 python tools/window_time.py
  run "window pass, v2 candidate + XCD chunks, samples=$pattern" 200 env ETM_WIN_SAMPLES=$pattern ETM_WIN_XCD_MAP=1 ETM_DIAG_LIB=$V2 python tools/window_time.py
done
run "window pass L=128, shipped" 200 python tools/window_time.py 128 384 4
run "window pass L=128, v2 candidate" 200 env ETM_DIAG_LIB=$V2 python tools/window_time.py 128 384 4
run "scan / loss kernels vs HBM roofline" 400 python tools/scan_roofline.py
GAE2=$ROOT/tools/diag_build/libetm_gae_v2.so
run "scan / loss kernels vs HBM roofline, GAE candidate" 400 env ETM_DIAG_LIB=$GAE2 python tools/scan_roofline.py
run "window pass candidate: attention parity tests" 400 env ETM_TUNABLE_GEMM=0 ETM_DIAG_LIB=$V2 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mha or transformer_variants or actor_critic"
run "rollout step: defaults" 200 python -u tools/rollout_profile.py
run "rollout step: native_step_launch" 200 python -u tools/rollout_profile.py native_step_launch=1
run "rollout step: four groups" 200 python -u tools/rollout_profile.py rollout_groups=4
run "rollout step: four groups + native_step_launch" 200 python -u tools/rollout_profile.py rollout_groups=4 native_step_launch=1
run "whole path, config 3: defaults" 300 python tools/config_bench.py synthetic_minigrid 3
run "whole path, config 3: v2 library + XCD chunks + sorted minibatches" 300 env ETM_DIAG_LIB=$V2 ETM_WIN_XCD_MAP=1 python tools/config_bench.py synthetic_minigrid 3 sort_minibatch=1
run "whole path, config 3: four groups + native step launch" 300 python tools/config_bench.py synthetic_minigrid 3 rollout_groups=4 native_step_launch=1
run "whole path, GTrXL L=128: defaults" 300 python tools/config_bench.py synthetic_mortar_gtrxl 2
run "whole path, GTrXL L=128: v2 library" 300 env ETM_DIAG_LIB=$V2 python tools/config_bench.py synthetic_mortar_gtrxl 2
run "bench.py" 400 python bench.py --steps 5 --warmup 2
grep -E "^===|exit code|env-steps|rollout:|folded |checksums|frac|passed|failed|error|Error" $LOG | tail -150
