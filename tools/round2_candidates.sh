#!/bin/bash
# One GPU call that measures / verifies everything that was prepared after round 1's GPU budget was spent:
#   gpurun --timeout 1500 -- 'bash tools/round2_candidates.sh'          -> gpurun_out/round2_candidates.log
# 1. env-gated parity test of the prepared rollout paths (native step launch, four worker groups)
# 2. window pass: shipped kernel vs the lower-register candidate (times + result checksums), three access patterns,
#    XCD-contiguous sample chunks
# 3. GAE / PPO-loss kernels against the HBM roofline at config and scaled sizes
# 4. rollout step with the prepared host paths
# 5. whole path (config 3 and the GTrXL L = 128 config) with the candidates switched on
# Every step runs under its own timeout and does not stop the others; about 30 process starts, roughly 15 GPU-minutes in all --
# delete the blocks that are not needed.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
LOG=$ROOT/gpurun_out/round2_candidates.log
: > $LOG
run() {  # label, timeout, command...
  local label=$1 t=$2; shift 2
  echo "=== $label" | tee -a $LOG
  timeout $t "$@" >> $LOG 2>&1
  echo "    exit code $?" | tee -a $LOG
}
export ETM_TUNABLE_GEMM=0
run "candidate rollout paths (parity)" 600 env ETM_TEST_CANDIDATES=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "candidate or fast_paths or teacher_forced"   # includes the library collective at world size 1
for rep in 1 2 3; do
  run "BASELINE config shapes (GTrXL included), repetition $rep" 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "baseline_config_shapes_train"
done
unset ETM_TUNABLE_GEMM

run "build window variants" 600 bash tools/diag_variants.sh window
V2=$ROOT/tools/diag_build/libetm_win_v2.so
for pattern in random shuffled sorted; do
  run "window pass, shipped kernel, samples=$pattern" 300 env ETM_WIN_SAMPLES=$pattern python tools/window_time.py
  run "window pass, v2 candidate, samples=$pattern" 300 env ETM_WIN_SAMPLES=$pattern ETM_DIAG_LIB=$V2 python tools/window_time.py
  run "window pass, v2 candidate + XCD chunks, samples=$pattern" 300 env ETM_WIN_SAMPLES=$pattern ETM_WIN_XCD_MAP=1 ETM_DIAG_LIB=$V2 python tools/window_time.py
done
run "window pass L=128, shipped" 300 python tools/window_time.py 128 384 4
run "window pass L=128, v2 candidate" 300 env ETM_DIAG_LIB=$V2 python tools/window_time.py 128 384 4
run "window pass v2 phase trace" 300 env ETM_DIAG_LIB=$ROOT/tools/diag_build/libetm_win_v2trace.so python tools/window_time.py

run "scan / loss kernels vs HBM roofline" 600 python tools/scan_roofline.py
run "build GAE candidate" 300 bash tools/diag_variants.sh gae
GAE2=$ROOT/tools/diag_build/libetm_gae_v2.so
run "scan / loss kernels vs HBM roofline, GAE candidate (register prefetch of the next time tile)" 600 env ETM_DIAG_LIB=$GAE2 python tools/scan_roofline.py
run "GAE candidate: bit-exact parity tests" 300 env ETM_DIAG_LIB=$GAE2 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gae"
run "window pass candidate: attention parity tests" 600 env ETM_DIAG_LIB=$V2 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mha or transformer or actor_critic or teacher_forced"

run "rollout step: defaults" 300 python -u tools/rollout_profile.py
run "rollout step: native_step_launch" 300 python -u tools/rollout_profile.py native_step_launch=1
run "rollout step: four groups" 300 python -u tools/rollout_profile.py rollout_groups=4
run "rollout step: four groups + native_step_launch" 300 python -u tools/rollout_profile.py rollout_groups=4 native_step_launch=1

run "whole path, config 3: defaults" 400 python tools/config_bench.py synthetic_minigrid 3
run "whole path, config 3: v2 library + XCD chunks + sorted minibatches" 400 env ETM_DIAG_LIB=$V2 ETM_WIN_XCD_MAP=1 python tools/config_bench.py synthetic_minigrid 3 sort_minibatch=1
run "whole path, config 3: four groups + native step launch" 400 python tools/config_bench.py synthetic_minigrid 3 rollout_groups=4 native_step_launch=1
run "whole path, GTrXL L=128: defaults" 400 python tools/config_bench.py synthetic_mortar_gtrxl 2
run "whole path, GTrXL L=128: v2 library" 400 env ETM_DIAG_LIB=$V2 python tools/config_bench.py synthetic_mortar_gtrxl 2

grep -E "^===|exit code|env-steps|rollout:|folded |checksums|frac|passed|failed|error" $LOG | tail -120
