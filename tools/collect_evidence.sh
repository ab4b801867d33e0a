#!/bin/bash
# One GPU call that collects the round's evidence into gpurun_out/<round>/ (copy what is to be judged into profiles/):
#   [SUITE=1] bash tools/collect_evidence.sh r03 [pmc targets...]      (SUITE=1: the GPU test suite first, its summary line -> gpu_suite.txt)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${1:-r06}; shift
TARGETS=${@:-encoder grouped_dw rollout_step window_sorted}
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p $OUT
if [ "${SUITE:-0}" = 1 ]; then
  echo "== pytest -m gpu"; rm -f $OUT/tf_measured.jsonl; (cd $ROOT && ETM_QUIET=1 ETM_TF_MEASURE_LOG=$OUT/tf_measured.jsonl timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $OUT/gpu_suite.txt)
fi
cd /tmp && export TMPDIR=/tmp
echo "== bench (the driver's command shape)"; timeout 900 python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --full-json $OUT/bench_full.json > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.err; wc -c $OUT/bench.json
echo "== rocprofv3 --kernel-trace --stats of the bench command"
rm -rf /tmp/prof_stats; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o b -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-rooflines > /tmp/prof_stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null; head -8 $OUT/bench_kernel_stats.csv | cut -c1-160
echo "== kernel trace of one optimisation phase"
rm -rf /tmp/prof_trace; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_trace -o b -- python $ROOT/bench.py --steps 2 --warmup 3 --no-rooflines --no-cpu-baseline --no-profile > /tmp/prof_trace.log 2>&1
python $ROOT/tools/train_phase_breakdown.py /tmp/prof_trace/b_results.db $OUT/train_phase_kernels.csv --sequence > $OUT/train_phase_timeline.txt 2>&1; head -3 $OUT/train_phase_timeline.txt; grep "one minibatch step" $OUT/train_phase_timeline.txt
echo "== PMC passes: $TARGETS"
ROUND=$ROUND bash $ROOT/tools/run_pmc.sh $TARGETS 2>&1 | tail -20
cd $ROOT && python tools/pmc_summarize.py gpurun_out/pmc_$ROUND $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1; tail -25 $OUT/pmc_summary.txt | cut -c1-220
echo "== other BASELINE shapes"
timeout 300 python tools/config_bench.py synthetic_cartpole 3 2>&1 | tail -1 | tee $OUT/config2.txt
timeout 300 python tools/config_bench.py synthetic_mortar_gtrxl 3 2>&1 | tail -1 | tee $OUT/config5.txt
echo "== direct observation rows: every float of 12 rollouts against a host-side copy, with and without the optimisation phase in between"
timeout 300 python tools/direct_rows_soak.py 12 train=1 2>&1 | tail -1 | tee $OUT/direct_rows_soak.txt
timeout 300 python tools/direct_rows_soak.py 12 train=0 2>&1 | tail -1 | tee -a $OUT/direct_rows_soak.txt
echo "== rollout split"
timeout 300 python tools/rollout_profile.py 2>&1 | tail -5 | tee $OUT/rollout_profile.txt
