#!/bin/bash
# rocprofv3 PMC passes of the kernel micro-benchmarks (tools/kernel_rooflines.py) and of the encoder kernels
# (tools/conv_time.py); run on the MI355X box.  Counters are collected in separate passes with --kernel-trace only (no other
# tracing domain), as /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2).
#   bash tools/run_pmc.sh [targets...]          default targets: window_cold window_train window_sorted window_cold128 mfma3 gae ppo encoder rollout_step
# Output: gpurun_out/pmc_r03/<target>/<pass>/counters.csv  ->  python tools/pmc_summarize.py gpurun_out/pmc_r03 profiles/r03_pmc_summary.json
set -u
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_${ROUND:-r03}
TARGETS=${@:-window_cold window_train window_sorted window_cold128 mfma3 gae ppo encoder rollout_step}
for target in $TARGETS; do
  # rollout_step5 / rollout_step2: the rollout_step target on BASELINE config 5 / 2 (group form of the step kernel); rollout_step5pw: per-worker form
  if [ $target = conv ]; then cmd="python $ROOT/tools/conv_time.py";
  elif [ $target = rollout_step5 ]; then cmd="env ETM_PROFILE_CONFIG=synthetic_mortar_gtrxl python $ROOT/tools/kernel_rooflines.py rollout_step 6";
  elif [ $target = rollout_step5pw ]; then cmd="env ETM_PROFILE_CONFIG=synthetic_mortar_gtrxl python $ROOT/tools/kernel_rooflines.py rollout_step 6 rollout_group_kernel=0";
  elif [ $target = rollout_step2 ]; then cmd="env ETM_PROFILE_CONFIG=synthetic_cartpole python $ROOT/tools/kernel_rooflines.py rollout_step 6";
  elif [ $target = bench_dense ]; then cmd="python $ROOT/bench.py --attention dense --steps 1 --warmup 2 --no-cpu-baseline --no-profile --no-rooflines";
  else cmd="python $ROOT/tools/kernel_rooflines.py $target 6"; fi
  # bench_dense (round 5): the dense MFMA attention kernels INSIDE the whole path (rollout + captured optimisation steps), counters of
  # those kernels only -- the rest of the process runs uninstrumented
  filter=""; if [ $target = bench_dense ]; then filter="--kernel-include-regex mha_fwd_kernel|bwd_dw_kernel|bwd_scores_kernel|bwd_uw_kernel|bwd_dx_kernel"; fi
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    if [ $target = bench_dense ] && [ "$pass" = "WRITE_SIZE" ]; then continue; fi
    tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
    rm -rf /tmp/pmc_run
    timeout 400 rocprofv3 --pmc $pass $filter --kernel-trace --output-format csv -d /tmp/pmc_run -o pmc -- $cmd > /tmp/pmc_run.log 2>&1
    mkdir -p $OUT/$target/$tag
    for f in $(find /tmp/pmc_run -name "*counter_collection.csv"); do
      # keep the rows of this build's kernels only (the csv of a whole process is tens of MB)
      python - "$f" "$OUT/$target/$tag/counters.csv" <<'PY'
import csv, sys
src, dst = sys.argv[1], sys.argv[2]
with open(src) as f, open(dst, "w", newline="") as g:
    r = csv.DictReader(f)
    w = csv.DictWriter(g, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value"])
    w.writeheader()
    for row in r:
        n = row["Kernel_Name"]
        if "anonymous namespace" in n and "at::native" not in n and " ck::" not in n:
            w.writerow({k: row[k] for k in ("Kernel_Name", "Counter_Name", "Counter_Value")})
PY
    done
    echo "$target / $tag: $(wc -l < $OUT/$target/$tag/counters.csv 2>/dev/null) rows"
  done
done
