"""Host side of a rank's rollout loop WITHOUT a device: the in-process synthetic environment (observation rows through the kernel
library's multi-threaded copier) stepped `steps` times with a simulated wait for the device between steps, under the host CPU plan
of etm/hostcpu.py.  Prints one JSON line: microseconds of host work per step (the simulated device time excluded).

    python tools/host_plumbing_probe.py <cpus e.g. 0-3> <local ranks sharing them> <steps> [plan|noplan] [start-at unix time]

Used by tests/test_host_logic.py: two ranks on four CPUs must not slow each other down when the plan is applied."""
import json
import os
import sys
import time

lo, hi = (int(x) for x in sys.argv[1].split("-"))
os.sched_setaffinity(0, set(range(lo, hi + 1)))
ranks, steps = int(sys.argv[2]), int(sys.argv[3])
use_plan = (sys.argv[4] if len(sys.argv) > 4 else "plan") == "plan"
start_at = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import numpy as np  # noqa: E402

from environments import synthetic  # noqa: E402
from etm import hostcpu  # noqa: E402

W, want_ct = 32, 4
budget = hostcpu.host_cpu_budget(local_world=ranks)
plan = hostcpu.plan_host_threads(copy_threads=want_ct, num_envs=W, budget=budget, quiet=True)
ct = plan["copy_threads"] if use_plan else want_ct
if use_plan and not plan["copier_spin"]:
    synthetic.set_copier_spin(False)
env = synthetic.SyntheticVecEnv(W, obs_shape=(3, 84, 84), num_actions=3, max_episode_steps=96, seed=0, pool=8, copy_threads=ct)
out = np.empty((W, 3, 84, 84), dtype=np.float32)
env.reset(out=out)
acts = np.zeros(W, dtype=np.int64)
sink = []
while time.time() < start_at:
    pass
busy = 0.0
for t in range(steps):
    t0 = time.perf_counter()
    env.step(acts, out=out, on_rows=lambda a, b: sink.append(b))
    busy += time.perf_counter() - t0
    sink.clear()
    # the device's share of a step: the trainer thread waits for the action flag (spinning, or -- polite plan -- mostly asleep)
    t_dev = time.perf_counter() + 100e-6
    if use_plan and plan["polite_wait"]:
        time.sleep(70e-6)
    while time.perf_counter() < t_dev:
        pass
print(json.dumps({"us_per_step": 1e6 * busy / steps, "copy_threads": ct, "plan": plan["reason"], "per_rank": budget["per_rank"]}))
