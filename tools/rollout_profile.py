"""Where does a rollout step go?  Host-side timers around graph replay / sync / env step (BASELINE config 3)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the HIP runtime starts: rollout_groups "auto" = 4 (trainer.py)
os.environ.setdefault("ETM_HW_QUEUES_SET_EARLY", "1")   # marker for trainer.py: the line above ran before the HIP runtime started
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import numpy as np, torch
from etm import lib as _etm_lib
if os.environ.get("ETM_DIAG_LIB"):
    _etm_lib.LIB_PATH = os.environ["ETM_DIAG_LIB"]      # A/B against another build of the library
from yaml_parser import YamlParser
from trainer import PPOTrainer

cfg = YamlParser(os.path.join(REPO, "episodic-transformer-memory-ppo_amd", "configs",
                              os.environ.get("ETM_PROFILE_CONFIG", "synthetic_minigrid") + ".yaml")).get_config()   # BASELINE config 3 by default
for k in sys.argv[1:]:
    name, val = k.split("=")
    tgt = cfg
    *path, leaf = name.split(".")
    for part in path:
        tgt = tgt[part]
    tgt[leaf] = (val == "1") if val in ("0", "1") else (int(val) if val.lstrip("-").isdigit() else val)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr = PPOTrainer(cfg, run_id="prof", device=dev, tensorboard=False)
tr._sample_training_data()          # warm-up: MIOpen find, graph capture
tr.buffer.prepare_batch_dict()
torch.cuda.synchronize()

S = cfg["worker_steps"]
for rep in range(2):
    tr._chain_log = [] if rep == 1 else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr._sample_training_data()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lt = tr.last_update_timing
    print(f"rollout: {dt * 1e3:.1f} ms = {dt / S * 1e6:.0f} us per step (host: env.step {lt['env_s'] / S * 1e6:.0f}, waiting for actions "
          f"{lt['wait_s'] / S * 1e6:.0f}, upload + launch {lt['launch_s'] / S * 1e6:.0f} us per step)  stream_observations={tr._stream_obs} "
          f"groups={len(tr._groups)}")

if tr._chain_log:
    import numpy as _np
    c = _np.array(tr._chain_log[8:-2])          # (wait start, flag seen, env.step + bookkeeping done, launch done)
    cyc = _np.diff(c[:, 1])
    print(f"first group, per step: flag seen -> env.step done {(c[:, 2] - c[:, 1]).mean() * 1e6:.1f} us, -> launch done "
          f"{(c[:, 3] - c[:, 2]).mean() * 1e6:.1f} us, launch done -> next flag seen {(c[1:, 1] - c[:-1, 3]).mean() * 1e6:.1f} us "
          f"(of which spinning {(c[1:, 1] - c[1:, 0]).mean() * 1e6:.1f} us); cycle {cyc.mean() * 1e6:.1f} us")
    if getattr(tr, "_native_rollout", False):
        # worker processes + native driver: the workers' own clocks (CLOCK_MONOTONIC, like the driver's) per step
        trw = tr._shm_env.v["trace"][8:S - 2]                     # [steps, processes, (go seen, ready set)]
        ppg = tr._shm_env.procs_per_group
        g0 = trw[:, :ppg]                                          # processes of the first group
        seen, done = g0[:, :, 0].min(axis=1), g0[:, :, 1].max(axis=1)
        cc = _np.array(tr._chain_log[8:S - 2])
        k = min(len(cc), len(seen))
        print(f"first group, native driver: go seen by a worker -> last worker done {(done - seen).mean() * 1e6:.1f} us (one worker's step "
              f"{(g0[:, :, 1] - g0[:, :, 0]).mean() * 1e6:.1f} us), -> driver saw ready {(cc[:k, 1] - done[:k]).mean() * 1e6:.1f} us, -> "
              f"upload + graph enqueued {(cc[:k, 3] - cc[:k, 1]).mean() * 1e6:.1f} us, enqueued -> next go seen "
              f"{(seen[1:k] - cc[:k - 1, 3]).mean() * 1e6:.1f} us; trainer-thread work per group and step "
              f"{tr._drive_timing[1] / S / len(tr._groups) * 1e6:.1f} us, waiting {tr._drive_timing[0] / S / len(tr._groups) * 1e6:.1f} us")
tr._chain_log = None

# device time of the step graphs alone (no host work in between): head = critical path of a step, tail = under env.step
if tr._step_graph is not None:
    g0 = tr._groups[0]                       # the first worker group's graphs (all workers when rollout_groups = 1)
    g0.t_dev.zero_(); torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    n = max(1, min(200, S - 8))          # the step counter must stay inside the staging arrays
    e0.record()
    for _ in range(n):
        tr._step_graph[0].replay()
    e1.record()
    for _ in range(n if tr._step_graph[1] is not None else 0):
        tr._step_graph[1].replay()
    e2.record(); torch.cuda.synchronize()
    print(f"step graphs of one group ({g0.W} workers) back-to-back: head {e0.elapsed_time(e1) / n * 1e3:.1f} us, "
          f"tail {e1.elapsed_time(e2) / n * 1e3:.1f} us  (host_flag={tr._host_flag})")
    # do the step graphs of DIFFERENT groups overlap on the device?  every group replays n times on its own stream, no host waits
    if len(tr._groups) > 1 and all(g.graphs is not None and g.stream is not None for g in tr._groups):
        for g in tr._groups:
            g.t_dev.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            for g in tr._groups:
                with torch.cuda.stream(g.stream):
                    g.graphs[0].replay()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{len(tr._groups)} groups replaying concurrently on their streams: {dt / n * 1e6:.1f} us per round of {len(tr._groups)} graphs "
              f"(host launch time {t_host / n * 1e6:.1f} us per round)")
