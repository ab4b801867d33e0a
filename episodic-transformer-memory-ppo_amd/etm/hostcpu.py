"""How many host CPUs does this rank really have, and how many threads may it keep busy?  (round 5)

A rank of the trainer keeps host threads BUSY during a rollout: the trainer thread spins on the action flag the sampling kernel
writes (upstream blocks in ``child.recv()``, trainer.py:189), ``copy_threads - 1`` helpers of the observation copier spin between
jobs (csrc/host_copy.hip), and -- with ``worker_processes`` -- every environment process spins on its go word.  On the 256-core
hosts of the MI355X nodes that is free; under a CPU quota (the 1-GPU boxes of this pool run a process under ``cpu.max`` = 16 CPUs)
or a narrow affinity mask it is not: CFS throttles a cgroup that burns more CPU time than its quota for the rest of the period,
which stalls EVERY thread of the rank (measured in round 4: 32 spinning one-environment workers under a 16-CPU quota 102 k -> 54.6 k
env-steps/s).  Eight ranks per node multiply the demand by eight.

``host_cpu_budget()`` reads what the kernel will really grant (affinity mask, cgroup v2 / v1 CPU quota), ``plan_host_threads()``
turns that into the settings the trainer applies at construction, and logs the decision once.
"""
import os
import sys


def _cgroup_quota():
    """CPU quota of this process's cgroup in CPUs (float), or None: cgroup v2 ``cpu.max`` ("max" or "<quota us> <period us>"),
    then cgroup v1 ``cpu.cfs_quota_us / cpu.cfs_period_us``; the tightest quota on the path from the process's cgroup to the root."""
    best = None

    def take(q):
        nonlocal best
        if q is not None and q > 0:
            best = q if best is None else min(best, q)

    try:      # v2: walk up from the process's own cgroup (inside a container the namespace root is usually the only level)
        rel = "/"
        for line in open("/proc/self/cgroup"):
            parts = line.strip().split(":", 2)
            if len(parts) == 3 and parts[0] == "0":
                rel = parts[2]
        path = os.path.normpath("/sys/fs/cgroup/" + rel)
        while path.startswith("/sys/fs/cgroup"):
            f = os.path.join(path, "cpu.max")
            if os.path.exists(f):
                q, per = open(f).read().split()
                if q != "max":
                    take(float(q) / float(per))
            if path == "/sys/fs/cgroup":
                break
            path = os.path.dirname(path)
    except Exception:      # noqa: BLE001 -- no cgroup v2 here
        pass
    try:      # v1
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            take(q / per)
    except Exception:      # noqa: BLE001
        pass
    return best


def host_cpu_budget(local_world=None):
    """{"affinity": CPUs in this process's mask, "cgroup_quota": CPUs or None, "usable": what the rank's NODE share is made of,
    "local_world": ranks sharing it, "per_rank": usable / local_world}.  ``local_world`` defaults to LOCAL_WORLD_SIZE (torchrun sets
    it); without it WORLD_SIZE counts only when the job is known to be one node (NNODES / GROUP_WORLD_SIZE == 1 or unset together
    with a loopback MASTER_ADDR), else 1 -- a multi-node launch without torchrun must not divide the node's CPUs by the global
    world size."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:      # noqa: BLE001
        affinity = os.cpu_count() or 1
    if local_world is None:
        if os.environ.get("LOCAL_WORLD_SIZE"):
            local_world = int(os.environ["LOCAL_WORLD_SIZE"])
        else:
            nnodes = os.environ.get("NNODES") or os.environ.get("GROUP_WORLD_SIZE")
            one_node = (nnodes == "1") if nnodes else os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1")
            local_world = int(os.environ.get("WORLD_SIZE", "1") or 1) if one_node else 1
            if not one_node and int(os.environ.get("WORLD_SIZE", "1") or 1) > 1:
                import sys
                print("[etm] host CPU plan: LOCAL_WORLD_SIZE is not set and the job is not known to be one node: assuming ONE rank on "
                      "this node (set LOCAL_WORLD_SIZE to the ranks per node)", file=sys.stderr, flush=True)
    local_world = max(1, int(local_world))
    quota = _cgroup_quota()
    # conservative: every local rank is assumed to share this mask (NUMA pinning gives the ranks of one node the same mask; ranks
    # pinned to different nodes get a smaller figure than they have -- still tens of CPUs on the MI355X hosts)
    per_rank = affinity / local_world
    if quota is not None:
        per_rank = min(per_rank, quota / local_world)
    usable = min(affinity, quota) if quota is not None else affinity
    return {"affinity": affinity, "cgroup_quota": quota, "usable": usable, "local_world": local_world, "per_rank": per_rank}


_logged = False


def plan_host_threads(copy_threads=1, worker_processes=False, num_envs=1, envs_per_process=1, groups=1, budget=None, quiet=False):
    """Settings that keep the rank's BUSY threads within its CPU share.  Returns a dict:

      copy_threads        threads of the observation copier (>= 1; helpers = copy_threads - 1)
      copier_spin         helpers spin between jobs (True) or sleep on their condition variable at once (False)
      polite_wait         the trainer thread sleeps through most of the expected device time before it spins on the action flag
      envs_per_process    environments per worker process (raised until the processes fit)
      worker_spin         worker processes spin on their go word (True) or back off with short sleeps (False)
      busy_threads        busy threads of this rank during a rollout under the plan
      reason / budget     what was decided on, for the log line and bench.py's config

    With room to spare nothing changes (round 4's settings: they are the fast ones)."""
    global _logged
    b = budget if budget is not None else host_cpu_budget()
    share = b["per_rank"]
    want_ct = max(1, int(copy_threads))
    procs = -(-int(num_envs) // max(1, int(envs_per_process))) if worker_processes else 0
    want_busy = 1 + (want_ct - 1) + procs
    plan = {"copy_threads": want_ct, "copier_spin": True, "polite_wait": False, "envs_per_process": int(envs_per_process),
            "worker_spin": True, "busy_threads": want_busy, "budget": b, "reason": "enough CPUs: unchanged"}
    if share >= want_busy + 0.5:          # half a CPU of slack for the torch / runtime helper threads
        return plan
    whole = max(1, int(share))            # threads that can be busy at once without throttling
    # 1. worker processes: fewer, larger ones (they step their environments one after the other) until they fit next to the trainer
    k = int(envs_per_process)
    if worker_processes:
        room = max(1, whole - 1)
        per_group = max(1, int(num_envs) // max(1, int(groups)))        # a process serves one worker group
        while -(-int(num_envs) // k) > room and k < per_group:
            k += 1
        while per_group % k != 0 and k < per_group:
            k += 1
        procs = -(-int(num_envs) // k)
        plan["envs_per_process"] = k
        plan["worker_spin"] = (1 + procs) <= whole
    # 2. the copier: as many threads as are left, helpers sleep between jobs (a job then costs a condition-variable wake-up, ~10 us)
    left = max(1, whole - procs)
    plan["copy_threads"] = max(1, min(want_ct, left))
    plan["copier_spin"] = False
    # 3. the trainer thread itself: below ~1.5 CPUs even one spinning thread starves the rest of the rank
    plan["polite_wait"] = share < 1.5 + procs
    plan["busy_threads"] = (0 if plan["polite_wait"] else 1) + (procs if plan["worker_spin"] else 0)
    plan["reason"] = (f"{share:.2f} CPUs per rank (affinity {b['affinity']}, cgroup quota {b['cgroup_quota']}, {b['local_world']} local rank(s)) "
                      f"< {want_busy} busy threads wanted")
    if not quiet and not _logged and os.environ.get("ETM_QUIET") != "1":
        _logged = True
        print(f"[etm] host CPU plan: {plan['reason']} -> copy_threads {plan['copy_threads']} (helpers sleep between jobs), "
              f"envs_per_process {plan['envs_per_process']}, worker processes {'spin' if plan['worker_spin'] else 'back off'}, trainer thread "
              f"{'sleeps through the expected device time, then spins' if plan['polite_wait'] else 'spins on the action flag'}",
              file=sys.stderr, flush=True)
    return plan


def set_timer_slack_ns(ns=1000):
    """Short sleeps of this thread wake up on time (default timer slack: 50 us): used by the polite flag wait."""
    try:
        import ctypes
        ctypes.CDLL(None, use_errno=True).prctl(29, int(ns), 0, 0, 0)      # PR_SET_TIMERSLACK
        return True
    except Exception:      # noqa: BLE001
        return False
