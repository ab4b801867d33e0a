"""Environment workers as PROCESSES over one shared-memory segment (round 4; upstream trainer.py:189-218, worker.py:20-48).

Upstream sends ``("step", action)`` down a pipe to every worker process and then collects the results: the environments step
concurrently, off the trainer thread.  This front-end keeps that structure -- one process per ``envs_per_proc`` environments --
but replaces the pipes of the per-step path by a shared segment that the trainer registers with the HIP runtime
(``etm_host_register``), so that

* the DEVICE hands the actions to the workers itself: the sampling kernel of a rollout step stores the actions and then the
  step's sequence number into the segment (``act`` / ``go``); the workers spin on ``go`` -- no trainer-thread work between the
  kernel and ``env.step``;
* the workers write observation rows, rewards, done flags and episode results straight into the segment (``obs`` is the
  staging buffer the observation upload DMA reads) and publish ``ready[p] = sequence number``;
* the trainer's native rollout driver (csrc/rollout_driver.hip, ``etm_rollout_drive``) only waits for the ``ready`` words, does
  the episode bookkeeping and enqueues the upload + the step graph -- no Python on the per-step path at all.

The same workers serve the host-driven protocol (``step(actions)`` from Python: eager rollouts, tools): then the front-end
writes ``act`` and ``go`` itself.  A worker that sees nothing for ``IDLE_PARK_S`` parks on its command pipe (no spinning
between rollouts unless the trainer holds it active).

Segment layout: see ``ShmLayout``.  Sequence protocol per worker group g: ``go[g]`` changes to a NEW non-zero value n  ->  the
group's workers step once and set ``ready[p] = n``;  ``go[g] = 0``  ->  "sequence restart": workers answer ``ready[p] = 0``.
While a worker writes the observation rows of step n it publishes its progress in ``rows[p] = (n << 16) | rows final so far``: the
native driver starts the upload of a row as soon as it is final (the copy engine then runs under the remaining row writes).

This module imports numpy only (the worker processes never load torch or the kernel library).
"""
import json
import os
import subprocess
import sys
import time
from multiprocessing import shared_memory

import numpy as np

LINE = 8                      # int64 words per cache line: every control word sits on its own 64-byte line
IDLE_PARK_S = 0.05          # a worker that sees no go word for this long parks on its pipe (the optimisation phase between two rollouts: 76 ms)
ST_PARKED, ST_ACTIVE, ST_DEAD = 0, 1, 2


def _align(n, a=4096):
    return (n + a - 1) // a * a


class ShmLayout:
    """Offsets of the fields of the shared segment (identical in the trainer and in every worker: pure arithmetic)."""

    def __init__(self, W, obs_shape, B, S, G, P):
        self.W, self.obs_shape, self.B, self.S, self.G, self.P = int(W), tuple(int(x) for x in obs_shape), int(B), int(S), int(G), int(P)
        row = int(np.prod(self.obs_shape))
        fields = [("obs", np.float32, (W,) + self.obs_shape), ("act", np.int64, (W, B)), ("go", np.int64, (G, LINE)),
                  ("ready", np.int64, (P, LINE)), ("rows", np.int64, (P, LINE)), ("state", np.int64, (P, LINE)), ("err", np.int64, (P, LINE)),
                  ("ctl", np.int64, (LINE,)),                                    # [0] abort, [1] hold (no self-parking), [2] activation epoch, [3] back off (sleep between polls)
                  #                                                                (state[p]: [0] ST_*, [1] the last activation epoch process p has SEEN while active)
                  ("rewards", np.float32, (S, W)), ("dones", np.uint8, (S, W)), ("info_reward", np.float64, (S, W)),
                  ("info_length", np.int64, (S, W)), ("info_success", np.int8, (S, W)),
                  ("last_rewards", np.float32, (W,)), ("last_dones", np.uint8, (W,)),
                  ("trace", np.float64, (S, P, 2))]      # per step and process: CLOCK_MONOTONIC when go was seen / when ready was set
        self.fields, off = {}, 0
        for name, dt, shape in fields:
            self.fields[name] = (off, np.dtype(dt), tuple(int(x) for x in shape))
            off = _align(off + int(np.prod(shape)) * np.dtype(dt).itemsize)
        self.nbytes = off
        self.row_floats = row

    def views(self, buf):
        return {name: np.ndarray(shape, dtype=dt, buffer=buf, offset=off) for name, (off, dt, shape) in self.fields.items()}

    def spec(self):
        return dict(W=self.W, obs_shape=list(self.obs_shape), B=self.B, S=self.S, G=self.G, P=self.P)


# --------------------------------------------------------------------------------------------- worker process
def _make_part(env_config, n, first_worker_id):
    """The environments of one worker process as a small in-process vector environment."""
    if env_config["type"] == "Synthetic":
        from environments.synthetic import SyntheticVecEnv
        # (no gen_threads: the worker processes ARE the parallelism; a drawing pool per process would only add spinning threads)
        keys = ("obs_shape", "num_actions", "max_episode_steps", "seed", "p_reward", "p_done", "pool", "step_cost_us")
        kw = {k: env_config[k] for k in keys if k in env_config}
        if "obs_shape" in kw:
            kw["obs_shape"] = tuple(kw["obs_shape"])
        return SyntheticVecEnv(n, first_worker_id=first_worker_id, copy_threads=1, row_chunks=n, min_chunked_envs=1, **kw)
    from environments.vec_env import SerialVecEnv
    from utils import create_env
    env = SerialVecEnv([create_env(env_config, worker_id=first_worker_id + w) for w in range(n)])
    env.ROW_CHUNKS = n                 # on_rows after every environment
    return env


def worker_main(argv):
    """``python -m environments.shm_env <json>``: attach, build the environments, serve until the pipe closes."""
    a = json.loads(argv[0])
    lay = ShmLayout(**a["layout"])
    shm = shared_memory.SharedMemory(name=a["shm"])
    try:
        # the segment belongs to the trainer: keep this process's resource tracker from unlinking it at exit
        from multiprocessing import resource_tracker
        resource_tracker.unregister(shm._name, "shared_memory")
    except Exception:
        pass
    v = lay.views(shm.buf)
    p, g, lo, hi = a["proc"], a["group"], a["lo"], a["hi"]
    env = _make_part(a["env"], hi - lo, a["first_worker_id"] + lo)
    obs_rows, act_rows = v["obs"][lo:hi], v["act"][lo:hi]
    go, ready, state, err, ctl = v["go"][g], v["ready"][p], v["state"][p], v["err"][p], v["ctl"]
    S = lay.S
    single = lay.B == 1
    stdin = sys.stdin.buffer
    out = sys.stdout.buffer

    def reply(msg):
        out.write(msg + b"\n")
        out.flush()

    trace = v["trace"]
    rows_word = v["rows"][p]
    tag = [0]

    def rows_final(a, b):
        rows_word[0] = tag[0] | b             # rows [0, b) of this process are final for the step in the upper bits

    def step_once(n):
        t_seen = time.perf_counter()
        acts = act_rows[:, 0] if single else act_rows
        tag[0] = n << 16
        _, r, d, infos = env.step(acts, out=obs_rows, on_rows=rows_final)
        t = (n - 1) % S                                   # row of the per-step result arrays (rollout: n = t + 1)
        v["rewards"][t, lo:hi] = r
        v["dones"][t, lo:hi] = d
        v["last_rewards"][lo:hi] = r
        v["last_dones"][lo:hi] = d
        if d.any():
            for w in np.flatnonzero(d):
                info = infos[w] or {}
                v["info_reward"][t, lo + w] = float(info.get("reward", 0.0))
                v["info_length"][t, lo + w] = int(info.get("length", 0))
                v["info_success"][t, lo + w] = int(bool(info["success"])) if "success" in info else -1
        trace[t, p, 0], trace[t, p, 1] = t_seen, time.perf_counter()
        ready[0] = n                                      # (x86: the stores above are visible before this one)

    last = int(go[0])
    parent = os.getppid()
    try:
        while True:
            # ---- parked: block on the command pipe
            state[0] = ST_PARKED
            if int(go[0]) != last and ctl[0] == 0:        # a go word slipped in between the last poll and parking: serve it
                state[0] = ST_ACTIVE
            else:
                line = stdin.readline()
                if not line:
                    break
                cmd = line.strip()
                if cmd == b"close":
                    break
                if cmd == b"reset":
                    env.reset(out=obs_rows)
                    last = int(go[0])
                    ready[0] = last
                    reply(b"ok")
                    continue
                if cmd != b"wake":
                    continue
                state[0] = ST_ACTIVE
            # ---- active: spin on the group's go word.  state[1] = the activation epoch this process has seen: the trainer's
            # activate() stores hold (ctl[1]) BEFORE the epoch (ctl[2]), so a process that has acknowledged the current epoch reads the
            # current hold flag in every later parking decision (x86: stores are seen in program order, loads are not reordered)
            state[1] = int(ctl[2])
            idle_since = time.perf_counter()
            spins = 0
            backoff = ctl[3] != 0
            while True:
                n = int(go[0])
                if n != last:
                    last = n
                    if n == 0:
                        ready[0] = 0
                    else:
                        step_once(n)
                    idle_since = time.perf_counter()
                    spins = 0
                    continue
                spins += 1
                if backoff and spins & 63 == 0:
                    time.sleep(20e-6)             # the rank's CPU share does not cover spinning workers (etm/hostcpu.py)
                if spins & 1023 == 0:
                    backoff = ctl[3] != 0
                    if ctl[0] != 0:
                        break
                    if state[1] != ctl[2]:
                        state[1] = int(ctl[2])
                        idle_since = time.perf_counter()        # a new activation: the idle clock starts again
                    if spins & 0xfffff == 0 and os.getppid() != parent:      # the trainer is gone (killed mid-rollout): do not spin on
                        return
                    if ctl[1] == 0 and time.perf_counter() - idle_since > IDLE_PARK_S:
                        break
            if ctl[0] != 0:
                state[0] = ST_PARKED
                # abort: wait for commands (close) on the pipe
    except Exception:      # noqa: BLE001 -- report through the segment; the trainer raises with the traceback from stderr
        import traceback
        traceback.print_exc()
        err[0] = 1
    finally:
        state[0] = ST_DEAD
        try:
            env.close()
        except Exception:
            pass
        shm.close()


# --------------------------------------------------------------------------------------------- trainer-side front-end
def _probe_env(env_config):
    """(observation shape, number of actions, max_episode_steps) of the configured environment."""
    if env_config["type"] == "Synthetic":
        from environments.synthetic import SyntheticEnv
        keys = ("obs_shape", "num_actions", "max_episode_steps")
        kw = {k: env_config[k] for k in keys if k in env_config}
        if "obs_shape" in kw:
            kw["obs_shape"] = tuple(kw["obs_shape"])
        e = SyntheticEnv(pool=1, **kw)
    else:
        from utils import create_env
        e = create_env(env_config)
    if not hasattr(e.action_space, "n"):
        e.close()
        raise NotImplementedError("worker_processes supports single-branch (Discrete) action spaces only: the shared segment carries one "
                                  "action word per environment and the trainer's device-side hand-over writes one; use the in-process "
                                  "environments (worker_processes: false) for a MultiDiscrete space")
    res = tuple(e.observation_space.shape), int(e.action_space.n), int(e.max_episode_steps)
    e.close()
    return res


class _ShmGroup:
    """One worker group of a ``ShmVecEnv`` behind the VecEnv protocol (host-driven stepping of that group only)."""

    def __init__(self, parent, g, lo, hi):
        self.parent, self.g, self.lo, self.hi = parent, g, lo, hi
        self.num_envs = hi - lo
        self.observation_space_shape = parent.observation_space_shape
        self.num_actions, self.max_episode_steps = parent.num_actions, parent.max_episode_steps

    def step(self, actions, out=None, on_rows=None):
        return self.parent._step_groups([self.g], actions, out, on_rows, self.lo)

    def reset(self, out=None):
        raise RuntimeError("reset the whole ShmVecEnv, not one of its groups")

    def close(self):
        return None


class ShmVecEnv:
    """``num_envs`` environments in ``num_envs / envs_per_proc`` worker processes, ``groups`` worker groups (see module docstring).

    ``steps_per_rollout``: rows of the per-step result arrays (the trainer's ``worker_steps``)."""

    def __init__(self, env_config: dict, num_envs: int, first_worker_id: int = 0, groups: int = 1, envs_per_proc: int = 1,
                 steps_per_rollout: int = 1, num_branches: int = 1, spin: bool = True):
        shape, n_act, T = _probe_env(env_config)
        self.observation_space_shape, self.num_actions, self.max_episode_steps = tuple(shape), int(n_act), int(T)
        W, G = int(num_envs), int(groups)
        if W % G != 0:
            raise ValueError("num_envs must be a multiple of groups")
        per_group = W // G
        k = max(1, min(int(envs_per_proc), per_group))
        while per_group % k != 0:
            k -= 1
        self.num_envs, self.groups, self.envs_per_proc = W, G, k
        self.procs_per_group = per_group // k
        P = G * self.procs_per_group
        self.layout = ShmLayout(W, self.observation_space_shape, num_branches, steps_per_rollout, G, P)
        self.shm = shared_memory.SharedMemory(create=True, size=self.layout.nbytes)
        self.v = self.layout.views(self.shm.buf)
        for a in self.v.values():
            a[...] = 0
        self.v["ctl"][3] = 0 if spin else 1        # workers sleep 20 us every 64 polls instead of spinning flat out
        self.bounds = [(g * per_group, (g + 1) * per_group) for g in range(G)]
        self.parts = [_ShmGroup(self, g, lo, hi) for g, (lo, hi) in enumerate(self.bounds)]
        self.proc_group = [p // self.procs_per_group for p in range(P)]
        self._seq = [0] * G
        self._procs = []
        pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ)
        env["PYTHONPATH"] = pkg + os.pathsep + env.get("PYTHONPATH", "")
        for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
            env[var] = "1"
        for p in range(P):
            g = self.proc_group[p]
            lo = self.bounds[g][0] + (p % self.procs_per_group) * k
            arg = json.dumps(dict(shm=self.shm.name, layout=self.layout.spec(), proc=p, group=g, lo=lo, hi=lo + k, env=env_config,
                                  first_worker_id=first_worker_id))
            self._procs.append(subprocess.Popen([sys.executable, "-m", "environments.shm_env", arg], stdin=subprocess.PIPE,
                                                stdout=subprocess.PIPE, env=env, cwd=pkg))
        self._closed = False

    # ---- control
    def _check(self):
        if self.v["err"][:, 0].any() or any(pr.poll() is not None for pr in self._procs):
            bad = [p for p, pr in enumerate(self._procs) if pr.poll() is not None or self.v["err"][p, 0]]
            raise RuntimeError(f"environment worker process(es) {bad} failed (traceback on stderr above)")

    def activate(self, hold=True):
        """Make every worker spin on its group's go word (``hold``: until ``park()``; else they park themselves when idle)."""
        ctl = self.v["ctl"]
        ctl[1] = 1 if hold else 0
        epoch = int(ctl[2]) + 1
        ctl[2] = epoch                      # (stored AFTER hold: a worker that acknowledges this epoch has the hold flag too)
        state, seen = self.v["state"][:, 0], self.v["state"][:, 1]
        # A worker counts as awake when it is ACTIVE and has acknowledged THIS epoch -- a worker that read "not held, idle" just
        # before the stores above still shows ACTIVE for a moment, then parks and blocks on its pipe: it never acknowledges, shows
        # PARKED on a later pass of this loop and is woken then (one pass over the states missed exactly that worker).
        t0 = time.perf_counter()
        last_wake = [0.0] * len(self._procs)
        while True:
            pending = [p for p in range(len(self._procs)) if state[p] != ST_ACTIVE or seen[p] != epoch]
            if not pending:
                return self
            now = time.perf_counter()
            for p in pending:
                if state[p] == ST_PARKED and now - last_wake[p] > 0.002:
                    self._wake(p)
                    last_wake[p] = now
            if now - t0 > 30.0:
                self._check()
                raise RuntimeError("environment workers did not wake up within 30 s")

    def _wake(self, p):
        """One "wake" line to process p (a surplus line is consumed at its next parking and costs one idle period of spinning)."""
        try:
            self._procs[p].stdin.write(b"wake\n")
            self._procs[p].stdin.flush()
        except (BrokenPipeError, OSError):
            self._check()
            raise

    def park(self):
        """Let the workers go back to blocking on their pipes (they park after IDLE_PARK_S without work)."""
        self.v["ctl"][1] = 0

    def restart_sequence(self):
        """go = 0 on every group and wait for the acknowledgement: the next go values count from 1 (the device's step counter)."""
        self.activate(hold=bool(self.v["ctl"][1]))
        self.v["go"][:, 0] = 0
        ready = self.v["ready"][:, 0]
        t0 = time.perf_counter()
        while (ready != 0).any():
            if time.perf_counter() - t0 > 30.0:
                self._check()
                raise RuntimeError("environment workers did not acknowledge the sequence restart within 30 s")
        self._seq = [0] * self.groups

    # ---- VecEnv protocol (host-driven)
    def reset(self, out=None):
        import select
        # Workers read their pipe only when PARKED, and a held worker (ctl[1] = 1) never parks: drop the hold for the duration of
        # the reset (active workers then park after IDLE_PARK_S and take the command), restore it afterwards.  Every reply is
        # awaited with a deadline -- a dead worker must raise, not hang the trainer.
        held = bool(self.v["ctl"][1])
        self.v["ctl"][1] = 0
        for pr in self._procs:
            pr.stdin.write(b"reset\n")
            pr.stdin.flush()
        deadline = time.perf_counter() + 30.0
        for p, pr in enumerate(self._procs):
            while True:
                left = deadline - time.perf_counter()
                ready, _, _ = select.select([pr.stdout], [], [], max(0.0, min(left, 0.5)))
                if ready:
                    break
                self._check()
                if left <= 0:
                    raise RuntimeError(f"environment worker process {p} did not answer the reset within 30 s")
            line = pr.stdout.readline()
            if line.strip() != b"ok":
                self._check()
                raise RuntimeError("environment worker did not answer the reset")
        self._seq = [int(x) for x in self.v["go"][:, 0]]
        if held:
            self.activate(hold=True)
        if out is not None and out.ctypes.data != self.v["obs"].ctypes.data:
            np.copyto(out, self.v["obs"])
            return out
        return self.v["obs"]

    def _step_groups(self, gs, actions, out, on_rows, row0):
        v = self.v
        lo, hi = self.bounds[gs[0]][0], self.bounds[gs[-1]][1]
        acts = np.asarray(actions).reshape(hi - lo, -1)
        if acts.ctypes.data != v["act"][lo:hi].ctypes.data:
            v["act"][lo:hi] = acts
        state = v["state"][:, 0]
        for g in gs:
            self._seq[g] += 1
            v["go"][g, 0] = self._seq[g]
        procs = [p for p in range(len(self._procs)) if self.proc_group[p] in gs]
        last_wake = {}
        for p in procs:                                    # a parked worker misses the go word: wake it (it re-reads go first)
            if state[p] != ST_ACTIVE:
                self._wake(p)
                last_wake[p] = time.perf_counter()
        ready = v["ready"][:, 0]
        t0 = time.perf_counter()
        spins = 0
        while any(ready[p] != self._seq[self.proc_group[p]] for p in procs):
            spins += 1
            if spins % 256 == 0:
                # "store go, load state" here against "store PARKED, load go" in the worker is a Dekker pair without fences: both
                # sides can read the old value.  So the states are read AGAIN while waiting and a worker that shows PARKED with its
                # step outstanding gets (another) wake -- no missed wake-up can outlive a few milliseconds.
                now = time.perf_counter()
                for p in procs:
                    if ready[p] != self._seq[self.proc_group[p]] and state[p] == ST_PARKED and now - last_wake.get(p, 0.0) > 0.002:
                        self._wake(p)
                        last_wake[p] = now
                if spins % 4096 == 0:
                    self._check()
                    if now - t0 > 60.0:
                        raise RuntimeError("environment workers did not finish a step within 60 s")
        obs = v["obs"][lo:hi]
        if out is not None and out.ctypes.data != obs.ctypes.data:
            np.copyto(out, obs)
            obs = out
        if on_rows is not None:
            on_rows(0, hi - lo)
        dones = v["last_dones"][lo:hi].astype(bool)
        infos = [None] * (hi - lo)
        if dones.any():
            for w in np.flatnonzero(dones):
                g = (lo + w) // (self.num_envs // self.groups)
                t = (self._seq[g] - 1) % self.layout.S
                infos[w] = self.info_at(t, lo + w)
        return obs, v["last_rewards"][lo:hi].copy(), dones, infos

    def info_at(self, t, w):
        """Episode result that worker ``w`` recorded at result row ``t`` (upstream's info dict: reward, length[, success])."""
        v = self.v
        info = {"reward": float(v["info_reward"][t, w]), "length": int(v["info_length"][t, w])}
        if v["info_success"][t, w] >= 0:
            info["success"] = bool(v["info_success"][t, w])
        return info

    def step(self, actions, out=None, on_rows=None):
        return self._step_groups(list(range(self.groups)), actions, out, on_rows, 0)

    def close(self):
        if self._closed:
            return
        self._closed = True
        try:
            self.v["ctl"][0] = 1
            for pr in self._procs:
                try:
                    pr.stdin.write(b"close\n")
                    pr.stdin.flush()
                    pr.stdin.close()
                except Exception:
                    pass
            for pr in self._procs:
                try:
                    pr.wait(timeout=5)
                except Exception:
                    pr.kill()
        finally:
            self.v = None
            try:
                self.shm.unlink()
            except Exception:
                pass
            try:
                self.shm.close()          # (raises while the trainer still holds views of the segment: the mapping then goes with them)
            except Exception:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


if __name__ == "__main__":
    worker_main(sys.argv[1:])
