"""Vectorised environment front-ends for the trainer.

``VecEnv`` protocol (what ``PPOTrainer`` steps):
    num_envs, observation_space_shape, num_actions, max_episode_steps
    reset(out=None) -> obs [W, *obs_shape] float32
    step(actions [W] or [W, B], out=None, on_rows=None) -> (obs, rewards [W] f32, dones [W] bool, infos [W] (dict or None))
        on_rows(lo, hi), optional: called as soon as observation rows [lo, hi) of ``out`` are final, in increasing order and
        covering [0, W) -- the trainer starts their upload while the remaining environments still step
with auto-reset: for a finished worker the returned observation is already the first one of the next episode, which
is exactly what upstream's loop does by hand (trainer.py:195-201).

* ``SerialVecEnv``  -- wraps in-process single envs with the upstream env API.
* ``PipeVecEnv``    -- wraps upstream-protocol ``Worker`` subprocesses (worker.py), one pipe round trip per step.
* ``environments.synthetic.SyntheticVecEnv`` implements the protocol natively.
"""
import numpy as np


class _VecBase:
    ROW_CHUNKS = 2   # on_rows granularity: W / ROW_CHUNKS workers per notification (each notification costs the trainer ~9 us)

    def _alloc(self, out):
        return out if out is not None else np.zeros((self.num_envs,) + self.observation_space_shape, dtype=np.float32)

    def _notify(self, on_rows, done_upto, sent_upto):
        """Call on_rows for the complete chunks in [sent_upto, done_upto); returns the new sent_upto."""
        if on_rows is None:
            return sent_upto
        step = max(1, -(-self.num_envs // self.ROW_CHUNKS))
        while sent_upto + step <= done_upto or (done_upto == self.num_envs and sent_upto < done_upto):
            hi = min(sent_upto + step, self.num_envs)
            on_rows(sent_upto, hi)
            sent_upto = hi
        return sent_upto


class SerialVecEnv(_VecBase):
    def __init__(self, envs):
        self.envs = list(envs)
        self.num_envs = len(self.envs)
        e = self.envs[0]
        self.observation_space_shape = tuple(e.observation_space.shape)
        self.num_actions = int(e.action_space.n)
        self.max_episode_steps = int(e.max_episode_steps)

    def reset(self, out=None):
        out = self._alloc(out)
        for w, e in enumerate(self.envs):
            out[w] = e.reset()
        return out

    def step(self, actions, out=None, on_rows=None):
        out = self._alloc(out)
        actions = np.asarray(actions).reshape(self.num_envs, -1)
        rewards = np.zeros(self.num_envs, dtype=np.float32)
        dones = np.zeros(self.num_envs, dtype=bool)
        infos = [None] * self.num_envs
        sent = 0
        for w, e in enumerate(self.envs):
            obs, rewards[w], dones[w], info = e.step(actions[w])
            if info:
                infos[w] = info
                obs = e.reset()
            out[w] = obs
            sent = self._notify(on_rows, w + 1, sent)
        return out, rewards, dones, infos

    def close(self):
        for e in self.envs:
            e.close()


class PipeVecEnv(_VecBase):
    """One subprocess per env, upstream (cmd, data) pipe protocol; actions are sent to all workers before any
    result is received so the environments step concurrently."""

    def __init__(self, env_config: dict, num_envs: int, first_worker_id: int = 0):
        from utils import create_env
        from worker import Worker
        probe = create_env(env_config)
        self.observation_space_shape = tuple(probe.observation_space.shape)
        self.num_actions = int(probe.action_space.n)
        self.max_episode_steps = int(probe.max_episode_steps)
        probe.close()
        self.num_envs = num_envs
        self.workers = [Worker(env_config, worker_id=first_worker_id + w) for w in range(num_envs)]

    def reset(self, out=None):
        out = self._alloc(out)
        for wk in self.workers:
            wk.child.send(("reset", None))
        for w, wk in enumerate(self.workers):
            out[w] = self._recv(wk)
        return out

    def step(self, actions, out=None, on_rows=None):
        out = self._alloc(out)
        actions = np.asarray(actions).reshape(self.num_envs, -1)
        for w, wk in enumerate(self.workers):
            wk.child.send(("step", actions[w]))
        rewards = np.zeros(self.num_envs, dtype=np.float32)
        dones = np.zeros(self.num_envs, dtype=bool)
        infos = [None] * self.num_envs
        sent = 0
        for w, wk in enumerate(self.workers):
            obs, rewards[w], dones[w], info = self._recv(wk)
            if info:
                infos[w] = info
                wk.child.send(("reset", None))
                obs = self._recv(wk)
            out[w] = obs
            sent = self._notify(on_rows, w + 1, sent)
        return out, rewards, dones, infos

    @staticmethod
    def _recv(wk):
        msg = wk.child.recv()
        if isinstance(msg, Exception):
            raise msg
        return msg

    def close(self):
        for wk in self.workers:
            try:
                wk.child.send(("close", None))
            except Exception:
                pass


class CompositeVecEnv(_VecBase):
    """Several vectorised environments side by side (workers of part p are rows [lo_p, hi_p) of every array).  ``step`` /
    ``reset`` behave like one environment over all workers; the trainer's pipelined rollout steps the ``parts`` one at a time
    (while one part is being stepped on the host the device runs the other part's forward pass).  All state lives in the
    parts, so both ways of stepping can be mixed."""

    def __init__(self, parts):
        self.parts = list(parts)
        self.bounds, lo = [], 0
        for p in self.parts:
            self.bounds.append((lo, lo + p.num_envs))
            lo += p.num_envs
        self.num_envs = lo
        e = self.parts[0]
        self.observation_space_shape = tuple(e.observation_space_shape)
        self.num_actions = int(e.num_actions)
        self.max_episode_steps = int(e.max_episode_steps)

    def reset(self, out=None):
        out = self._alloc(out)
        for p, (lo, hi) in zip(self.parts, self.bounds):
            p.reset(out=out[lo:hi])
        return out

    def step(self, actions, out=None, on_rows=None):
        out = self._alloc(out)
        actions = np.asarray(actions)
        rewards = np.zeros(self.num_envs, dtype=np.float32)
        dones = np.zeros(self.num_envs, dtype=bool)
        infos = []
        for p, (lo, hi) in zip(self.parts, self.bounds):
            cb = None if on_rows is None else (lambda a, b, lo=lo: on_rows(lo + a, lo + b))
            _, rewards[lo:hi], dones[lo:hi], inf = p.step(actions[lo:hi], out=out[lo:hi], on_rows=cb)
            infos.extend(inf)
        return out, rewards, dones, infos

    def close(self):
        for p in self.parts:
            p.close()


def make_vec_env(env_config: dict, num_envs: int, first_worker_id: int = 0, groups: int = 1):
    """Pick the front-end for ``env_config["type"]``.  ``groups`` > 1: a ``CompositeVecEnv`` of that many equal parts (worker
    ids stay consecutive, so every worker sees the same stream as in a single front-end)."""
    if groups > 1 and num_envs % groups == 0:
        per = num_envs // groups
        return CompositeVecEnv([make_vec_env(env_config, per, first_worker_id + g * per) for g in range(groups)])
    if env_config["type"] == "Synthetic":
        from environments.synthetic import SyntheticVecEnv
        keys = ("obs_shape", "num_actions", "max_episode_steps", "seed", "p_reward", "p_done", "pool", "copy_threads", "row_chunks", "step_cost_us", "gen_threads")
        kw = {k: env_config[k] for k in keys if k in env_config}
        if "obs_shape" in kw:
            kw["obs_shape"] = tuple(kw["obs_shape"])
        return SyntheticVecEnv(num_envs, first_worker_id=first_worker_id, **kw)
    if env_config.get("vectorize", "pipe") == "serial":
        from utils import create_env
        return SerialVecEnv([create_env(env_config, worker_id=first_worker_id + w) for w in range(num_envs)])
    return PipeVecEnv(env_config, num_envs, first_worker_id)
