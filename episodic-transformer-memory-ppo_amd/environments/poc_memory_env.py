"""Proof-of-concept memory task (BASELINE config 1): a 1-D corridor with a rewarding and a punishing end.

Own implementation of the task described in upstream environments/poc_memory_env.py: the two goal signs are
visible only during the first two steps (while the agent is frozen), afterwards the observation shows only the
position, so solving it requires memory.  Observation [goal_left, position, goal_right]; actions {0: left, 1: right};
reward -0.1 per step, +/-(1 + 0.1 * min_steps) at the ends; episodes are cut at ``max_episode_steps``.
Uses its own ``numpy.random.Generator`` (seedable) instead of the global numpy RNG.
"""
from types import SimpleNamespace

import numpy as np


class PocMemoryEnv:
    def __init__(self, step_size: float = 0.2, glob: bool = False, freeze: bool = False, max_episode_steps: int = -1,
                 seed=None):
        self.freeze = freeze
        self._step = step_size
        self.max_episode_steps = max_episode_steps
        self._min_steps = int(1.0 / step_size) + 1
        self._penalty = 0.1
        self._show_steps = 2
        self._rng = np.random.default_rng(seed)
        ticks = int(0.4 / step_size)
        if glob:
            lo, hi = -1 + step_size, 1
        else:
            lo = min(-2.0 * step_size, -ticks * step_size)
            hi = max(3.0 * step_size, step_size, (ticks + 1) * step_size)
        grid = np.arange(lo, hi, step_size).clip(-1 + step_size, 1 - step_size)
        self.possible_positions = [round(float(x), 2) for x in grid]

    @property
    def observation_space(self):
        return SimpleNamespace(shape=(3,), low=0.0, high=1.0, dtype=np.float32)

    @property
    def action_space(self):
        return SimpleNamespace(n=2)

    def _obs(self, show_goals: bool):
        if show_goals:
            return np.asarray([self._goals[0], self._pos, self._goals[1]], dtype=np.float32)
        return np.asarray([0.0, self._pos, 0.0], dtype=np.float32)

    def reset(self, **kwargs):
        self._pos = float(self._rng.choice(self.possible_positions))
        self._goals = np.asarray([-1.0, 1.0])[self._rng.permutation(2)]
        self._rewards = []
        self._t = 0
        return self._obs(True)

    def step(self, action):
        a = int(np.asarray(action).reshape(-1)[0])
        direction = 1.0 if a == 1 else -1.0
        done = self.max_episode_steps > 0 and self._t >= self.max_episode_steps - 1
        showing = self._t < self._show_steps
        if showing and self.freeze:
            self._t += 1
            self._rewards.append(0.0)
            return self._obs(True), 0.0, bool(done), None
        self._pos = float(np.round(self._pos + direction * self._step, 2))
        obs = self._obs(showing)
        reward, success = 0.0, False
        bonus = 1.0 + self._min_steps * self._penalty
        if self._pos == -1.0 or self._pos == 1.0:
            good = self._goals[0 if self._pos == -1.0 else 1] == 1.0
            reward = bonus if good else -bonus
            success = bool(good)
            done = True
        else:
            reward = -self._penalty
        self._rewards.append(reward)
        self._t += 1
        info = {"success": success, "reward": float(sum(self._rewards)), "length": len(self._rewards)} if done else None
        return obs, reward, bool(done), info

    def close(self):
        return None
