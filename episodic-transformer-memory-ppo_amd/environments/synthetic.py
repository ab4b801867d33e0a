"""Synthetic environments used for the throughput metric and for parity tests.

Two forms that generate *identical* streams:

* ``SyntheticEnv`` -- one env with the reference's per-worker env API
  (``reset() -> obs``; ``step(action) -> (obs, reward, done, info-or-None)``;
  properties ``observation_space``, ``action_space``, ``max_episode_steps``;
  cf. /root/reference/worker.py:20-34, README.md:216).  Used behind the pipe
  ``Worker`` and when the real reference is driven for golden vectors.
* ``SyntheticVecEnv`` -- the batched in-process form the MI355X trainer steps
  (one call per rollout step for all workers, observations written straight
  into a caller-provided -- normally pinned -- host buffer).

Workload (BASELINE.md section 3 / SURVEY.md section 8d): observation ~ U[0,1) float32 of
``obs_shape`` from ``numpy.random.default_rng(seed + worker_id)``, reward
Bernoulli(0.05), done when the episode reaches ``max_episode_steps`` or with
probability ``p_done`` per step.  Observations are drawn once per worker into
a ring of ``pool`` frames at construction and then replayed (one frame per
emitted observation), so that RNG cost -- which is not what the metric is about
-- stays out of the timed region while every step still hands a fresh host
buffer to the trainer.

``pool: 0`` (round 5) is SURVEY section 8d to the letter: EVERY emitted observation is a fresh
``default_rng(seed + worker_id).random(obs_shape, dtype=float32)`` draw, written straight into the
output row (21,168 draws per env-step at 3x84x84: ~30 us of one host core); the (reward, done)
uniforms then come from a second generator, ``default_rng((seed + worker_id, 1))``, so that both
forms still produce identical streams whatever the order of their draws.  ``gen_threads`` > 1
(vector form): the rows of a step are drawn by that many threads (numpy releases the GIL while it
fills) -- what upstream's worker processes do with one process per environment.

Round 6: the fresh draws of the vector form come from ``libetm_envgen.so`` (csrc/envgen.cc): numpy's PCG64 float32 stream restated in
C with the LCG advanced in 32 AVX-512 (or 8 scalar) lanes -- the SAME floats bit for bit (tests/test_host_logic.py), 9 - 12 us per
3x84x84 observation instead of 30 - 50, the rows of a step drawn side by side by ``gen_threads`` threads of the library's pool.
Without the library numpy draws them (identical values).
"""
from types import SimpleNamespace

import numpy as np

_CHUNK = 4096
_copiers = {}   # threads -> (library, copier handle): one pool of helper threads per process, shared by all front-ends
_copier_spin = None   # None: the library's default (helpers spin ~1 ms between jobs); False: they sleep at once (etm/hostcpu.py)


def set_copier_spin(spin):
    """Do the copier's helper threads spin between jobs (default) or sleep at once?  Applies to existing and future copiers."""
    global _copier_spin
    _copier_spin = None if spin else False
    for lib, handle in _copiers.values():
        lib.etm_host_copier_set_spin(handle, 40000 if spin else 0)
    if _native_pools:
        from environments import envgen
        for handle in _native_pools.values():
            envgen.load().etm_envgen_pool_set_spin(handle, 40000 if spin else 0)


_native_pools = {}   # threads -> pool handle of libetm_envgen.so (one per process and size, like the copier)


def _native_pool(threads):
    if threads not in _native_pools:
        from environments import envgen
        handle = envgen.load().etm_envgen_pool_create(int(threads))
        if not handle:
            raise RuntimeError(f"etm_envgen_pool_create({threads}) failed")
        if _copier_spin is False:
            envgen.load().etm_envgen_pool_set_spin(handle, 0)
        _native_pools[threads] = handle
    return _native_pools[threads]


def _copier(threads):
    """The process-wide multi-threaded copier of the kernel library (etm_host_copier_create); created on first use."""
    if threads not in _copiers:
        from etm import lib as etm_lib
        lib = etm_lib.load()
        handle = lib.etm_host_copier_create(int(threads))
        if not handle:
            raise RuntimeError(f"etm_host_copier_create({threads}) failed")
        if _copier_spin is False:
            lib.etm_host_copier_set_spin(handle, 0)
        _copiers[threads] = (lib, handle)
    return _copiers[threads]


class _WorkerStream:
    """Per-worker RNG state: frame ring + chunked (reward, done) uniforms."""

    def __init__(self, worker_id, obs_shape, seed, pool):
        self.rng = np.random.default_rng(seed + worker_id)
        if pool > 0:
            self.frames = self.rng.random((pool,) + tuple(obs_shape), dtype=np.float32)
            self.rng_obs = None
        else:                       # fresh draws: observations from default_rng(seed + worker_id), uniforms from a second stream
            self.frames = None
            self.rng_obs = self.rng
            self.rng = np.random.default_rng((seed + worker_id, 1))
        self.u = None
        self.upos = _CHUNK

    def next_uniforms(self):
        if self.upos >= _CHUNK:
            self.u = self.rng.random((_CHUNK, 2))
            self.upos = 0
        row = self.u[self.upos]
        self.upos += 1
        return row


class SyntheticEnv:
    """Single env, reference env API.  ``worker_id`` selects the RNG stream."""

    def __init__(self, obs_shape=(3, 84, 84), num_actions=3, max_episode_steps=96, seed=0, worker_id=0,
                 p_reward=0.05, p_done=0.02, pool=64):
        self._shape = tuple(obs_shape)
        self._n_act = int(num_actions)
        self._T = int(max_episode_steps)
        self._p_r, self._p_d = float(p_reward), float(p_done)
        self._s = _WorkerStream(worker_id, self._shape, seed, pool)
        self._pool = pool
        self._cursor = 0
        self._t = 0
        self._ret = 0.0

    @property
    def observation_space(self):
        return SimpleNamespace(shape=self._shape, low=0.0, high=1.0, dtype=np.float32)

    @property
    def action_space(self):
        return SimpleNamespace(n=self._n_act)

    @property
    def max_episode_steps(self):
        return self._T

    def _emit(self):
        if self._pool == 0:
            return self._s.rng_obs.random(self._shape, dtype=np.float32)
        frame = self._s.frames[self._cursor % self._pool]
        self._cursor += 1
        return frame

    def reset(self, **kwargs):
        self._t = 0
        self._ret = 0.0
        return self._emit()

    def step(self, action):
        u_r, u_d = self._s.next_uniforms()
        reward = 1.0 if u_r < self._p_r else 0.0
        self._t += 1
        self._ret += reward
        done = bool(self._t >= self._T or u_d < self._p_d)
        if done:
            # terminal observation is never consumed by the trainer (it resets immediately): no frame is spent
            return np.zeros(self._shape, dtype=np.float32), reward, True, {"reward": self._ret, "length": self._t}
        return self._emit(), reward, False, None

    def close(self):
        return None


class SyntheticVecEnv:
    """Batched form of ``num_envs`` ``SyntheticEnv`` instances (same streams, auto-reset on done)."""

    def __init__(self, num_envs, obs_shape=(3, 84, 84), num_actions=3, max_episode_steps=96, seed=0,
                 p_reward=0.05, p_done=0.02, pool=64, first_worker_id=0, copy_threads=1, row_chunks=None, step_cost_us=0.0, min_chunked_envs=None,
                 gen_threads=1):
        """``copy_threads`` > 1: the observation rows of a step are written by that many threads (the kernel library's host
        copier; the reference's workers write theirs in n_workers processes) instead of one numpy copy.  ``pool`` = 0: every row is a
        fresh draw of its worker's generator (module docstring); ``gen_threads`` > 1: drawn by that many threads."""
        self.num_envs = int(num_envs)
        # step_cost_us: a simulated simulator -- every environment of this front-end burns that much CPU time per step, one after
        # the other (what stepping real Python environments in one process costs; worker processes run them side by side)
        self._step_cost_s = float(step_cost_us) * 1e-6
        self._copy_threads = int(copy_threads)
        if row_chunks is not None:
            self.ROW_CHUNKS = int(row_chunks)          # instance override of the on_rows granularity
        if min_chunked_envs is not None:
            self.MIN_CHUNKED_ENVS = int(min_chunked_envs)
        self._row_bytes = int(np.prod(obs_shape)) * 4
        self.observation_space_shape = tuple(obs_shape)
        self.num_actions = int(num_actions)
        self.max_episode_steps = int(max_episode_steps)
        self._p_r, self._p_d = float(p_reward), float(p_done)
        self._pool = pool
        streams = [_WorkerStream(first_worker_id + w, obs_shape, seed, pool) for w in range(self.num_envs)]
        self._rngs = [s.rng for s in streams]
        # [pool, W, *obs]: all cursors advance in lock-step, so the observations of one step are ONE contiguous block
        # (a single memcpy per step / per row chunk instead of W strided ones)
        self._frames = np.ascontiguousarray(np.stack([s.frames for s in streams], axis=1)) if pool > 0 else None
        self._rngs_obs = [s.rng_obs for s in streams]
        self._gen_pool = None
        self._native = self._native_pool = self._obs_states = None
        if pool == 0:
            from environments import envgen
            self._native = envgen.load()
            if self._native is not None and int(np.prod(obs_shape)) % 2 == 0:
                # the observation streams live in the library's state vectors from here on (numpy's generators are not advanced)
                self._obs_states = np.ascontiguousarray(np.stack([envgen.state_of(g) for g in self._rngs_obs]))
                if int(gen_threads) > 1:
                    self._native_pool = _native_pool(int(gen_threads))
            else:
                self._native = None
        if pool == 0 and self._native is None and int(gen_threads) > 1:
            from concurrent.futures import ThreadPoolExecutor
            self._gen_pool = ThreadPoolExecutor(max_workers=int(gen_threads))
            self._gen_threads = int(gen_threads)
        self._u = np.empty((self.num_envs, _CHUNK, 2))
        self._upos = _CHUNK
        self._cursor = 0
        # episode state AFTER the planned chunk (see _plan); per-position plan of the current chunk of uniforms
        self._t = np.zeros(self.num_envs, dtype=np.int64)
        self._ret = np.zeros(self.num_envs, dtype=np.float64)
        self._rew_c = np.zeros((_CHUNK, self.num_envs), dtype=np.float32)
        self._done_c = np.zeros((_CHUNK, self.num_envs), dtype=bool)
        self._any_done = np.zeros(_CHUNK, dtype=bool)
        self._info_c = {}

    ROW_CHUNKS = 2   # on_rows granularity (vec_env protocol); 2 measured best: every upload call costs ~9 us of host time
    MIN_CHUNKED_ENVS = 16   # fewer environments (a small worker group): one notification for all rows

    def _draw_rows(self, out, lo, hi):
        for w in range(lo, hi):
            self._rngs_obs[w].random(dtype=np.float32, out=out[w])

    def _emit_fresh(self, out, on_rows):
        """pool = 0: every row of the step is drawn now, by its worker's own generator, into the output row."""
        if out is None:
            out = np.empty((self.num_envs,) + self.observation_space_shape, dtype=np.float32)
        if not (out.flags.c_contiguous and out.dtype == np.float32):
            raise ValueError("fresh observations are drawn in place: a C-contiguous float32 output buffer is needed")
        chunks = (self.ROW_CHUNKS if self.num_envs >= self.MIN_CHUNKED_ENVS else 1) if on_rows is not None else 1
        step = max(1, -(-self.num_envs // chunks))
        row_floats = self._row_bytes // 4
        for lo in range(0, self.num_envs, step):
            hi = min(lo + step, self.num_envs)
            if self._native is not None:
                if self._native.etm_pcg64_fill_rows_f32(self._native_pool, self._obs_states[lo:].ctypes.data,
                                                        out.ctypes.data + lo * self._row_bytes, row_floats, hi - lo) != 0:
                    raise RuntimeError("etm_pcg64_fill_rows_f32 failed")
            elif self._gen_pool is not None and hi - lo > 1:
                k = min(self._gen_threads, hi - lo)
                per = -(-(hi - lo) // k)
                futs = [self._gen_pool.submit(self._draw_rows, out, a, min(a + per, hi)) for a in range(lo + per, hi, per)]
                self._draw_rows(out, lo, min(lo + per, hi))         # this thread takes the first share
                for f in futs:
                    f.result()
            else:
                self._draw_rows(out, lo, hi)
            if on_rows is not None:
                on_rows(lo, hi)
        return out

    def _emit(self, out, on_rows=None):
        if self._pool == 0:
            return self._emit_fresh(out, on_rows)
        frame = self._frames[self._cursor % self._pool]
        self._cursor += 1
        if out is None:
            out = np.empty_like(frame)
        if on_rows is None:
            np.copyto(out, frame)
            return out
        chunks = self.ROW_CHUNKS if self.num_envs >= self.MIN_CHUNKED_ENVS else 1
        step = max(1, -(-self.num_envs // chunks))
        if self._copy_threads > 1 and out.flags.c_contiguous and out.dtype == np.float32:
            lib, handle = _copier(self._copy_threads)
            dst, src = out.ctypes.data, frame.ctypes.data
            for lo in range(0, self.num_envs, step):
                hi = min(lo + step, self.num_envs)
                if lib.etm_host_copy(handle, dst + lo * self._row_bytes, src + lo * self._row_bytes, (hi - lo) * self._row_bytes) != 0:
                    raise RuntimeError("etm_host_copy failed")
                on_rows(lo, hi)
            return out
        for lo in range(0, self.num_envs, step):       # rows are handed over as soon as they are written
            hi = min(lo + step, self.num_envs)
            np.copyto(out[lo:hi], frame[lo:hi])
            on_rows(lo, hi)
        return out

    def reset(self, out=None):
        self._t[:] = 0
        self._ret[:] = 0.0
        if self._upos < _CHUNK:
            self._plan(self._upos)           # the rest of the current chunk, from fresh episodes
        return self._emit(out)

    def _plan(self, p0):
        """Rewards, done flags and episode results of the chunk positions [p0, _CHUNK) from the pre-drawn uniforms, starting from
        the episode state (self._t, self._ret) -- what the per-step arithmetic of ``SyntheticEnv.step`` yields, evaluated once per
        chunk instead of with a handful of small array operations per step (they were ~10 us of every rollout step on the host).
        Leaves (self._t, self._ret) at the state AFTER the chunk's last position."""
        k, T = self.num_envs, self.max_episode_steps
        rew = self._u[:, p0:, 0] < self._p_r                    # [k, n]
        bern = self._u[:, p0:, 1] < self._p_d
        n = rew.shape[1]
        self._rew_c[p0:] = rew.T
        done = self._done_c
        done[p0:] = False
        self._info_c = {key: val for key, val in self._info_c.items() if key[0] < p0}
        csum = np.cumsum(rew, axis=1, dtype=np.int64)
        for w in range(k):
            t0, ret0, start, j = int(self._t[w]), float(self._ret[w]), 0, 0
            pos = np.flatnonzero(bern[w])
            while True:
                cut = start + T - t0 - 1                        # position at which the episode reaches max_episode_steps
                while j < len(pos) and pos[j] < start:
                    j += 1
                i = min(cut, int(pos[j]) if j < len(pos) else n)
                if i >= n:
                    break
                before = int(csum[w, start - 1]) if start > 0 else 0
                done[p0 + i, w] = True
                self._info_c[(p0 + i, w)] = (ret0 + float(int(csum[w, i]) - before), t0 + (i - start + 1))
                start, t0, ret0 = i + 1, 0, 0.0
            before = int(csum[w, start - 1]) if start > 0 else 0
            self._t[w] = t0 + (n - start)
            self._ret[w] = ret0 + float((int(csum[w, n - 1]) if n > 0 else 0) - before)
        self._any_done = done.any(axis=1)

    def step(self, actions, out=None, on_rows=None):
        # the frame a worker shows next does not depend on its done flag (a finished worker's next frame IS its reset observation),
        # so the rows go out first: their upload overlaps the bookkeeping below
        if self._step_cost_s > 0.0:
            import time
            t_end = time.perf_counter() + self._step_cost_s * self.num_envs
            while time.perf_counter() < t_end:
                pass
        obs = self._emit(out, on_rows)
        if self._upos >= _CHUNK:
            for w, rng in enumerate(self._rngs):
                self._u[w] = rng.random((_CHUNK, 2))
            self._upos = 0
            self._info_c = {}
            self._plan(0)
        i = self._upos
        self._upos += 1
        rewards, dones = self._rew_c[i].copy(), self._done_c[i].copy()
        infos = [None] * self.num_envs
        if self._any_done[i]:
            for w in np.flatnonzero(dones):
                ret, length = self._info_c[(i, int(w))]
                infos[w] = {"reward": float(ret), "length": int(length)}
        return obs, rewards, dones, infos

    def close(self):
        if getattr(self, "_gen_pool", None) is not None:
            self._gen_pool.shutdown(wait=False)
            self._gen_pool = None
        return None
