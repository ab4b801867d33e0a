"""PPO trainer with TransformerXL episodic memory on one MI355X per process.

Keeps upstream's ``PPOTrainer`` surface (trainer.py:17 ctor, :101 run_training, :145 _sample_training_data,
:227 get_last_value, :239 _train_epochs, :258 _train_mini_batch, :364 close) and its bookkeeping semantics --
mask / window-index tables (bit-exact, :78, :88-90), per-step window rule (:165-166), the different window of
``get_last_value`` (:230-236, quirk Q5), episode hand-over on ``done`` (:195-213) -- while the data path is redesigned
for the GPU:

* rollout state (episode bank, tables, buffer) is HBM-resident; per step the observation rows are streamed from pinned
  memory into the staging array while the environments still step and the actions are stored into pinned memory by the
  sampling kernel (upstream: one ``.cpu()`` per worker, :190); the step itself is two captured HIP graphs;
* memory windows are never gathered: the attention kernel reads the bank through (slot, row) indices;
* GAE and the PPO loss (+ its backward) are single fused kernels; loss statistics stay on the device and are
  fetched once per update (upstream: 6 host syncs per minibatch, :318-323); the whole minibatch step (gather, forward,
  loss, backward, clip, AdamW) is one captured HIP graph with device-resident schedules;
* data parallelism (absent upstream): ``dp`` all-reduces (sums) one flat gradient bucket with RCCL between ``backward()``
  and gradient clipping (:310-311) -- the division by the world size rides in the clip coefficient -- and merges advantage
  statistics so normalisation is over the global minibatch.

There is no CPU path: constructing the trainer without a HIP device raises.
"""
import os
import sys
import pickle
import time
from collections import deque

import numpy as np
import torch

# Four worker groups want four pairs of streams (upload + step graph) running CONCURRENTLY; the HIP runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and reads that variable once, when it starts (the first device query of the
# process).  With 4 queues four groups serialise (rollout 0.161 s per update instead of 0.087 s; two groups: 0.094 s), so "auto"
# only trusts a value that is in the environment when this module is imported -- the entry points (bench.py, train.py, the
# tools) set GPU_MAX_HW_QUEUES=8 as their first statement, before torch is imported; a caller who does not gets two groups.
def _trusted_hw_queues() -> int:
    """GPU_MAX_HW_QUEUES as the HIP runtime of this process has read it (or will): the value in os.environ counts only if the runtime
    has not started yet, or the process was STARTED with it (/proc/self/environ), or one of this package's entry points set it as
    its first statement (they leave the marker ETM_HW_QUEUES_SET_EARLY).  An application that initialised torch.cuda first and set
    the variable afterwards runs on the runtime's default four queues whatever os.environ says now: 0."""
    try:
        val = int(os.environ.get("GPU_MAX_HW_QUEUES", "0"))
    except ValueError:
        return 0
    if val < 8 or not torch.cuda.is_initialized() or os.environ.get("ETM_HW_QUEUES_SET_EARLY") == "1":
        return val
    try:
        with open("/proc/self/environ", "rb") as f:
            for item in f.read().split(b"\0"):
                if item.startswith(b"GPU_MAX_HW_QUEUES=") and int(item.split(b"=", 1)[1]) >= 8:
                    return val
    except (OSError, ValueError):
        pass
    return 0


_HW_QUEUES_AT_IMPORT = _trusted_hw_queues()


def default_rollout_groups(num_workers: int, min_group: int) -> int:
    """``rollout_groups: auto``: 4 when the runtime has 8 hardware queues (see above) and the groups keep ``min_group`` workers,
    else 2 (when THEY keep ``min_group`` workers), else 1."""
    if _HW_QUEUES_AT_IMPORT >= 8 and num_workers % 4 == 0 and num_workers // 4 >= min_group:
        return 4
    if num_workers % 2 == 0 and num_workers // 2 >= min_group:
        return 2
    return 1


from buffer import Buffer
from environments.vec_env import make_vec_env
from etm import lib as etm_lib
from etm import ops
from etm.ops import WindowSpec
from etm.optim import FlatAdamW
from model import ActorCriticModel, IndexedObservations
from utils import polynomial_decay, process_episode_info
from trainer_parts import _DataParallelStep, _NativeRolloutDrive, _RunOutputs


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    def close(self):
        pass


def _make_writer(run_id):
    try:
        from torch.utils.tensorboard import SummaryWriter
    except Exception:
        return _NullWriter()
    os.makedirs("./summaries", exist_ok=True)
    return SummaryWriter("./summaries/" + run_id + time.strftime("/%Y%m%d-%H%M%S/"))


def build_window_tables(memory_length: int, max_episode_length: int):
    """Attention-mask table [L, L] and sliding-window index table [T, L] (upstream trainer.py:78, :88-90).

    Integer/0-1 valued, built on the host: row s of the index table is the window used at episode step s
    ([0..L-1] while s < L-1, then [s-L+1..s]); mask row r has its first r entries set.
    """
    L, T = int(memory_length), int(max_episode_length)
    if T < L:
        raise ValueError(f"max_episode_steps ({T}) must be >= memory_length ({L})")
    mask = torch.tril(torch.ones((L, L), dtype=torch.float32), diagonal=-1)
    first = torch.clamp(torch.arange(T, dtype=torch.int64) - (L - 1), min=0)
    indices = first.unsqueeze(1) + torch.arange(L, dtype=torch.int64).unsqueeze(0)
    return mask, indices


def check_kernel_shapes(tcfg: dict):
    """Fail early (before any environment or buffer is built) if the transformer shape is outside what the gfx950
    kernels are built for; there is no fallback path."""
    d, h, mem = tcfg["embed_dim"], tcfg["num_heads"], tcfg["memory_length"]
    if d % h != 0:
        raise ValueError("Embedding dimension needs to be divisible by the number of heads")
    hd = d // h
    problems = []
    if d % 32 != 0 or d > 1024:
        problems.append(f"embed_dim={d} (need a multiple of 32, <= 1024)")
    if hd % 32 != 0 or hd > 128:
        problems.append(f"head_dim={hd} (need 32, 64, 96 or 128)")
    if mem > 128:
        problems.append(f"memory_length={mem} (need <= 128)")
    if problems:
        raise ValueError("transformer shape not supported by the MI355X kernels: " + "; ".join(problems))


class PPOTrainer(_DataParallelStep, _NativeRolloutDrive, _RunOutputs):
    def __init__(self, config: dict, run_id: str = "run", device: torch.device = None, env=None, dp=None,
                 first_worker_id: int = 0, tensorboard: bool = True) -> None:
        if device is None:
            device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("PPOTrainer needs an MI355X (HIP) device: this build has no CPU training path "
                               "(the CPU restatement under oracle/ is test infrastructure only)")
        etm_lib.load()  # fail loudly if the kernels are not built
        # (placement of the step kernel's workgroups: a worker's whole team on one XCD, the library's default; the member-per-XCD map
        # measured equal -- 111.9 vs 112.2 us per step graph -- and is a kernel-level test only since round 6)
        etm_lib.check(etm_lib.load().etm_rollout_trxl_set_placement(0), "etm_rollout_trxl_set_placement")
        self.config = config
        self.device = device
        self.run_id = run_id
        self.dp = dp
        self.num_workers = config["n_workers"]
        self.lr_schedule = config["learning_rate_schedule"]
        self.beta_schedule = config["beta_schedule"]
        self.cr_schedule = config["clip_range_schedule"]
        t = config["transformer"]
        self.memory_length, self.num_blocks, self.embed_dim = t["memory_length"], t["num_blocks"], t["embed_dim"]
        check_kernel_shapes(t)
        self.writer = _make_writer(run_id) if tensorboard else _NullWriter()

        # environments (batched front-end over the upstream per-worker protocol)
        # rollout_groups (default "auto": 4 with 8 hardware queues, else 2): the workers are stepped as that many groups in a software
        # pipeline -- while the host steps one group's environments the device runs the other groups' forward passes (small step
        # graphs overlap almost perfectly on the GPU: 117 us per round of two, 122 us per round of four vs 112 us each,
        # tools/rollout_profile.py; config 3: rollout 0.094 s per update with two groups, 0.087 s with four).  Needs an environment front-end made
        # of parts (make_vec_env(groups=...)); an externally supplied environment is stepped as one group.
        # Groups of fewer than 8 workers are not formed: they do not pay off, and one intermittent mismatch was seen with
        # groups of 2 workers while both groups' graphs had been captured on torch's shared capture stream, i.e. with ONE
        # BLAS scratch buffer between them (split-K solutions at a handful of rows per GEMM); the graphs are captured per
        # group stream now (_capture_step_graph), the threshold stays until that has been re-measured.
        min_group = int(config.get("rollout_min_group_size", 8))
        n_groups = config.get("rollout_groups", "auto")
        n_groups = default_rollout_groups(self.num_workers, min_group) if n_groups == "auto" else int(n_groups)
        if n_groups < 1 or self.num_workers % n_groups != 0 or self.num_workers // n_groups < min_group:
            n_groups = 1
        if config.get("rollout_groups", "auto") == "auto" and os.environ.get("ETM_QUIET") != "1":
            print(f"[etm] rollout worker groups: {n_groups} (rollout_groups: auto; hardware queues the HIP runtime is known to have been "
                  f"started with: {_HW_QUEUES_AT_IMPORT or 'runtime default (4)'})", file=sys.stderr, flush=True)
        # round 5: the rank's host CPU share (affinity mask, cgroup CPU quota, ranks per node) against the threads a rollout keeps
        # busy -- trainer thread on the action flag, observation-copier helpers, worker processes (etm/hostcpu.py).  With room to
        # spare nothing changes; without it the helpers sleep between jobs, the copier / the worker processes shrink to what fits and
        # the trainer thread sleeps through most of the expected device time before it spins.  `host_cpu_plan: false` keeps the
        # configured values whatever the host looks like.
        from etm import hostcpu
        env_cfg = dict(config["environment"])
        # the threads that write a step's observation rows: the copier's (frame ring) or the generator pool's (pool: 0, fresh draws)
        row_threads = ("gen_threads" if int(env_cfg.get("pool", 64)) == 0 else "copy_threads") if env_cfg.get("type") == "Synthetic" else None
        self._host_plan = hostcpu.plan_host_threads(copy_threads=int(env_cfg.get(row_threads, 1)) if row_threads else 1,
                                                    worker_processes=bool(env is None and config.get("worker_processes", False)),
                                                    num_envs=self.num_workers, envs_per_process=int(config.get("envs_per_process", 1)),
                                                    groups=n_groups, quiet=not config.get("host_cpu_plan", True))
        if not config.get("host_cpu_plan", True):
            self._host_plan.update(copy_threads=int(env_cfg.get(row_threads, 1)) if row_threads else 1, copier_spin=True, polite_wait=False,
                                   envs_per_process=int(config.get("envs_per_process", 1)), worker_spin=True, reason="host_cpu_plan: false")
        if row_threads and row_threads in env_cfg:
            env_cfg[row_threads] = self._host_plan["copy_threads"]
        if not self._host_plan["copier_spin"]:
            from environments import synthetic as _syn
            _syn.set_copier_spin(False)
        if self._host_plan["polite_wait"]:
            hostcpu.set_timer_slack_ns(1000)
        self._flag_wait_ema = 0.0
        ops.set_ln_grad_kernel(config.get("fused_ln_grad", True))      # (true / "outputs": from the window passes' outputs; "rows": round 5's pass over the window rows; false: the generic dX kernel -- tested options)
        # worker_processes (round 4; upstream trainer.py:62-66, worker.py): the environments live in worker PROCESSES over one shared,
        # HIP-registered segment (environments/shm_env.py): they take their actions straight from the device and step concurrently;
        # the per-step host loop is then the native driver of the kernel library (etm_rollout_drive) -- see _sample_training_data
        self._shm_env = None
        if env is None and config.get("worker_processes", False):
            from environments.shm_env import ShmVecEnv
            self._shm_env = ShmVecEnv(env_cfg, self.num_workers, first_worker_id, groups=n_groups,
                                      envs_per_proc=self._host_plan["envs_per_process"], steps_per_rollout=config["worker_steps"],
                                      spin=self._host_plan["worker_spin"])
            env = self._shm_env
        self.env = env if env is not None else make_vec_env(env_cfg, self.num_workers, first_worker_id, groups=n_groups)
        W = self.num_workers
        obs_shape = tuple(self.env.observation_space_shape)
        self.observation_space = type("Space", (), {"shape": obs_shape})()
        self.action_space_shape = (self.env.num_actions,)  # one branch, like upstream (trainer.py:47)
        self.max_episode_length = self.env.max_episode_steps

        if config.get("tunable_gemm", os.environ.get("ETM_TUNABLE_GEMM", "1") != "0"):
            # let PyTorch pick the fastest hipBLASLt / rocBLAS solution per GEMM shape (the small [N, D] x [D, D] products around
            # the kernels are far from the libraries' default heuristics: -5 % optimisation time, -14 us per rollout step);
            # every shape is met in the eager warm-up steps, i.e. before any graph capture.  fp32 in, fp32 out: only the
            # summation order can differ.
            try:
                import torch.cuda.tunable as tunable
                tunable.enable(True)
                tunable.tuning_enable(True)
                # results file (written by the library at exit) goes to the temp directory, not the working directory
                # one results file per rank: data-parallel ranks tune independently and must not write the same file
                rank_tag = "" if dp is None else f"_rank{dp.rank}"
                tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"etm_tunableop_results{rank_tag}.csv"), True)
                tunable.set_max_tuning_duration(int(os.environ.get("ETM_TUNABLE_MS", 30)))
                tunable.set_max_tuning_iterations(int(os.environ.get("ETM_TUNABLE_ITERS", 20)))
            except Exception as exc:        # an older / newer torch without this API: run with the default heuristics
                print(f"[trainer] per-shape GEMM tuning not available ({exc})")
        self.buffer = Buffer(config, self.observation_space, self.action_space_shape, self.max_episode_length, device)
        self.model = ActorCriticModel(config, self.observation_space, self.action_space_shape, self.max_episode_length).to(device)
        self.model.train()
        self.model.graph_refresh = bool(config.get("hip_graph_rollout", True))      # (refresh_rollout_weights as a graph replay from its third call on)
        if self.dp is not None:
            self.dp.broadcast_parameters(self.model)
        self.params = [p for p in self.model.parameters() if p.requires_grad]
        # The optimisation step of one minibatch (gather, forward, loss, backward, clipping, AdamW) is captured in a HIP graph
        # after two eager warm-up steps and replayed for every other minibatch of the run: the host then issues one launch
        # per minibatch instead of ~280.  lr / clip range / entropy coefficient live on the device so that their schedules
        # keep working under replay.  (Data-parallel runs replay two graphs around the eager RCCL all-reduce of the bucket.)
        self._use_train_graph = bool(config.get("hip_graph_train", True))
        self._train_graph = None
        self._train_warm = 0
        self._obs_train = None
        if self._use_train_graph:
            self._dyn = torch.zeros(2, dtype=torch.float64, device=device)      # (clip range, entropy coefficient)
            self._sched_host = [None, None, None]   # (lr, clip, beta) currently on the device
            self.profile_sample_every = 0     # bench.py: run every k-th minibatch eagerly so that per-kernel events exist
            self._mb_counter = 0
        # AdamW (torch defaults, like upstream's optim.AdamW(parameters, lr)) + global-norm clipping on flat arenas: parameters,
        # gradients and both moments share one layout, the step is two launches (etm/optim.py, csrc/optim.hip); lr and the step
        # counter live on the device.  The gradient arena is the one bucket the data-parallel all-reduce sums.
        self.optimizer = FlatAdamW(self.params, lr=self.lr_schedule["initial"])
        self.flat_grads = self.optimizer.flat_grads[: self.optimizer.total]
        self._grad_views = self.optimizer.grad_views
        if self.dp is not None:
            self.dp.flat = self.flat_grads
        self._build_grad_groups()

        # host <-> device staging (pinned)
        if self._shm_env is not None:
            # observation rows and action words live in the workers' shared segment, registered with the HIP runtime: the copy
            # engine reads the rows there, the sampling kernel writes the actions (and the step's sequence number) there
            seg = np.frombuffer(self._shm_env.shm.buf, dtype=np.uint8)
            etm_lib.check(etm_lib.load().etm_host_register(seg.ctypes.data, seg.nbytes), "etm_host_register")
            self._shm_registered = seg.ctypes.data
            self._obs_pin = torch.from_numpy(self._shm_env.v["obs"])
            self._act_pin = torch.from_numpy(self._shm_env.v["act"])
        else:
            self._obs_pin = torch.zeros((W,) + obs_shape, dtype=torch.float32).pin_memory()
            self._act_pin = torch.zeros((W, len(self.action_space_shape)), dtype=torch.int64).pin_memory()
        self.obs = self._obs_pin.numpy()
        # (episode step, episode slot) of every worker: one pinned [2, W] block, uploaded with ONE copy per rollout step
        self._ss_pin = torch.zeros((2, W), dtype=torch.int64).pin_memory()
        self._step_pin, self._slot_pin = self._ss_pin[0], self._ss_pin[1]
        self.worker_current_episode_step = self._step_pin.numpy()   # host truth, mirrored on device each step
        self.worker_episode_slot = self._slot_pin.numpy()
        self.worker_episode_slot[:] = np.arange(W)
        self._ss_dev = torch.zeros((2, W), dtype=torch.int64, device=device)
        self._ss_dev[1] = torch.arange(W, dtype=torch.int64, device=device)
        self._step_dev, self._slot_dev = self._ss_dev[0], self._ss_dev[1]
        self.env.reset(out=self.obs)

        # fixed-address operands of the rollout step (HIP-graph friendly) and time-major staging of the step outputs
        S, L, B = config["worker_steps"], self.memory_length, len(self.action_space_shape)
        self._obs_dev = torch.zeros((W,) + obs_shape, dtype=torch.float32, device=device)
        self._t_dev = torch.zeros((), dtype=torch.int64, device=device)
        self._stage = {
            "obs": torch.zeros((S, W) + obs_shape, dtype=torch.float32, device=device),
            "memory_mask": torch.zeros((S, W, L), dtype=torch.bool, device=device),
            "memory_indices": torch.zeros((S, W, L), dtype=torch.int64, device=device),
            "actions": torch.zeros((S, W, B), dtype=torch.int64, device=device),
            "log_probs": torch.zeros((S, W, B), dtype=torch.float32, device=device),
            "values": torch.zeros((S, W), dtype=torch.float32, device=device),
        }
        self._mask_t = torch.zeros((W, L), dtype=torch.bool, device=device)
        self._win_t = torch.zeros((W, L), dtype=torch.int64, device=device)
        self._act_dev = torch.zeros((W, B), dtype=torch.int64, device=device)
        self._uniforms = torch.zeros((S, W), dtype=torch.float32, device=device)
        # teacher forcing (parity tests): a non-negative entry replaces the sampled action of that (step, worker); the table has
        # a fixed address, so the captured step graphs read it too -- every rollout path can be driven with recorded actions
        self._forced_tab = torch.full((S, W), -1, dtype=torch.int64, device=device)
        self._step_graph = None
        self._act_ready = torch.cuda.Event()
        # observation streaming (graph rollout with the fused encoder): rows of the next observation go from pinned memory
        # straight into their row of the time-major staging array on a second stream while the environments still step
        self._flag_pin = torch.zeros((1,), dtype=torch.int64).pin_memory()   # step counter written by the sampling kernel
        self._flag_np = self._flag_pin.numpy()
        self._host_flag = False      # decided when the step graph is captured
        self._up_stream = torch.cuda.Stream(device=device)
        self._chain_log = None       # tools/rollout_profile.py: per-step host timestamps of the first group
        self._up_done = torch.cuda.Event()
        self._stream_obs = False     # decided when the step graph is captured
        self._t_row = torch.zeros((), dtype=torch.int64, device=device)
        self._item = torch.zeros((self.num_blocks, W, self.embed_dim), dtype=torch.float32, device=device)   # block-major
        # rollout K/V cache (weights are frozen while sampling): per worker [T, blocks, 2D] projections of its episode
        self._use_kv_cache = bool(config.get("kv_cache_rollout", True))
        T, nb, D = self.max_episode_length, self.num_blocks, self.embed_dim
        self._kv_cache = torch.zeros((W, T, nb, 2 * D), dtype=torch.float32, device=device)
        self._kv_init = torch.zeros((T, nb, 2 * D), dtype=torch.float32, device=device)
        self._kv_weights = None
        self._worker_ids = torch.arange(W, dtype=torch.int64, device=device)

        # worker groups: the full-width group (eager path, single-group graph path) aliases the buffers above; the pipelined
        # groups own what cannot be a contiguous slice of them
        self._group_all = self._make_group(0, W, self.env, full=True)
        parts = getattr(self.env, "parts", None)
        self._groups = [self._group_all]
        if parts is not None and len(parts) > 1:
            self._groups = [self._make_group(lo, hi, part, full=False) for part, (lo, hi) in zip(parts, self.env.bounds)]

        mask, indices = build_window_tables(self.memory_length, self.max_episode_length)
        self.memory_mask, self.memory_indices = mask, indices                       # host copies (upstream names)
        self._mask_table = mask.bool().contiguous().to(device)
        self._index_table = indices.contiguous().to(device)
        self.last_update_timing = {}

    # ------------------------------------------------------------------ properties mirroring upstream members
    @property
    def memory(self):
        """[W, T, blocks, D] live episodic memory of every worker (upstream ``self.memory``); a gathered copy."""
        return self.buffer.bank.index_select(0, self._slot_dev)

    # ------------------------------------------------------------------ training loop
    def run_training(self) -> None:
        print("Step 6: Starting training using " + str(self.device))
        episode_infos = deque(maxlen=100)
        for update in range(self.config["updates"]):
            lr, beta, clip = self.schedules(update)
            t0 = time.perf_counter()
            sampled_episode_info = self._sample_training_data()
            self.buffer.prepare_batch_dict()
            training_stats, grad_info = self._train_epochs(lr, clip, beta)
            torch.cuda.synchronize(self.device)
            dt = time.perf_counter() - t0
            training_stats = np.mean(training_stats, axis=0)
            episode_infos.extend(sampled_episode_info)
            episode_result = process_episode_info(episode_infos)
            steps_per_s = self.num_workers * self.config["worker_steps"] / dt
            vmean, amean = torch.mean(self.buffer.values).item(), torch.mean(self.buffer.advantages).item()
            if episode_result:
                head = "{:4} reward={:.2f} std={:.2f} length={:.1f} std={:.2f}".format(
                    update, episode_result["reward_mean"], episode_result["reward_std"], episode_result["length_mean"],
                    episode_result["length_std"])
            else:
                head = "{:4} (no finished episode yet)".format(update)
            if "success_percent" in episode_result:
                head += " success={:.2f}".format(episode_result["success_percent"])
            if self._is_main:
                print(head + " pi_loss={:3f} v_loss={:3f} entropy={:.3f} loss={:3f} value={:.3f} advantage={:.3f} steps/s={:.0f}".format(
                    training_stats[0], training_stats[1], training_stats[3], training_stats[2], vmean, amean, steps_per_s))
            self._write_gradient_summary(update, grad_info)
            self._write_training_summary(update, training_stats, episode_result, vmean, amean, steps_per_s)
        if self._is_main:              # replicas are identical: one rank writes the checkpoint
            self._save_model()
        if self.dp is not None:
            self.dp.barrier()

    @property
    def _is_main(self):
        return self.dp is None or self.dp.rank == 0

    def schedules(self, update: int):
        s = lambda c: polynomial_decay(c["initial"], c["final"], c["max_decay_steps"], c["power"], update)
        return s(self.lr_schedule), s(self.beta_schedule), s(self.cr_schedule)

    # ------------------------------------------------------------------ rollout
    def _make_group(self, lo, hi, env, full):
        """Device / pinned state of the workers [lo, hi) for one rollout step (see ``rollout_groups``)."""
        from types import SimpleNamespace
        dev, Wg, B = self.device, hi - lo, len(self.action_space_shape)
        g = SimpleNamespace(lo=lo, hi=hi, W=Wg, env=env, full=full, graphs=None, rf_scratch=None)
        g.obs_pin, g.act_pin = self._obs_pin[lo:hi], self._act_pin[lo:hi]
        g.obs_np = self.obs[lo:hi]
        acts = g.act_pin.numpy()
        g.acts_host = acts[:, 0] if B == 1 else acts            # [Wg] for one branch, [Wg, B] for multi-discrete
        g.obs_dev, g.mask_t, g.win_t, g.act_dev = self._obs_dev[lo:hi], self._mask_t[lo:hi], self._win_t[lo:hi], self._act_dev[lo:hi]
        g.kv = self._kv_cache[lo:hi]
        if full:
            g.ss_pin, g.ss_dev, g.item, g.t_dev, g.t_row = self._ss_pin, self._ss_dev, self._item, self._t_dev, self._t_row
            g.ids, g.flag_pin, g.act_ready, g.up_done, g.stream = self._worker_ids, self._flag_pin, self._act_ready, self._up_done, None
        else:
            g.ss_pin = torch.zeros((2, Wg), dtype=torch.int64).pin_memory()
            g.ss_dev = torch.zeros((2, Wg), dtype=torch.int64, device=dev)
            g.item = torch.zeros((self.num_blocks, Wg, self.embed_dim), dtype=torch.float32, device=dev)
            g.t_dev = torch.zeros((), dtype=torch.int64, device=dev)
            g.t_row = torch.zeros((), dtype=torch.int64, device=dev)
            g.ids = torch.arange(Wg, dtype=torch.int64, device=dev)
            g.flag_pin = torch.zeros((1,), dtype=torch.int64).pin_memory()
            g.act_ready, g.up_done = torch.cuda.Event(), torch.cuda.Event()
            g.stream = torch.cuda.Stream(device=dev)
        g.ss_np = g.ss_pin.numpy()
        g.flag_np = g.flag_pin.numpy()
        g.step_dev, g.slot_dev = g.ss_dev[0], g.ss_dev[1]
        # (episode step, slot) as LATCHED by the head of a step for its tail: the host uploads the next step's block on the
        # upload stream while the tail (bank / cache writes under env.step) may still be running, and only the group's own
        # stream orders tail t before head t + 1 -- so the tail must not read the uploaded block itself
        g.ss_latch = torch.zeros((2, Wg), dtype=torch.int64, device=dev)
        g.step_l, g.slot_l = g.ss_latch[0], g.ss_latch[1]
        return g

    def _sample_training_data(self, forced_actions=None) -> list:
        """Runs all workers for ``worker_steps`` steps; fills the buffer; returns finished-episode infos.

        The device work of one step (window lookup, model forward, action sampling, staging of the step's buffer rows,
        action hand-over; then memory write and K/V projection of the new items) is captured ONCE in two HIP graphs -- head
        and tail -- per worker group and replayed per step (``hip_graph_rollout: false`` in the config selects the eager
        path).  With ``rollout_groups`` = 2 the groups form a software pipeline: the host steps one group's environments
        while the device runs the other group's head graph.  With the fused encoder the observation rows of step t+1 are
        streamed from pinned memory into row t+1 of the staging array on a second stream while the environments still step
        (``stream_observations``), together with the workers' (episode step, slot) vector; the actions arrive in pinned host
        memory straight from the sampling kernel.
        ``forced_actions`` [W, S] (optional) replays recorded actions instead of sampling (teacher forcing for parity
        tests -- CPU and GPU RNG streams differ) on whichever path the config selects: the sampling
        kernels read them from a fixed-address table, so the captured graphs, the observation streaming and the worker-group
        pipeline run exactly as they do when sampling."""
        buf, W, S = self.buffer, self.num_workers, self.config["worker_steps"]
        main = torch.cuda.current_stream(self.device)
        use_graph = bool(self.config.get("hip_graph_rollout", True))
        episode_infos = []
        buf.begin_rollout(self._slot_dev)
        self.worker_episode_slot[:] = np.arange(W)
        self._slot_dev.copy_(self._slot_pin, non_blocking=True)
        if self._use_kv_cache:
            self._refresh_kv_cache()
        self.model.refresh_rollout_weights()       # encoder weight copies for the fused rollout convolutions
        if forced_actions is not None:
            fa = torch.as_tensor(np.asarray(forced_actions), dtype=torch.int64).reshape(W, S)
            self._forced_tab.copy_(fa.t().to(self.device))
        groups = self._groups if use_graph else [self._group_all]
        if use_graph and groups[0].graphs is None:
            self._capture_step_graph(groups)
        self._uniforms.uniform_()                # one draw per (step, worker) for the whole rollout
        for g in groups:
            g.t_dev.zero_()
            g.flag_np[0] = 0
        stream_obs = use_graph and self._stream_obs
        host_flag = use_graph and self._host_flag
        lib = etm_lib.load()
        up = self._up_stream.cuda_stream
        row_bytes = self._obs_pin[0].numel() * 4
        src_base, stage_base = self._obs_pin.data_ptr(), self._stage["obs"].data_ptr()
        ss_global = self._ss_pin.numpy()
        side_streams = [g.stream for g in groups if g.stream is not None]
        for st_ in side_streams:
            st_.wait_stream(main)                  # buffers prepared above on the main stream
        if stream_obs:
            self._up_stream.wait_stream(main)      # the staging array may still be read by the previous update

        # With a stream per group (the pipelined default) the observation rows of a group go to the device on the GROUP's stream
        # -- stream order alone puts them before the step that reads them -- and the step's window kernel reads the
        # (episode step, slot) block straight from pinned host memory: no upload of that block, no event between an upload
        # stream and the step (each of those was a few us on the critical path of every step).
        own_stream = stream_obs and all(g.stream is not None for g in groups)
        # direct observation rows (round 6; in-process environments): the front-end writes the rows of step t + 1 straight into
        # their row of the staging array in DEVICE memory (large BAR: the hipMalloc pointer is a host address) -- no pinned
        # intermediate, no copy-engine transfer (677 KB per group and step at 3x84x84: ~20 us of the step's critical path) and no
        # runtime call; etm_host_store_fence (sfence + the device's HDP flush register) sits between the rows and the launch.
        # The pinned buffer still receives the observation AFTER the last step (the bootstrap value and the next rollout's
        # observation 0 read it).  `direct_observation_rows: false`, a device without large BAR or a failed self-test: uploads.
        direct = bool(own_stream and host_flag and self._shm_env is None and self.config.get("direct_observation_rows", True)
                      and ops.host_direct_write_ok(self.device))
        if direct:
            if getattr(self, "_stage_host", None) is None:
                self._stage_host = ops.host_view(self._stage["obs"])
            ev = getattr(self, "_stage_read", None)
            if ev is not None:
                ev.synchronize()                   # the previous update's copy out of the staging array has run
        self._direct_rows = direct
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()

        def obs_stream(g):
            return g.stream.cuda_stream if own_stream else up

        # (CUDAGraph.raw_cuda_graph_exec exists in torch >= 2.8; without it the framework's replay() is used)
        direct_launch = bool(host_flag and hasattr(torch.cuda.CUDAGraph, "raw_cuda_graph_exec"))
        polite = bool(self._host_plan["polite_wait"])

        def upload_state(g):
            """(episode step, slot) of the group's workers -> where the device finds them, after the host bookkeeping of the step."""
            if not g.full:
                g.ss_np[:] = ss_global[:, g.lo:g.hi]
            if own_stream:
                return
            lib.etm_upload(g.ss_dev.data_ptr(), g.ss_pin.data_ptr(), g.ss_pin.numel() * 8, up)
            g.up_done.record(self._up_stream)

        def launch(g, t):
            """Device work of step t of group g (graph mode: two replays on the group's stream)."""
            if use_graph and direct_launch and g.graphs[1] is None and g.stream is not None and (own_stream or not stream_obs):
                # one captured graph per step on the group's own stream, nothing to wait for: hipGraphLaunch through the library,
                # without the framework's stream switches around the replay (round 4: ~6 us of every group's step on the host)
                if not g.full:
                    g.ss_np[:] = ss_global[:, g.lo:g.hi]
                if getattr(g, "graph_exec", None) is None:
                    g.graph_exec = g.graphs[0].raw_cuda_graph_exec()
                rc = lib.etm_graph_launch(g.graph_exec, g.stream.cuda_stream)
                if rc != 0:
                    etm_lib.check(rc, "etm_graph_launch")
                return
            if use_graph:
                if g.stream is not None:
                    torch.cuda.set_stream(g.stream)
                cur = g.stream if g.stream is not None else main
                if stream_obs and not own_stream:
                    cur.wait_event(g.up_done)        # observation rows and (step, slot) of step t are on the device
                elif not g.full:
                    g.ss_np[:] = ss_global[:, g.lo:g.hi]
                g.graphs[0].replay()
                if not host_flag:
                    g.act_ready.record(cur)          # actions are in pinned memory once this event completes
                if g.graphs[1] is not None:
                    g.graphs[1].replay()             # tail runs while the host steps the environments
                if g.stream is not None:
                    torch.cuda.set_stream(main)
            else:
                with torch.no_grad():
                    carry = self._rollout_step_head(g)
                    g.act_ready.record(main)
                    self._rollout_step_tail(g, carry)

        # the native driver runs this rollout iff the captured steps hand over through the segment's go words (_native_rollout, decided
        # at capture) AND the step is the single flag-hand-over graph on the group's own stream -- ONE predicate for "workers held
        # spinning", "sequence restarted" and "etm_rollout_drive called" 
        native = bool(use_graph and getattr(self, "_native_rollout", False) and host_flag and own_stream
                      and all(g.graphs[1] is None for g in groups))
        if use_graph and getattr(self, "_native_rollout", False):
            # the device's step counter restarts at 1: go = 0 on every group, acknowledged by every worker, BEFORE step 0 is launched;
            # the workers then spin (no self-parking) until the rollout is over (native driver) or park when idle (host loop: the
            # captured kernels still write the go words, the Python loop below steps the workers through the same sequence numbers)
            self._shm_env.activate(hold=native)
            self._shm_env.restart_sequence()
        if stream_obs:
            for g in groups:                       # observation 0 -> staging row 0
                lib.etm_upload(stage_base + g.lo * row_bytes, src_base + g.lo * row_bytes, g.W * row_bytes, obs_stream(g))
                upload_state(g)
        for g in groups:
            launch(g, 0)
        t_env = t_wait = t_launch = 0.0
        if native:
            try:
                t_wait, t_launch = self._drive_rollout_native(groups, episode_infos)
            finally:
                self._shm_env.park()          # whatever happened: no worker keeps spinning through the optimisation phase
        for t in (range(S) if not native else ()):
            for g in groups:
                lo, hi = g.lo, g.hi
                tw = time.perf_counter()
                if host_flag:
                    # the sampling kernel stored the actions and then step t + 1 into pinned memory: spin on the counter
                    flag, target, spins, t_wait0 = g.flag_np, t + 1, 0, time.perf_counter()
                    if polite and flag[0] != target and self._flag_wait_ema > 80e-6:
                        # not enough CPUs for a spinning trainer thread (etm/hostcpu.py): sleep through most of the wait this flag
                        # usually takes (an average of the earlier waits), spin for the rest
                        time.sleep(0.7 * self._flag_wait_ema)
                    while flag[0] != target:
                        spins += 1
                        if spins % 4096 == 0 and time.perf_counter() - t_wait0 > 30.0:
                            raise RuntimeError("rollout step did not complete within 30 s (device hang?)")
                else:
                    g.act_ready.synchronize()
                te = time.perf_counter()
                t_wait += te - tw
                if polite:
                    self._flag_wait_ema += 0.1 * ((te - tw) - self._flag_wait_ema)
                if direct and t + 1 < S:
                    _, rewards, dones, infos = g.env.step(g.acts_host, out=self._stage_host[t + 1, lo:hi])
                    lib.etm_host_store_fence(dev_index)
                elif stream_obs and t + 1 < S:
                    dst_base = stage_base + ((t + 1) * W + lo) * row_bytes
                    src_g = src_base + lo * row_bytes
                    up_g = obs_stream(g)

                    def rows_ready(a, b):
                        lib.etm_upload(dst_base + a * row_bytes, src_g + a * row_bytes, (b - a) * row_bytes, up_g)

                    _, rewards, dones, infos = g.env.step(g.acts_host, out=g.obs_np, on_rows=rows_ready)
                else:
                    _, rewards, dones, infos = g.env.step(g.acts_host, out=g.obs_np)
                t_env += time.perf_counter() - te
                self.worker_current_episode_step[lo:hi] += 1
                if dones.any():
                    for wl in np.flatnonzero(dones):
                        w = lo + int(wl)
                        self.worker_current_episode_step[w] = 0
                        episode_infos.append(infos[wl])
                        slot = buf.open_episode()                  # fresh zero memory for the next episode (upstream :208-213)
                        self.worker_episode_slot[w] = slot
                        if t < S - 1:
                            buf.memory_index_host[w, t + 1:] = slot
                if t + 1 < S:
                    tl = time.perf_counter()
                    if stream_obs:
                        upload_state(g)              # bookkeeping of this step is final: (step, slot) follow the observation rows
                    launch(g, t + 1)
                    t_launch += time.perf_counter() - tl
                    if self._chain_log is not None and g is groups[0]:
                        self._chain_log.append((tw, te, tl, time.perf_counter()))
                buf.rewards[lo:hi, t] = rewards    # (after the launch: nothing on the device waits for these)
                buf.dones[lo:hi, t] = dones
        for st_ in side_streams:
            main.wait_stream(st_)
        t_ = self.model.transformer
        for g in groups + [self._group_all]:
            if g.rf_scratch is not None and int(ops.rollout_trxl_error(g.rf_scratch).item()) != 0:
                # a team member gave up waiting for its partners (not all workgroups were resident): this rollout's data are
                # unusable.  Leave the trainer in a state that can continue: clear the error word, switch to the multi-launch
                # step for the rest of the run and drop the captured graphs so that the next rollout re-captures them.
                for gg in groups + [self._group_all]:
                    if gg.rf_scratch is not None:
                        ops.rollout_trxl_clear_error(gg.rf_scratch)
                    gg.graphs = None
                self.model.fused_rollout_block = False
                self.model._rf = self.model._rfg = None
                self._step_graph = None
                raise RuntimeError("fused rollout step: a team member timed out waiting for its partners; this rollout is void. The "
                                   "trainer has switched to the multi-launch step (fused_rollout_block: false) for the following rollouts")
        if forced_actions is not None:
            self._forced_tab.fill_(-1)
        # time-major staging -> the buffer's [W, S, ...] fields (one strided copy per field)
        self._step_dev.copy_(self._step_pin, non_blocking=True)
        self._slot_dev.copy_(self._slot_pin, non_blocking=True)
        for name, stage in self._stage.items():
            getattr(buf, name).copy_(stage.transpose(0, 1))
        if getattr(self, "_direct_rows", False):
            if getattr(self, "_stage_read", None) is None:
                self._stage_read = torch.cuda.Event()
            self._stage_read.record(main)          # the next rollout's host writes into the staging array wait for this
        last_value = self.get_last_value()
        buf.calc_advantages(last_value, self.config["gamma"], self.config["lamda"])
        self.last_update_timing.update(env_s=t_env, wait_s=t_wait, launch_s=t_launch)
        return episode_infos

    def _rollout_step_device(self, g, stream_obs=False, host_flag=False):
        """Device side of one rollout step of group ``g`` (upstream trainer.py:161-186) = head + tail."""
        carry = self._rollout_step_head(g, stream_obs, host_flag)
        self._rollout_step_tail(g, carry, stream_obs)

    def _rollout_step_head(self, g, stream_obs=False, host_flag=False):
        """Everything the ACTIONS of group ``g`` depend on: (observation / step / slot upload,) window lookup, model forward,
        sampling, staging of the step's rows, action hand-over.  Every operand has a fixed address (HIP-graph capturable).
        Returns what the tail needs (the new memory items, block-major)."""
        buf = self.buffer
        st = self._stage
        rows = None if g.full else (g.lo, g.hi)
        if stream_obs:      # the observation of this step is already in row t of the staging array (see _sample_training_data)
            obs, obs_index = st["obs"], g.t_dev
        else:
            g.obs_dev.copy_(g.obs_pin, non_blocking=True)
            obs, obs_index, rows = g.obs_dev, None, None
            g.ss_dev.copy_(g.ss_pin, non_blocking=True)      # (streamed mode: uploaded with the observation rows)
        single = len(self.action_space_shape) == 1
        mask_t, win_t = g.mask_t, g.win_t
        # window lookup + staging; the same launch records the staging row of this step for the tail (t_dev is incremented by
        # the sampling kernel) and resets the K/V cache of workers at episode step 0 (they start from the projection of an
        # empty memory)
        # streamed + pipelined mode: the (step, slot) block is read from pinned host memory (see _sample_training_data)
        zero_copy = stream_obs and g.stream is not None
        ss_src = g.ss_pin if zero_copy else g.ss_dev
        rf_ = getattr(self.model, "_rf", None) if self._use_kv_cache else None
        # (every team of the step kernel must be resident at once, and the groups' step kernels run concurrently: the workgroups
        # of ALL groups together must fit the 256 CUs -- one 512-thread workgroup per CU --, else the multi-launch path)
        n_conc = len(self._groups) if not g.full else 1
        # round 5: GRU-gated layouts in groups of <= 8 workers take the GROUP form of the step kernel (weights once per group and
        # step, 32 workgroups per launch; csrc/rollout_group.hip)
        rfg_ = getattr(self.model, "_rfg", None) if rf_ is not None else None
        g.group_kernel = bool(single and rfg_ is not None and self.config.get("rollout_group_kernel", True)
                              and ops.rollout_trxl_group_ok(rfg_, g.W, self.memory_length, self.model.hidden_size, self.action_space_shape[0])
                              and n_conc * etm_lib.load().etm_rollout_trxl_group_grid() <= 256)
        fused_step = (single and rf_ is not None and self.model.rollout_heads_fusable()
                      and (g.group_kernel or n_conc * etm_lib.load().etm_rollout_trxl_grid(g.W, rf_["H"]) <= 256))
        # the fused step kernel does the window lookup (and the cache reset of new episodes) itself: one launch fewer in the chain
        if not fused_step:
            ops.rollout_window(ss_src[0], self._mask_table, self._index_table, g.t_dev, mask_t, win_t,
                               st["memory_mask"], st["memory_indices"], t_row=g.t_row,
                               reset=(g.kv, self._kv_init) if self._use_kv_cache else None, w_off=g.lo,
                               latch=(ss_src, g.ss_latch))
        fused_policy = False
        if self._use_kv_cache:
            kv_spec = WindowSpec.from_bank(g.kv, None, win_t, None, mask_t)
            if fused_step:
                # post-LN blocks without gates: the transformer, the heads and the sampling are ONE launch -- one workgroup per
                # worker walks the whole chain as matrix-vector products over the L2-resident weights (csrc/rollout_fused.hip);
                # the step is then encoder (4 launches) + window lookup + this kernel instead of 26 dependent launches
                rf = self.model._rfg if g.group_kernel else self.model._rf
                h_bias = None
                if obs_index is not None and "hid_t" in rf:
                    # lin_hidden as K-slice partial sums on 12 x 16 workgroups; the step kernel adds slices + bias + ReLU
                    m_ = self.model
                    hh_, ww_ = m_.observation_space_shape[-2:]
                    h2_, w2_ = hh_, ww_
                    for cv in (m_.conv1, m_.conv2):                                                      # spatial size after conv1, conv2
                        h2_, w2_ = (h2_ - cv.kernel_size[0]) // cv.stride[0] + 1, (w2_ - cv.kernel_size[1]) // cv.stride[1] + 1
                    if (self.config.get("fused_conv3_hidden", True)
                            and ops.rollout_conv3_hidden_supported(m_.conv3, h2_, w2_, rf["hid_t"].shape[1])):
                        # round 4: the last encoder layer and lin_hidden's partial sums as ONE launch, one workgroup per output pixel
                        # (csrc/conv3_hidden.hip): the step graph is conv1, conv2, this, the step kernel
                        x2 = m_._encode_fused(obs, obs_index, rows, features_only="conv2")
                        if getattr(g, "h_part", None) is None:
                            g.h_part = ops.rollout_conv3_hidden(x2, m_._w3k, m_.conv3.bias, rf["hid_t"])
                        h_in = ops.rollout_conv3_hidden(x2, m_._w3k, m_.conv3.bias, rf["hid_t"], out=g.h_part)
                    else:
                        feats = m_._encode_fused(obs, obs_index, rows, features_only=True)
                        if getattr(g, "h_part", None) is None:
                            g.h_part = ops.rollout_hidden_partial(feats, rf["hid_t"])
                        h_in = ops.rollout_hidden_partial(feats, rf["hid_t"], out=g.h_part)
                    h_bias = self.model.lin_hidden.bias
                else:
                    h_in = self.model._encode(obs, obs_index, rows)
                if getattr(g, "rf_scratch", None) is None or getattr(g, "rf_scratch_kind", None) != g.group_kernel:
                    # (the two forms of the kernel lay their scratch out differently: launch counter, tags and slots belong to one form)
                    t_ = self.model.transformer
                    g.rf_scratch = ops.rollout_trxl_scratch(g.W, t_.embed_dim, t_.num_heads, t_.num_blocks, self.device, group=g.group_kernel)
                    g.rf_scratch_kind = g.group_kernel
                # ... and, after the action hand-over, the memory-bank write and the K | V projection of the new items (the tail)
                tail = None
                if self.config.get("fused_rollout_tail", True) and getattr(self, "_kv_w_blocked", None) is not None:   # (pre-LN: the kernel applies norm_kv)
                    tail = (self._kv_w_blocked, self.model.transformer._pos(), g.step_l, g.slot_l, buf.bank)
                g.tail_in_kernel = tail is not None
                ops.rollout_trxl(h_in, rf, g.kv, win_t, mask_t, g.item, self.model.policy_branches[0], self.model.value,
                                 self._uniforms, self._forced_tab, g.t_dev, g.act_dev, st["actions"], st["log_probs"], st["values"],
                                 g.rf_scratch, host_actions=g.act_pin, host_flag=g.flag_pin if host_flag else None, w_off=g.lo,
                                 tail=tail, h_bias=h_bias,
                                 window=(ss_src, self._mask_table, self._index_table, st["memory_mask"], st["memory_indices"],
                                         g.ss_latch, g.t_row, self._kv_init))
                item = g.item
                fused_policy = True
            elif single and self.model.rollout_heads_fusable():
                # hidden heads -> ONE launch for output heads, sampling, staging, t += 1; the kernel stores the actions straight
                # into the pinned host buffer (no copy launch): they are visible to the host when the step's event (or, with
                # host_flag_actions, the flag) says the launch is done
                h2, item = self.model.forward_hidden_cached(obs, kv_spec, items_out=g.item, obs_index=obs_index, raw=True,
                                                            obs_rows=rows)
                flag = host_flag
                ops.rollout_policy(h2, self.model.policy_branches[0], self.model.value, self._uniforms, self._forced_tab, g.t_dev,
                                   g.act_dev, st["actions"], st["log_probs"], st["values"],
                                   host_actions=g.act_pin, host_flag=g.flag_pin if flag else None,
                                   h_bias=self.model._b_heads, w_off=g.lo)
                fused_policy = True
            else:
                logits, value, item = self.model.forward_logits_cached(obs, kv_spec, items_out=g.item, obs_index=obs_index,
                                                                        obs_rows=rows)
        else:
            spec = WindowSpec.from_bank(buf.bank, g.slot_dev, win_t, win_t, mask_t)
            logits, value, item = self.model.forward_logits(obs, spec)
            item = item.transpose(0, 1)
        if fused_policy:
            pass
        else:
            if not g.full:
                raise RuntimeError("worker groups need the fused policy path (single-branch policy, K/V cache)")
            if single:
                # log-softmax + categorical sample (inverse CDF on pre-drawn uniforms) + log-prob + staging + t += 1: one launch
                ops.rollout_sample(logits[0], value, self._uniforms, self._forced_tab, g.t_dev, g.act_dev,
                                   st["actions"], st["log_probs"], st["values"])
                g.act_pin.copy_(g.act_dev, non_blocking=True)
            else:
                row = g.t_row.view(1)
                acts, logps = [], []
                forced_t = self._forced_tab.index_select(0, row)[0]       # one recorded action per worker, shared by the branches
                for lg in logits:
                    lsm = torch.log_softmax(lg, dim=-1)
                    a = torch.multinomial(lsm.exp(), 1).squeeze(1)
                    a = torch.where(forced_t >= 0, forced_t, a)
                    acts.append(a)
                    logps.append(lsm.gather(1, a.unsqueeze(1)).squeeze(1))
                g.act_dev.copy_(torch.stack(acts, dim=1))
                st["actions"].index_copy_(0, row, g.act_dev.unsqueeze(0))
                st["log_probs"].index_copy_(0, row, torch.stack(logps, dim=1).unsqueeze(0))
                st["values"].index_copy_(0, row, value.unsqueeze(0))
                g.t_dev.add_(1)
                g.act_pin.copy_(g.act_dev, non_blocking=True)
        if item.data_ptr() != g.item.data_ptr():
            g.item.copy_(item)
        return g.item

    def _rollout_step_tail(self, g, item, stream_obs=False):
        """What the host does NOT have to wait for before stepping the environments: memory-bank write (upstream :174),
        K/V projection of the new item into the cache, observation staging.  Runs under the host's env.step()."""
        buf, st = self.buffer, self._stage
        if getattr(g, "tail_in_kernel", False):      # etm_rollout_trxl has written the bank and cache rows itself
            if not stream_obs:
                st["obs"][:, g.lo:g.hi].index_copy_(0, g.t_row.view(1), g.obs_dev.unsqueeze(0))
            return
        item = item.transpose(0, 1)                  # block-major staging -> [Wg, blocks, D]
        buf.bank[g.slot_l, g.step_l] = item           # (step, slot) as latched by this step's head, see _make_group
        if self._use_kv_cache:
            tr = self.model.transformer
            pos = tr._pos()
            pos_rows = pos.index_select(0, g.step_l) if pos is not None else None
            g.kv[g.ids, g.step_l] = tr.project_memory(item, pos_rows, self._kv_weights)
        if not stream_obs:
            st["obs"][:, g.lo:g.hi].index_copy_(0, g.t_row.view(1), g.obs_dev.unsqueeze(0))

    def _refresh_kv_cache(self):
        """Start of a rollout: re-project every live episode's memory with the CURRENT weights (they changed in the
        last optimisation phase) into the per-worker K/V cache [W, T, blocks, 2D]; rows that are not written yet hold the
        projection of a zero item, which is also the initial state of every episode that starts during the rollout.
        (Round 6: a graph replay from its third call on, ops.ReplayAfterWarmup -- with ``hip_graph_rollout``.)"""
        r = getattr(self, "_kv_refresh_replay", None)
        if r is None:
            r = self._kv_refresh_replay = ops.ReplayAfterWarmup(self._refresh_kv_cache_now, self.device, what="_refresh_kv_cache",
                                                                enabled=bool(self.config.get("hip_graph_rollout", True)))
        r.enabled = bool(self.config.get("hip_graph_rollout", True)) and self.buffer.address_captured      # (the bank keeps its address from the first captured step on)
        r()

    def _refresh_kv_cache_now(self):
        W, T = self.num_workers, self.max_episode_length
        tr = self.model.transformer
        with torch.no_grad():
            fresh = tr.kv_projection_weights()
            if self._kv_weights is None:      # fixed-address buffers: the captured step graph reads them every replay
                self._kv_weights = tuple(t.clone() if torch.is_tensor(t) else t for t in fresh)
            else:
                for dst, src in zip(self._kv_weights, fresh):
                    if torch.is_tensor(dst):
                        dst.copy_(src)
            if self.device.type == "cuda":
                # the step kernel's tail reads the projection weights member-blocked: [blocks, P, D, 2D / P] (fixed address)
                team = etm_lib.load().etm_rollout_trxl_team(tr.num_heads)
                w = self._kv_weights[0]                                    # [blocks, D, 2D] = [Wk^T | Wv^T]
                if w.shape[2] % (2 * team) == 0:
                    # member m's block = [its D / P columns of K | its D / P columns of V]: exactly the cache columns it reads in the
                    # attention phases, so the cache rows a member reads are only ever written by that member
                    nb_, d_, d2_ = w.shape
                    wb = w.reshape(nb_, d_, 2, team, d2_ // (2 * team)).permute(0, 3, 1, 2, 4).reshape(nb_, team, d_, d2_ // team)
                    if getattr(self, "_kv_w_blocked", None) is None:
                        self._kv_w_blocked = wb.contiguous()
                    else:
                        self._kv_w_blocked.copy_(wb)
            pos = tr._pos()
            live = self.buffer.bank[:W].reshape(W * T, self.num_blocks, self.embed_dim)
            pos_all = pos.repeat(W, 1) if pos is not None else None
            self._kv_cache.copy_(tr.project_memory(live, pos_all, self._kv_weights).reshape(self._kv_cache.shape))
            zeros = torch.zeros((T, self.num_blocks, self.embed_dim), dtype=torch.float32, device=self.device)
            self._kv_init.copy_(tr.project_memory(zeros, pos, self._kv_weights))

    def _capture_step_graph(self, groups):
        """Warm the step of every worker group up on a side stream (library handles, MIOpen find, GEMM tuning, allocator),
        then capture it as TWO graphs per group: the head (ends with the action hand-over) and the tail (bank / cache /
        staging writes)."""
        with torch.no_grad():
            self._stream_obs = bool(self.config.get("stream_observations", True) and self._use_kv_cache
                                    and self.model._fused_encoder_ok(self._obs_dev))
            fusable = self._use_kv_cache and len(self.action_space_shape) == 1 and self.model.rollout_heads_fusable()
            # host_flag_actions (default on): the sampling kernel stores the actions and then the step counter into pinned memory
            # and the host spins on the counter -- no event between the action hand-over and the rest of the step, so a step of a
            # group is ONE captured graph (one launch) instead of head + event + tail (measured: 287 -> 279 us per step)
            self._host_flag = bool(self.config.get("host_flag_actions", True) and fusable)
        if len(groups) > 1 and not fusable:
            raise RuntimeError("rollout_groups > 1 needs a single-branch policy and the K/V cache (set rollout_groups: 1)")
        so, hf = self._stream_obs, self._host_flag
        # native rollout driver (worker_processes): needs the flag hand-over, streamed observations on the groups' own streams and
        # the (step, slot) block read in place -- then the sampling kernels write the step's sequence number into the SEGMENT's go
        # words (the workers spin on them) instead of a private pinned word (decided here: the address is captured below)
        self._native_rollout = bool(self._shm_env is not None and so and hf and all(g.stream is not None for g in groups)
                                    and hasattr(torch.cuda.CUDAGraph, "raw_cuda_graph_exec"))
        if self._native_rollout:
            for gi, g in enumerate(groups):
                g.flag_pin = torch.from_numpy(self._shm_env.v["go"][gi, 0:1])
                g.flag_np = g.flag_pin.numpy()
        # the warm-up executions below write the CURRENT step's memory item (and its K/V projection) into the bank / cache rows
        # (slot, step) of every worker.  A worker at episode step 0 attends over a fully masked window -- uniform weights over
        # ALL L rows, row 0 included (upstream quirk, transformer.py:66-68 with an all-zero mask row) -- so a row 0 left behind by
        # the warm-up would leak into the real step 0.  Keep the rows as they were.
        with torch.no_grad():
            ss = torch.from_numpy(self._ss_pin.numpy().copy()).to(self.device)
            all_ids = torch.arange(self.num_workers, device=self.device)
            saved_bank = self.buffer.bank[ss[1], ss[0]].clone()
            saved_kv = self._kv_cache[all_ids, ss[0]].clone()
        for g in groups:
            g.t_dev.zero_()
            # warm-up and capture run on the stream the group's graphs are replayed on: the BLAS workspace of the library
            # GEMMs is keyed by (handle, stream), so two groups whose graphs replay concurrently must not have been captured on
            # one shared capture stream (torch's default) -- split-K solutions would accumulate in the same scratch
            side = g.stream if g.stream is not None else torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side), torch.no_grad():
                for i in range(3):
                    self._rollout_step_device(g, so, hf)
                    side.synchronize()
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            pool = torch.cuda.graph_pool_handle()
            head, tail = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            # thread_local: only this thread's calls are checked during capture (RCCL's watchdog thread may query events)
            if hf:
                # the host learns about the actions from the flag the sampling kernel writes, not from an event between head and
                # tail: the whole step is ONE graph (one launch per group and step on the host instead of two)
                with torch.no_grad(), torch.cuda.graph(head, pool=pool, stream=g.stream, capture_error_mode="thread_local"):
                    item = self._rollout_step_head(g, so, hf)
                    self._rollout_step_tail(g, item, so)
                g.graphs = (head, None)
                g.graph_exec = None
            else:
                with torch.no_grad(), torch.cuda.graph(head, pool=pool, stream=g.stream, capture_error_mode="thread_local"):
                    self._rollout_step_head(g, so, hf)
                if getattr(g, "tail_in_kernel", False) and so:
                    tail = None                       # nothing left to launch after the hand-over
                else:
                    with torch.no_grad(), torch.cuda.graph(tail, pool=pool, stream=g.stream, capture_error_mode="thread_local"):
                        self._rollout_step_tail(g, g.item, so)
                g.graphs = (head, tail)
                g.graph_exec = None
            g.t_dev.zero_()
        with torch.no_grad():
            torch.cuda.synchronize(self.device)
            self.buffer.bank[ss[1], ss[0]] = saved_bank
            self._kv_cache[all_ids, ss[0]] = saved_kv
        self.buffer.address_captured = True
        ops.freeze_workspaces(self.device)
        self._step_graph = groups[0].graphs

    def get_last_value(self):
        """Value of the observation after the last step (bootstrap for GAE), with upstream's window rule:
        rows [clip(step - L, 0), clip(step, L)) and positional indices of the last stored step (trainer.py:230-236)."""
        L = self.memory_length
        step = torch.from_numpy(self.worker_current_episode_step.copy())
        start = torch.clamp(step - L, min=0)
        rows_host = start.unsqueeze(1) + torch.arange(L, dtype=torch.int64).unsqueeze(0)
        lv = getattr(self, "_lv", None)
        if lv is None:      # fixed-address operands: the forward pass below is replayed from a captured graph (round 6)
            from types import SimpleNamespace
            lv = self._lv = SimpleNamespace(rows=torch.empty((self.num_workers, L), dtype=torch.int64, device=self.device),
                                            obs=torch.empty_like(self._obs_dev), out=torch.empty(self.num_workers, dtype=torch.float32, device=self.device),
                                            graph=None, calls=0, failed=False)
        lv.rows.copy_(rows_host, non_blocking=False)
        lv.obs.copy_(self._obs_pin, non_blocking=True)

        def body():
            mask = self._mask_table[torch.clamp(self._step_dev, 0, L - 1)]
            spec = WindowSpec.from_bank(self.buffer.bank, self._slot_dev, lv.rows, self.buffer.memory_indices[:, -1], mask)
            _, last_value, _ = self.model.forward_logits(lv.obs, spec, want_items=False)
            lv.out.copy_(last_value)

        # ~100 small launches on 32 samples: 1.4 ms eager, once per update; as a graph replay it is the kernels' own time.  The first
        # two calls run eagerly (library handles, GEMM tuning), any capture failure keeps the eager path for good.
        use_graph = bool(self.config.get("hip_graph_rollout", True)) and self.buffer.address_captured and not lv.failed
        with torch.no_grad():
            lv.calls += 1
            if use_graph and lv.graph is None and lv.calls > 2:
                try:
                    torch.cuda.synchronize(self.device)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        body()
                    lv.graph = g
                except Exception as exc:       # noqa: BLE001
                    lv.failed = True
                    torch.cuda.synchronize(self.device)
                    print(f"[etm] get_last_value stays eager (capture failed: {exc!r})", file=sys.stderr, flush=True)
            if lv.graph is not None:
                lv.graph.replay()
            else:
                body()
        return lv.out

    # ------------------------------------------------------------------ optimisation
    def _train_epochs(self, learning_rate: float, clip_range: float, beta: float, perms=None):
        """``epochs`` passes over shuffled minibatches.  Returns (list of 6-stat rows, {grad key: [norms]})."""
        stats, norms = [], []
        monitor = self.config.get("monitor_gradients", True)
        # sinusoidal positions: add them to the (read-only) episode bank once per update instead of once per window row,
        # block, minibatch and epoch inside the kernels (bit-identical sums; halves the kernels' window-row loads)
        with torch.no_grad():
            self._bank_pos = self._bank_with_positions()
            self._obs_train = self._observations_channels_last()
        mbs = self.buffer.batch_size // self.buffer.n_mini_batches
        # sort_minibatch (default on): the samples of a minibatch in ascending flat (worker, step) order.  The minibatch is the
        # same SET (every loss term is a mean over it; only summation order changes), but neighbouring samples then share most of
        # their window rows, which the window pass -- every XCD handles a contiguous chunk of the samples -- turns into L2 hits
        # (measured at config 3: 37.9 -> 29.9 us per pass).  One sort per epoch covers all of its minibatches.
        sort_mb = bool(self.config.get("sort_minibatch", True))
        for epoch in range(self.config["epochs"]):
            if perms is None:
                perm = torch.randperm(self.buffer.batch_size, device=self.device)
            else:
                perm = torch.as_tensor(perms[epoch], device=self.device, dtype=torch.long)
            if sort_mb and perm.numel() % mbs == 0:
                perm = perm.view(-1, mbs).sort(dim=1).values.reshape(-1)
            self._epoch_stats3 = None
            if self.dp is not None and self.dp.active and perm.numel() % mbs == 0:
                # data parallel: the global-minibatch advantage statistics of ALL minibatches of the epoch from one all-gather
                # (they depend only on the advantages and the permutation, not on the weights)
                adv = self.buffer.samples_flat["advantages"].index_select(0, perm).view(-1, mbs)
                local = torch.stack([ops.adv_stats(adv[i]) for i in range(adv.shape[0])])
                self._epoch_stats3 = self.dp.merge_adv_stats(local)
            for start in range(0, self.buffer.batch_size, mbs):
                idx = perm[start: start + mbs]
                self._mb_stats3 = self._epoch_stats3[start // mbs] if self._epoch_stats3 is not None else None
                if self._use_train_graph and idx.numel() == mbs:
                    st_row, norm_row = self._train_step_graph(idx, learning_rate, clip_range, beta, monitor)
                    stats.append(st_row)
                    if monitor:
                        norms.append(norm_row)
                    continue
                mini_batch = self.buffer.gather(idx)
                stats.append(self._train_mini_batch(mini_batch, learning_rate, clip_range, beta))
                if monitor:
                    norms.append(self._grad_group_norms())
        train_info = torch.stack(stats).cpu().numpy()          # the only host sync of the optimisation phase
        self._bank_pos = self._row_stats = None
        self._mb_stats3 = self._epoch_stats3 = None
        grad_info = {}
        if norms:
            allnorms = torch.stack(norms).cpu().numpy()
            grad_info = {k: allnorms[:, i].tolist() for i, k in enumerate(self._grad_keys)}
        return [row for row in train_info], grad_info

    def _train_mini_batch(self, samples: dict, learning_rate: float, clip_range: float, beta: float):
        """One optimiser step on one minibatch.  Returns a device tensor [policy, value, loss, entropy, kl, clip_frac]."""
        ep = samples.get("memory_index")  # None: upstream layout, ``memories`` already gathered per sample
        bank_pos = getattr(self, "_bank_pos", None)
        if bank_pos is not None and ep is not None and samples["memories"].data_ptr() == self.buffer.bank.data_ptr():
            spec = WindowSpec.from_bank(bank_pos, ep, samples["memory_indices"], None, samples["memory_mask"])
            spec.pos_included = True
            spec.row_stats = getattr(self, "_row_stats", None)
        else:
            spec = WindowSpec.from_bank(samples["memories"], ep, samples["memory_indices"], samples["memory_indices"],
                                        samples["memory_mask"])
            if ep is not None and self.model.transformer.pos_kind == "" and samples["memories"].data_ptr() == self.buffer.bank.data_ptr():
                spec.row_stats = getattr(self, "_row_stats", None)       # (as the captured bodies do: same work on both paths)
        stats3 = getattr(self, "_mb_stats3", None)
        if stats3 is None:
            stats3 = ops.adv_stats(samples["advantages"])
            if self.dp is not None:
                stats3 = self.dp.merge_adv_stats(stats3)
        loss, stats = self._loss_from(samples["obs"], spec, samples, clip_range, beta, stats3, dyn=None)
        self._set_lr(learning_rate)
        self._backward_into_arena(loss)
        if self.dp is not None:
            self.dp.all_reduce_grads(average=False)       # the sum; the 1 / world rides in the clip coefficient below
        # global-norm clipping (the rule of torch.nn.utils.clip_grad_norm_, upstream :311) + AdamW on the flat arenas: 2 launches
        self.optimizer.step(self.config["max_grad_norm"], grad_scale=self._grad_scale())
        return stats

    def _loss_from(self, obs, spec, mb, clip_range, beta, stats3, dyn="device"):
        """Model forward + PPO loss of one minibatch (trainer.py:268-304): -> (loss, stats[6]).  Single-branch policies whose hidden size
        fits take the fused hidden-heads + loss kernel (``fused_heads_loss``, default on); others the separate heads and loss."""
        dyn = getattr(self, "_dyn", None) if dyn == "device" else dyn
        m = self.model
        if self.config.get("fused_heads_loss", True) and len(m.policy_branches) == 1:
            h, _ = m.forward_state(obs, spec)
            if ops.heads_loss_supported(h, m.lin_policy, m.policy_branches[0]):
                return ops.heads_ppo_loss(h, m.lin_policy, m.lin_value, m.policy_branches[0], m.value, mb["actions"], mb["log_probs"],
                                          mb["advantages"], mb["values"], clip_range, self.config["value_loss_coefficient"], beta, stats3, dyn=dyn,
                                          unit_grad=True)       # (both callers run loss.backward() on this loss)
            h_policy = ops.linear_relu(m.lin_policy, h)
            h_value = ops.linear_relu(m.lin_value, h)
            logits, value = [m.policy_branches[0](h_policy)], m.value(h_value).reshape(-1)
        else:
            logits, value, _ = m.forward_logits(obs, spec, want_items=False)
        return ops.ppo_loss(logits, value, mb["actions"], mb["log_probs"], mb["advantages"], mb["values"], clip_range,
                            self.config["value_loss_coefficient"], beta, stats3, dyn=dyn)

    def _dw_destinations(self):
        """{parameter data_ptr: its gradient view in the flat arena}: the [out, in] views of the 2-D parameters for the grouped weight-
        gradient launch (``grouped_dw_train: false`` in the config: none, every layer multiplies its own weight gradient) and the
        1-D views (LayerNorm weights / biases, linear / convolution biases) and the convolution weights' views for the grouped
        column-sum / slice reductions (``grouped_colsum_train``)."""
        if getattr(self, "_dw_dest", None) is None:
            dw, cs = self.config.get("grouped_dw_train", True), self.config.get("grouped_colsum_train", True)
            self._dw_dest = {p.data_ptr(): v for p, v in zip(self.params, self._grad_views)
                             if (p.dim() == 2 and dw) or (p.dim() == 1 and cs) or (p.dim() == 4 and cs)}
        return self._dw_dest

    def _unit_gradient(self, loss):
        """d loss / d loss = 1 from a cached tensor (autograd would fill a fresh one every step: one launch)."""
        one = getattr(self, "_one", None)
        if one is None or one.device != loss.device or one.shape != loss.shape:
            one = self._one = torch.ones_like(loss)
        return one

    def _grad_scale(self):
        return self.dp.grad_scale if self.dp is not None else 1.0

    def _set_lr(self, learning_rate: float):
        self.optimizer.set_lr(learning_rate)

    def _bank_with_positions(self):
        """Episode bank with the sinusoidal positional rows pre-added, in a buffer that keeps its address (the captured
        training step reads it); None when the positional encoding is not the fixed sinusoid."""
        tr = self.model.transformer
        self._row_stats = None
        if tr.pos_kind not in ("relative", ""):
            return None
        mem = self.buffer.memories
        if tr.pos_kind == "relative":
            if getattr(self, "_bank_pos_buf", None) is None or self._bank_pos_buf.shape != self.buffer.bank.shape:
                self._bank_pos_buf = torch.empty_like(self.buffer.bank)
            out = self._bank_pos_buf[: mem.shape[0]]
            torch.add(mem, tr._pos_table[None, : mem.shape[1], None, :], out=out)
        self._row_stats = self._bank_row_stats(self._bank_pos_buf if tr.pos_kind == "relative" else self.buffer.bank, mem.shape[0])
        return self._bank_pos_buf if tr.pos_kind == "relative" else None

    def _bank_row_stats(self, bank, used):
        """Pre-LN models (norm_kv, transformer.py:128-131): LayerNorm statistics of every used row of the (position-augmented) bank,
        once per update, in a buffer [blocks, slots, T, 2] that keeps its address -- the window passes gather their per-window
        statistics from it instead of re-reading every window row (ops.WindowSpec.row_stats).  None: not applicable."""
        blk = self.model.transformer.transformer_blocks[0]
        # bank_row_stats (default on; False: every window pass computes the statistics of its own rows).  Measured at config 5:
        # optimisation phase 0.107 -> 0.102 s per update.  (It was opt-in for part of round 5: one teacher-forced flow diverged with it.
        # The cause was elsewhere -- torch's column sum in the norm_kv gradient pass's fallback inside the captured step,
        # profiles/r05/graph_reduce_hazard.txt; this option merely changed the graph enough to expose it.)
        if blk.layer_norm != "pre" or not self.buffer.block_major or not self.config.get("bank_row_stats", True):
            return None
        E, T, nb, D = bank.shape
        if D % 128 != 0 or D > 1024:
            return None
        if getattr(self, "_row_stats_buf", None) is None or self._row_stats_buf.shape != (nb, E, T, 2):
            self._row_stats_buf = torch.zeros((nb, E, T, 2), dtype=torch.float32, device=self.device)
        lib = etm_lib.load()
        st = torch.cuda.current_stream(self.device).cuda_stream
        for b in range(nb):           # the used slots of block b are one contiguous run of rows in a block-major bank
            rows = bank[:used, :, b, :]
            etm_lib.check(lib.etm_ln_row_stats(rows.data_ptr(), float(blk.norm_kv.eps), self._row_stats_buf[b].data_ptr(), used * T, D, st),
                          "etm_ln_row_stats")
        return self._row_stats_buf

    def _observations_channels_last(self):
        """Visual observations of the whole buffer in NHWC memory order, converted ONCE per update into a fixed-address buffer
        (the library convolutions of the optimisation phase run on channels_last activations; converting every gathered
        minibatch costs a 173 MB copy per minibatch at config 3).  Returns the flat [W*S, H, W, C] buffer or None."""
        obs = self.buffer.samples_flat["obs"]
        if obs.dim() != 4 or not getattr(self.model, "channels_last", False):
            return None
        n, c, h, w = obs.shape
        if getattr(self, "_obs_nhwc_buf", None) is None or self._obs_nhwc_buf.shape != (n, h, w, c):
            self._obs_nhwc_buf = torch.empty((n, h, w, c), dtype=torch.float32, device=self.device)
        self._obs_nhwc_buf.copy_(obs.permute(0, 2, 3, 1))
        return self._obs_nhwc_buf

    def _train_body_a(self, idx, clip_range, beta, stats3=None):
        """First half of one optimiser step on the minibatch ``idx`` (device int64 [mbs], fixed address): gather, forward,
        loss, backward, gradients packed into the flat bucket.  ``stats3``: (count, mean, M2) of the GLOBAL minibatch's
        advantages (data-parallel runs merge them over ranks before this graph); None: computed here.  Returns stats[6]."""
        buf = self.buffer
        skip = ("obs",) if self._obs_train is not None else ()
        keys = [k for k in buf.samples_flat if k not in skip]
        mb = dict(zip(keys, ops.gather_rows([buf.samples_flat[k] for k in keys], idx)))    # one launch for the small fields
        if self._bank_pos is not None:
            spec = WindowSpec.from_bank(self._bank_pos_buf, mb["memory_index"], mb["memory_indices"], None, mb["memory_mask"])
            spec.pos_included = True
            spec.row_stats = getattr(self, "_row_stats", None)
        else:
            spec = WindowSpec.from_bank(buf.bank, mb["memory_index"], mb["memory_indices"], mb["memory_indices"], mb["memory_mask"])
            if self.model.transformer.pos_kind == "":
                spec.row_stats = getattr(self, "_row_stats", None)
        obs = mb.get("obs")
        if self._obs_train is not None:     # NHWC rows of the minibatch: gathered by the first encoder layer itself
            obs = IndexedObservations(self._obs_train, idx)
        if stats3 is None:
            stats3 = ops.adv_stats(mb["advantages"])
        loss, stats = self._loss_from(obs, spec, mb, clip_range, beta, stats3)
        self._backward_into_arena(loss)
        return stats

    def _backward_into_arena(self, loss):
        """``loss.backward()`` with every gradient ending up in its view of the flat gradient arena."""
        # backward() hands every parameter its gradient tensor (no accumulate launch while .grad is None); ONE multi-tensor copy
        # packs them into the flat bucket that the all-reduce, clipping and the fused AdamW read -- ~50 launches fewer per step
        # than accumulating into the zeroed bucket (the python-side re-aliasing below costs nothing under graph replay)
        for p in self.params:
            p.grad = None
        # the weight gradients of the dense layers (dW = dy^T x, a contraction over the minibatch with a small output) are collected
        # during backward and computed by ONE grouped launch straight into their arena views (csrc/grouped_dw.hip)
        with ops.DeferredDw(self._dw_destinations()) as dw:
            loss.backward(self._unit_gradient(loss))
        views, grads = [], []
        for p, v in zip(self.params, self._grad_views):
            if p.data_ptr() in dw.written:
                if p.grad is not None:               # a second use of the parameter that the collector refused: add it to the arena view
                    v.add_(p.grad)
                continue                             # already in the arena
            views.append(v)
            grads.append(p.grad if p.grad is not None else torch.zeros_like(v))   # None: parameter outside this graph (unused head)
        torch._foreach_copy_(views, grads)
        for p, v in zip(self.params, self._grad_views):
            p.grad = v

    def minibatch_gradients(self, idx, clip_range: float, beta: float) -> dict:
        """Un-clipped gradient of the PPO loss (trainer.py:276-310) on the minibatch ``idx`` (flat sample indices of the prepared
        buffer) under the current weights: {parameter name: tensor}.  No optimiser step, no schedule side effects -- for parity
        tests and diagnostics (the gradient is what `loss.backward()` leaves in `.grad` before upstream's clip_grad_norm_)."""
        idx = torch.as_tensor(np.asarray(idx.cpu() if torch.is_tensor(idx) else idx), device=self.device, dtype=torch.long)
        if self.config.get("sort_minibatch", True):
            idx = idx.sort().values
        with torch.no_grad():
            self._bank_pos = self._bank_with_positions()
            self._obs_train = self._observations_channels_last()
        if self._use_train_graph:
            self._dyn.copy_(torch.tensor([clip_range, beta], dtype=torch.float64))
            self._sched_host[1:] = [clip_range, beta]
        stats3 = None
        if self.dp is not None:
            stats3 = self.dp.merge_adv_stats(ops.adv_stats(self.buffer.samples_flat["advantages"].index_select(0, idx)))
        self._train_body_a(idx, clip_range, beta, stats3)
        grads = {n: p.grad.detach().clone() for n, p in self.model.named_parameters() if p.requires_grad}
        self._bank_pos = self._row_stats = None
        return grads

    def _train_body_b(self, monitor):
        """Second half: global-norm clipping (same rule as torch.nn.utils.clip_grad_norm_, upstream :311) on the flat bucket,
        fused AdamW, monitored gradient norms."""
        self.optimizer.step(self.config["max_grad_norm"], grad_scale=self._grad_scale())
        return self._grad_group_norms() if monitor else None

    def _train_step_graph(self, idx, learning_rate, clip_range, beta, monitor):
        """Minibatch step through captured graphs (two eager warm-up steps first).  Single GPU: one graph.  Data parallel: the
        merged advantage statistics are computed eagerly first (one all-gather per epoch), then ONE graph that holds the packed
        gradients' RCCL all-reduce between the backward pass and clip + AdamW (round 6; upstream insertion point trainer.py:310-311)
        when the library collective's captured form passed its self-test on every rank (etm/dist.py: graph_collective_ok) --
        otherwise graph A, the all-reduce as a host call, graph B.  Returns (stats[6], norms or None)."""
        dp = self.dp
        if getattr(self, "_tg_idx", None) is None:
            self._tg_idx = torch.empty_like(idx)
            self._tg_stats3 = torch.zeros(3, dtype=torch.float32, device=self.device) if dp is not None else None
        self._tg_idx.copy_(idx)
        self._set_lr(learning_rate)
        if self._sched_host[1] != clip_range or self._sched_host[2] != beta:
            self._dyn.copy_(torch.tensor([clip_range, beta], dtype=torch.float64))
            self._sched_host[1:] = [clip_range, beta]
        if dp is not None:
            if getattr(self, "_mb_stats3", None) is not None:
                self._tg_stats3.copy_(self._mb_stats3)
            else:
                adv = self.buffer.samples_flat["advantages"].index_select(0, self._tg_idx)
                self._tg_stats3.copy_(dp.merge_adv_stats(ops.adv_stats(adv)))
        self._mb_counter += 1
        sample_eager = self.profile_sample_every and self._mb_counter % self.profile_sample_every == 0
        key = (monitor, self._bank_pos is not None)
        overlap = self._dp_overlap_ready()
        one_graph = bool(dp is not None and self.config.get("dp_graph_collective", True) and dp.graph_collective_ok()
                         and not getattr(self, "_dp_one_graph_failed", False))
        if (self._train_graph is None and self._train_warm < 2) or sample_eager:
            self._train_warm += 1
            if overlap:
                st, dfe = self._train_body_a1(self._tg_idx, clip_range, beta, self._tg_stats3)
                self._allreduce_rest_async()
                self._train_body_a2(dfe)
                self._allreduce_conv_and_join()
            else:
                st = self._train_body_a(self._tg_idx, clip_range, beta, self._tg_stats3)
                if dp is not None:
                    dp.all_reduce_grads(average=False)
            nm = self._train_body_b(monitor)
            return st.clone(), (nm.clone() if nm is not None else None)
        if self._train_graph is None or self._tg_key != key:
            torch.cuda.synchronize(self.device)
            ga, gb, ga2 = self._capture_one_graph_dp_step(clip_range, beta, monitor, overlap) if one_graph else None, None, None
            one_graph = ga is not None
            if dp is not None and dp.world > 1 and self.config.get("dp_graph_collective", True) and dp.graph_collective_ok():
                # every rank must replay the same form: a rank whose capture failed takes everybody to the three-call step
                if not dp.agree(one_graph):
                    ga, one_graph, self._dp_one_graph_failed = None, False, True
            if ga is None:
                ga = torch.cuda.CUDAGraph()
                # (one memory pool for the pieces of the overlapped step: part 2 reads tensors part 1 allocated)
                pool = torch.cuda.graph_pool_handle() if overlap else None
                with torch.cuda.graph(ga, pool=pool, capture_error_mode="thread_local"):
                    if overlap:
                        self._tg_stats, self._tg_dfeats = self._train_body_a1(self._tg_idx, clip_range, beta, self._tg_stats3)
                    else:
                        self._tg_stats = self._train_body_a(self._tg_idx, clip_range, beta, self._tg_stats3)
                    if dp is None:
                        self._tg_norms = self._train_body_b(monitor)
                if overlap:
                    ga2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(ga2, pool=pool, capture_error_mode="thread_local"):
                        self._train_body_a2(self._tg_dfeats)
                if dp is not None:
                    gb = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gb, pool=pool, capture_error_mode="thread_local"):
                        self._tg_norms = self._train_body_b(monitor)
            self._train_graph, self._tg_key, self._train_graph_a2 = (ga, gb), key, ga2
            self._dp_one_graph = one_graph
            self.buffer.address_captured = True
            ops.freeze_workspaces(self.device)
        ga, gb = self._train_graph
        ga.replay()
        if gb is not None:
            if getattr(self, "_train_graph_a2", None) is not None:
                self._allreduce_rest_async()                # heads + transformer + lin_hidden slice on the side stream ...
                self._train_graph_a2.replay()               # ... under the encoder's backward pass
                self._allreduce_conv_and_join()
            else:
                dp.all_reduce_grads(average=False)
            gb.replay()
        return self._tg_stats.clone(), (self._tg_norms.clone() if monitor else None)

    def close(self, exit_process: bool = False) -> None:
        """Releases environments and the summary writer (upstream also ``exit(0)``s; opt in with exit_process)."""
        if getattr(self, "_shm_registered", None):
            try:
                torch.cuda.synchronize(self.device)
                etm_lib.load().etm_host_unregister(self._shm_registered)
            except Exception:
                pass
            self._shm_registered = None
        for closer in (self.env.close, self.writer.close):
            try:
                closer()
            except Exception:
                pass
        if exit_process:
            time.sleep(1.0)
            raise SystemExit(0)
