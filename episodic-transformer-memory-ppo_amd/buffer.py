"""Rollout buffer + whole-episode memory bank, resident in HBM.

API of upstream buffer.py (constructor, fields, prepare_batch_dict / mini_batch_generator / calc_advantages), with
the storage redesigned for the MI355X path:

* every sample tensor lives on the device; ``rewards`` / ``dones`` / ``memory_index`` are produced by the host-side
  environment loop, so they are pinned host arrays that are uploaded once per update;
* ``memories`` is not a python list of per-episode tensors (upstream buffer.py:40, trainer.py:154,205-213) but the
  first ``num_episodes`` slots of one preallocated bank [capacity, T, blocks, D]: a live episode writes its items
  straight into its slot, finishing an episode just opens a new (zero) slot -- no clone, no ``torch.stack``;
* minibatches carry *indices* (``memory_index`` [N] into the bank + ``memory_indices`` [N, L]); the attention kernel
  gathers the window rows in place, so upstream's 906 MB ``memories[memory_index[idx]]`` copy (buffer.py:90) and the
  [N, L, blocks, D] gather (trainer.py:271) do not exist.  ``materialize=True`` reproduces upstream's minibatch
  layout for API users that want it.
* GAE (buffer.py:95-113) is the ``etm_gae`` kernel, bit-identical to the upstream loop.
"""
import numpy as np
import torch

from etm import ops


class Buffer:
    def __init__(self, config: dict, observation_space, action_space_shape: tuple, max_episode_length: int,
                 device: torch.device) -> None:
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Buffer is HBM-resident: it needs the MI355X (HIP) device; there is no CPU path in this build")
        self.n_workers = config["n_workers"]
        self.worker_steps = config["worker_steps"]
        self.n_mini_batches = config["n_mini_batch"]
        self.batch_size = self.n_workers * self.worker_steps
        self.mini_batch_size = self.batch_size // self.n_mini_batches
        self.max_episode_length = max_episode_length
        t = config["transformer"]
        self.memory_length, self.num_blocks, self.embed_dim = t["memory_length"], t["num_blocks"], t["embed_dim"]
        W, S, L, dev = self.n_workers, self.worker_steps, self.memory_length, self.device
        B = len(action_space_shape)

        pin = lambda shape, dtype: torch.zeros(shape, dtype=dtype).pin_memory()
        self._rewards_host = pin((W, S), torch.float32)
        self._dones_host = pin((W, S), torch.bool)
        self._memory_index_host = pin((W, S), torch.int64)
        self.rewards = self._rewards_host.numpy()            # written by the env loop
        self.dones = self._dones_host.numpy()
        self.memory_index_host = self._memory_index_host.numpy()
        self.rewards_dev = torch.zeros((W, S), dtype=torch.float32, device=dev)
        self.dones_dev = torch.zeros((W, S), dtype=torch.bool, device=dev)

        self.actions = torch.zeros((W, S, B), dtype=torch.long, device=dev)
        self.obs = torch.zeros((W, S) + tuple(observation_space.shape), dtype=torch.float32, device=dev)
        self.log_probs = torch.zeros((W, S, B), dtype=torch.float32, device=dev)
        self.values = torch.zeros((W, S), dtype=torch.float32, device=dev)
        self.advantages = torch.zeros((W, S), dtype=torch.float32, device=dev)
        self.memory_mask = torch.zeros((W, S, L), dtype=torch.bool, device=dev)
        self.memory_index = torch.zeros((W, S), dtype=torch.long, device=dev)
        self.memory_indices = torch.zeros((W, S, L), dtype=torch.long, device=dev)

        # whole-episode memory bank: slot e holds one episode's [T, blocks, D] items
        # default capacity = worst case (every step ends an episode): the bank then never reallocates, which keeps its
        # address stable for the captured rollout graph; 288 GB of HBM make this affordable (7.3 GB at BASELINE config 3)
        cap = config.get("episode_bank_capacity", W + self.batch_size)
        # BLOCK-MAJOR in memory (round 4): [blocks][slots][T][D] -- the L window rows of a (sample, block) are one
        # contiguous run of L * D floats (upstream's [slots, T, blocks, D] order interleaves the blocks: a row every blocks * D floats).
        # ``bank`` keeps upstream's logical shape [slots, T, blocks, D] as a VIEW; every kernel addresses it through its strides
        # (etm/ops.py:WindowSpec.from_bank, the tail of etm_rollout_trxl).  (Round 3's interleaved order was an option until round 6:
        # window passes 8 % slower at config 3)
        self.block_major = True
        self.bank = self._new_bank(cap)
        self.num_episodes = W
        self.address_captured = False      # set by the trainer once a captured graph reads / writes the bank
        self.samples_flat = None

    def _new_bank(self, slots: int) -> torch.Tensor:
        shape = (slots, self.max_episode_length, self.num_blocks, self.embed_dim)
        store = torch.zeros((self.num_blocks, slots, self.max_episode_length, self.embed_dim), dtype=torch.float32, device=self.device)
        return store.permute(1, 2, 0, 3)

    # ------------------------------------------------------------------ episode bank
    @property
    def memories(self):
        """[E, T, blocks, D] -- the episodes referenced by ``memory_index`` (upstream: stacked list, buffer.py:65)."""
        return self.bank[: self.num_episodes]

    def begin_rollout(self, live_slots: torch.Tensor):
        """Start of an update: live episodes move to slots 0..W-1, every other used slot is cleared."""
        # the pinned bookkeeping arrays (memory_index, rewards, done flags) are the sources of ASYNCHRONOUS uploads of the previous
        # update (prepare_batch_dict, calc_advantages): the host must not rewrite them before those copies have run.  With an
        # optimisation phase in between they long have; a caller that samples again at once (tests, tools) runs ahead of the device
        # since round 6 (get_last_value is one graph launch instead of ~100 eager ones) and would hand the copy the NEW contents.
        ev = getattr(self, "_host_arrays_uploaded", None)
        if ev is not None:
            ev.synchronize()
        W = self.n_workers
        live = self.bank.index_select(0, live_slots)
        self.bank[:W].copy_(live)
        if self.num_episodes > W:
            self.bank[W: self.num_episodes].zero_()
        self.num_episodes = W
        self.memory_index_host[:] = np.arange(W, dtype=np.int64)[:, None]

    def _mark_host_arrays_uploaded(self):
        if self.device.type == "cuda":
            if getattr(self, "_host_arrays_uploaded", None) is None:
                self._host_arrays_uploaded = torch.cuda.Event()
            self._host_arrays_uploaded.record(torch.cuda.current_stream(self.device))

    def open_episode(self) -> int:
        """Reserve the next (zero-filled) slot and return its index; grows the bank if it is full."""
        if self.num_episodes == self.bank.shape[0]:
            if self.address_captured:
                raise RuntimeError(f"episode bank is full ({self.bank.shape[0]} slots) and captured HIP graphs hold its address: raise "
                                   "episode_bank_capacity (default n_workers * (worker_steps + 1) never fills) or disable "
                                   "hip_graph_rollout / hip_graph_train")
            grown = self._new_bank(2 * self.bank.shape[0])
            grown[: self.bank.shape[0]].copy_(self.bank)
            self.bank = grown
        slot = self.num_episodes
        self.num_episodes += 1
        return slot

    # ------------------------------------------------------------------ upstream API
    def prepare_batch_dict(self) -> None:
        """Upload the host-side bookkeeping and expose the samples flattened to [W*S, ...] (W-major, views)."""
        self.memory_index.copy_(self._memory_index_host, non_blocking=True)
        self._mark_host_arrays_uploaded()
        samples = {
            "actions": self.actions, "values": self.values, "log_probs": self.log_probs, "advantages": self.advantages,
            "obs": self.obs, "memory_mask": self.memory_mask, "memory_index": self.memory_index,
            "memory_indices": self.memory_indices,
        }
        self.samples_flat = {k: v.reshape(v.shape[0] * v.shape[1], *v.shape[2:]) for k, v in samples.items()}

    def mini_batch_generator(self, indices: torch.Tensor = None, materialize: bool = False):
        """Yield shuffled minibatches.  ``indices``: optional explicit permutation (tests / teacher forcing).

        Each dict has the upstream keys ``actions, values, log_probs, advantages, obs, memory_mask, memory_indices``
        plus ``memory_index`` [N] and ``memories`` = the episode bank [E, T, blocks, D] (indexed by ``memory_index``).
        With ``materialize=True`` ``memories`` is upstream's gathered [N, T, blocks, D] tensor instead.
        """
        if indices is None:
            indices = torch.randperm(self.batch_size, device=self.device)
        else:
            indices = torch.as_tensor(indices, device=self.device, dtype=torch.long)
        mbs = self.batch_size // self.n_mini_batches
        for start in range(0, self.batch_size, mbs):
            idx = indices[start: start + mbs]
            mini_batch = {k: v.index_select(0, idx) for k, v in self.samples_flat.items()}
            if materialize:
                mini_batch["memories"] = self.memories[mini_batch.pop("memory_index")]
            else:
                mini_batch["memories"] = self.memories
            yield mini_batch

    def gather(self, idx: torch.Tensor, materialize: bool = False) -> dict:
        """One minibatch dict (see ``mini_batch_generator``) for the flat sample indices ``idx``."""
        mini_batch = {k: v.index_select(0, idx) for k, v in self.samples_flat.items()}
        mini_batch["memories"] = self.memories[mini_batch.pop("memory_index")] if materialize else self.memories
        return mini_batch

    def calc_advantages(self, last_value: torch.Tensor, gamma: float, lamda: float) -> None:
        """GAE over the [W, S] buffer on device (upstream buffer.py:95-113)."""
        self.rewards_dev.copy_(self._rewards_host, non_blocking=True)
        self.dones_dev.copy_(self._dones_host, non_blocking=True)
        self._mark_host_arrays_uploaded()
        ops.gae(self.rewards_dev, self.dones_dev, self.values, last_value.detach(), gamma, lamda, out=self.advantages)
