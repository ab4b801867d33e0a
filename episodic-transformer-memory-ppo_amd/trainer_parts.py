"""Parts of ``PPOTrainer`` (trainer.py) that are not the sampler / optimiser core, as mix-ins (round 6: trainer.py was 1,591 lines):

* ``_DataParallelStep``   the overlapped / one-graph forms of the data-parallel minibatch step (insertion point upstream trainer.py:310-311);
* ``_NativeRolloutDrive`` the rollout loop through the kernel library's driver (worker processes; upstream trainer.py:159-218);
* ``_RunOutputs``         TensorBoard summaries, the monitored gradient norms and the checkpoint (upstream trainer.py:325-362, model.py:128-151).

Every method runs on the trainer's own attributes; nothing here is importable on its own."""
import os
import pickle
import sys

import numpy as np
import torch

from etm import lib as etm_lib
from etm import ops
from etm.ops import WindowSpec
from model import IndexedObservations


class _DataParallelStep:
    # ---- data-parallel overlap (dp_overlap, SURVEY 8e / upstream insertion point trainer.py:310-311): the backward pass is cut at the
    # encoder output.  Part 1 (heads, transformer, lin_hidden: 98 % of the gradient arena) is summed over the ranks on a side stream
    # while part 2 (the encoder's backward, ~0.5 ms at config 3) runs; the small convolution slice follows on the main stream.
    def _train_body_a1(self, idx, clip_range, beta, stats3=None):
        """Gather, forward, loss, backward DOWN TO the encoder features; every gradient except the convolutions' is in its arena
        view afterwards.  Returns (stats[6], d loss / d features) -- the features themselves stay in ``self.model._encoder_features``."""
        buf = self.buffer
        skip = ("obs",) if self._obs_train is not None else ()
        keys = [k for k in buf.samples_flat if k not in skip]
        mb = dict(zip(keys, ops.gather_rows([buf.samples_flat[k] for k in keys], idx)))
        if self._bank_pos is not None:
            spec = WindowSpec.from_bank(self._bank_pos_buf, mb["memory_index"], mb["memory_indices"], None, mb["memory_mask"])
            spec.pos_included = True
            spec.row_stats = getattr(self, "_row_stats", None)
        else:
            spec = WindowSpec.from_bank(buf.bank, mb["memory_index"], mb["memory_indices"], mb["memory_indices"], mb["memory_mask"])
            if self.model.transformer.pos_kind == "":
                spec.row_stats = getattr(self, "_row_stats", None)
        obs = IndexedObservations(self._obs_train, idx)
        if stats3 is None:
            stats3 = ops.adv_stats(mb["advantages"])
        self.model._encoder_features, self.model._keep_encoder_features = None, True
        try:
            loss, stats = self._loss_from(obs, spec, mb, clip_range, beta, stats3)
        finally:
            self.model._keep_encoder_features = False
        feats = self.model._encoder_features
        if feats is None or feats.grad_fn is None:
            raise RuntimeError("dp_overlap needs the hand-written training encoder (visual observations, fused_train_encoder)")
        n_conv = self._n_conv_params
        rest = self.params[n_conv:]
        for p in self.params:
            p.grad = None
        with ops.DeferredDw(self._dw_destinations()) as dw:
            got = torch.autograd.grad(loss, [feats] + rest, grad_outputs=self._unit_gradient(loss), allow_unused=True)
        dfeats, grads_rest = got[0], got[1:]
        views, grads = [], []
        for p, v, g in zip(rest, self._grad_views[n_conv:], grads_rest):
            if p.data_ptr() in dw.written:
                if g is not None:
                    v.add_(g)
                continue
            views.append(v)
            grads.append(g if g is not None else torch.zeros_like(v))
        if views:
            torch._foreach_copy_(views, grads)
        return stats, dfeats

    def _train_body_a2(self, dfeats):
        """The encoder's backward pass from d loss / d features; the convolutions' gradients end up in their arena views."""
        feats = self.model._encoder_features
        n_conv = self._n_conv_params
        convs = self.params[:n_conv]
        with ops.DeferredDw(self._dw_destinations()) as dw:
            got = torch.autograd.grad(feats, convs, grad_outputs=dfeats, allow_unused=True)
        views, grads = [], []
        for p, v, g in zip(convs, self._grad_views[:n_conv], got):
            if p.data_ptr() in dw.written:
                if g is not None:
                    v.add_(g)
                continue
            views.append(v)
            grads.append(g if g is not None else torch.zeros_like(v))
        if views:
            torch._foreach_copy_(views, grads)
        for p, v in zip(self.params, self._grad_views):
            p.grad = v
        self.model._encoder_features = None

    def _dp_overlap_ready(self):
        """dp_overlap applies when the run is data parallel, the optimisation phase runs the hand-written encoder on indexed
        observations, and the convolution parameters are the FIRST parameters of the arena (model.py: conv1..3 are created first)."""
        if self.dp is None or not self.config.get("dp_overlap", False) or self._obs_train is None:
            return False
        if getattr(self, "_n_conv_params", None) is None:
            names = [n for n, p in self.model.named_parameters() if p.requires_grad]
            k = 0
            while k < len(names) and names[k].startswith("conv"):
                k += 1
            self._n_conv_params = k if (k > 0 and not any(n.startswith("conv") for n in names[k:])) else 0
            self._conv_floats = sum(p.numel() for p in self.params[: self._n_conv_params])
            v0, vk = self._grad_views[0], self._grad_views[self._n_conv_params]
            if self._n_conv_params and (vk.data_ptr() - v0.data_ptr()) // 4 != self._conv_floats:
                self._conv_floats = (vk.data_ptr() - v0.data_ptr()) // 4        # (arena views are padded: the slice boundary in floats)
            self._ar_side = torch.cuda.Stream(device=self.device)
            self._ar_fork, self._ar_fork2, self._ar_join = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        return self._n_conv_params > 0

    def _capture_one_graph_dp_step(self, clip_range, beta, monitor, overlap):
        """Data-parallel minibatch step as ONE graph (round 6): gather, forward, loss, backward, the library's RCCL all-reduce of
        the flat gradient arena (etm_allreduce_f32 enqueues on the capturing stream like every other entry; with dp_overlap the
        two slices on the side stream, which joins the capture through the fork event and leaves it through the join event), clip +
        AdamW.  Returns the captured graph or None (capture failed: the caller captures graph A / graph B around the host-side
        collective instead)."""
        ga = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(ga, capture_error_mode="thread_local"):
                if overlap:
                    self._tg_stats, dfe = self._train_body_a1(self._tg_idx, clip_range, beta, self._tg_stats3)
                    self._allreduce_rest_async()
                    self._train_body_a2(dfe)
                    self._allreduce_conv_and_join()
                else:
                    self._tg_stats = self._train_body_a(self._tg_idx, clip_range, beta, self._tg_stats3)
                    self.dp.all_reduce_grads(average=False)
                self._tg_norms = self._train_body_b(monitor)
            return ga
        except Exception as exc:       # noqa: BLE001
            print(f"[etm] one-graph data-parallel step not captured ({exc!r}); using graph A -> all-reduce -> graph B", file=sys.stderr, flush=True)
            torch.cuda.synchronize(self.device)
            self._dp_one_graph_failed = True
            return None

    def _allreduce_rest_async(self):
        main = torch.cuda.current_stream(self.device)
        self._ar_fork.record(main)
        self._ar_side.wait_event(self._ar_fork)
        self.dp.all_reduce_slice(self._conv_floats, self.flat_grads.numel(), self._ar_side)

    def _allreduce_conv_and_join(self):
        """The convolutions' slice follows on the SAME side stream (one communicator: its collectives stay on one stream, in one
        order on every rank), after the encoder's backward pass; the main stream then waits for both."""
        main = torch.cuda.current_stream(self.device)
        self._ar_fork2.record(main)
        self._ar_side.wait_event(self._ar_fork2)
        self.dp.all_reduce_slice(0, self._conv_floats, self._ar_side)
        self._ar_join.record(self._ar_side)
        if getattr(self, "_ar_probe", None) is not None:       # bench.py: how long the main stream really waits for the side stream
            self._ar_probe[0].record(main)
        main.wait_event(self._ar_join)
        if getattr(self, "_ar_probe", None) is not None:
            self._ar_probe[1].record(main)


class _NativeRolloutDrive:
    def _drive_rollout_native(self, groups, episode_infos):
        """Steps 0 .. S - 1 of a rollout through the kernel library's driver (csrc/rollout_driver.hip): step 0 of every group is
        already enqueued; the workers (processes, environments/shm_env.py) take their actions from the device and publish their
        results in the shared segment; this call blocks until the bookkeeping of the last step is done.  Afterwards: rewards /
        done flags / episode results / memory_index rows are taken over from the segment and the driver's event list."""
        import ctypes
        buf, W, S = self.buffer, self.num_workers, self.config["worker_steps"]
        env, lib = self._shm_env, etm_lib.load()
        G = len(groups)
        arr = (etm_lib.RolloutGroup * G)()
        row_bytes = self._obs_pin[0].numel() * 4
        stage = self._stage["obs"]
        for gi, g in enumerate(groups):
            a = arr[gi]
            a.graph_exec = g.graphs[0].raw_cuda_graph_exec()
            a.stream = g.stream.cuda_stream
            first = gi * env.procs_per_group
            a.ready = env.v["ready"][first:].ctypes.data
            a.n_procs, a.ready_stride = env.procs_per_group, env.v["ready"].shape[1]
            a.lo, a.hi = g.lo, g.hi
            a.obs_src = self._obs_pin.data_ptr() + g.lo * row_bytes
            a.stage_dst = stage.data_ptr() + g.lo * row_bytes
            a.ss_dst = g.ss_pin.data_ptr()
        if getattr(self, "_drive_events", None) is None:
            self._drive_events = np.zeros((W * S, 3), dtype=np.int64)
            self._drive_counters = np.zeros(2, dtype=np.int64)          # [next slot, number of events]
            self._drive_timing = np.zeros(2, dtype=np.float64)
        ctr = self._drive_counters
        ctr[0], ctr[1] = buf.num_episodes, 0
        chain = None
        if self._chain_log is not None:
            chain = np.zeros((S, 4), dtype=np.float64)
        abort = env.v["err"]          # the workers' error words (one cache line apart) ...
        # (groups are served ready-first: as soon as a group's worker processes have published the step; slot numbers stay in
        # (step, group) order -- csrc/rollout_driver.hip)
        rc = lib.etm_rollout_drive(ctypes.cast(arr, ctypes.c_void_p), G, 0, S, W, row_bytes, W * row_bytes,
                                   env.v["dones"].ctypes.data, self._ss_pin[0].data_ptr(), self._ss_pin[1].data_ptr(),
                                   ctr.ctypes.data, int(buf.bank.shape[0]), self._drive_events.ctypes.data, self._drive_events.shape[0],
                                   ctr[1:].ctypes.data, abort.ctypes.data, abort.shape[0], abort.shape[1],
                                   float(self.config.get("rollout_step_timeout_s", 30.0)), self._drive_timing.ctypes.data,
                                   chain.ctypes.data if chain is not None else None)
        env.park()
        if rc != 0:
            env._check()
            etm_lib.check(rc, "etm_rollout_drive")
        buf.num_episodes = int(ctr[0])
        buf.rewards[:, :] = env.v["rewards"].T
        buf.dones[:, :] = env.v["dones"].T.astype(bool)
        for t, w, slot in self._drive_events[: int(ctr[1])]:
            episode_infos.append(env.info_at(int(t), int(w)))
            if t < S - 1:
                buf.memory_index_host[w, t + 1:] = slot
        if chain is not None:
            self._chain_log.extend(tuple(r) for r in chain[: S - 1])
        return float(self._drive_timing[0]), float(self._drive_timing[1])


class _RunOutputs:
    def _write_training_summary(self, update, training_stats, episode_result, value_mean, advantage_mean, steps_per_s) -> None:
        if episode_result:
            for key in episode_result:
                if "std" not in key:
                    self.writer.add_scalar("episode/" + key, episode_result[key], update)
        self.writer.add_scalar("losses/loss", training_stats[2], update)
        self.writer.add_scalar("losses/policy_loss", training_stats[0], update)
        self.writer.add_scalar("losses/value_loss", training_stats[1], update)
        self.writer.add_scalar("losses/entropy", training_stats[3], update)
        self.writer.add_scalar("training/value_mean", value_mean, update)
        self.writer.add_scalar("training/advantage_mean", advantage_mean, update)
        # upstream swaps these two tags (trainer.py:343-344 vs :322-323); written correctly here
        self.writer.add_scalar("other/kl", training_stats[4], update)
        self.writer.add_scalar("other/clip_fraction", training_stats[5], update)
        self.writer.add_scalar("other/env_steps_per_second", steps_per_s, update)

    def _write_gradient_summary(self, update, grad_info):
        for key, value in grad_info.items():
            self.writer.add_scalar("gradients/" + key, np.mean(value), update)

    def _save_model(self) -> None:
        """``pickle((state_dict, config))`` to ./models/<run_id>.nn -- upstream's checkpoint format (trainer.py:356-362)."""
        os.makedirs("./models", exist_ok=True)
        state = {k: v.detach().cpu() for k, v in self.model.state_dict().items()}
        path = "./models/" + self.run_id + ".nn"
        with open(path + ".tmp", "wb") as f:
            pickle.dump((state, self.config), f)
        os.replace(path + ".tmp", path)                # never a torn file at the final path
        print("Model saved to " + "./models/" + self.run_id + ".nn")

    def _build_grad_groups(self):
        """Group-membership matrix so all monitored gradient norms come from one pass over per-parameter norms."""
        groups = self.model._grad_groups()
        index = {id(p): i for i, p in enumerate(self.params)}
        self._grad_keys = list(groups.keys())
        member = torch.zeros((len(groups), len(self.params)), dtype=torch.float32)
        for g, modules in enumerate(groups.values()):
            for m in modules:
                for p in m.parameters():
                    member[g, index[id(p)]] += 1.0   # upstream concatenates, so a parameter listed twice counts twice
        self._grad_member = member.to(self.device)
        # segments of the flat gradient arena (<= 4096 floats, inside one tensor) for etm_group_norms
        base = self.flat_grads.data_ptr()
        starts, lens, owner = [], [], []
        for i, p in enumerate(self.params):
            off = (self._grad_views[i].data_ptr() - base) // 4
            for s in range(0, p.numel(), 4096):
                starts.append(off + s)
                lens.append(min(4096, p.numel() - s))
                owner.append(i)
        self._seg_start = torch.tensor(starts, dtype=torch.int64, device=self.device)
        self._seg_len = torch.tensor(lens, dtype=torch.int32, device=self.device)
        self._seg_member = member[:, owner].contiguous().to(self.device)
        self._seg_partial = torch.empty(len(starts), dtype=torch.float32, device=self.device)

    def _grad_group_norms(self):
        if self.flat_grads.is_cuda and all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(self.params[:2], self._grad_views[:2])):
            out = torch.empty(len(self._grad_keys), dtype=torch.float32, device=self.device)
            etm_lib.check(etm_lib.load().etm_group_norms(self.flat_grads.data_ptr(), self._seg_start.data_ptr(), self._seg_len.data_ptr(),
                                                         self._seg_start.numel(), self._seg_member.data_ptr(), len(self._grad_keys),
                                                         self._seg_partial.data_ptr(), out.data_ptr(),
                                                         torch.cuda.current_stream(self.device).cuda_stream), "etm_group_norms")
            return out
        sq = torch.stack(torch._foreach_norm([p.grad for p in self.params])) ** 2
        return torch.sqrt(self._grad_member @ sq)

    # ------------------------------------------------------------------ logging / checkpoint
