"""Data-parallel plumbing: one process per GPU, gradients all-reduced with RCCL over xGMI (torch.distributed,
backend "nccl" on ROCm) -- the exchange step the upstream code does not have.

Design: workers (environments) are partitioned across ranks; each rank owns its rollout shard, memory bank and a
full model replica.  Per minibatch there is ONE collective on ONE flat fp32 bucket that aliases every ``p.grad``
(no per-parameter launches, message = 4 * n_params bytes), placed between ``backward()`` and gradient clipping
(upstream trainer.py:310-311), plus an 3-float all-gather so the advantage normalisation uses global-minibatch
statistics.  With the gloo backend the same code runs on CPU tensors (tests).
"""
import os

import torch
import torch.distributed as dist


def gpu_numa_cpus(device_index: int):
    """CPUs of the NUMA node the GPU hangs off (sysfs: the PCI function's numa_node -> that node's cpulist), or None if the
    platform does not say (single-node hosts report -1)."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            spec = f.read().strip()
        cpus = set()
        for part in spec.split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        return cpus or None
    except Exception:              # noqa: BLE001 -- no sysfs entry, no permission, odd format: leave the affinity alone
        return None


def pin_to_gpu_numa_node(device_index: int):
    """Restrict this process (and the threads it creates afterwards: the observation copier's helpers, torch's pool) to the CPUs
    of the GPU's NUMA node, so that the pinned staging buffers are first-touched on that node and the rollout loop's polls /
    uploads do not cross the socket interconnect.  Returns the CPU set, or None when nothing was changed."""
    cpus = gpu_numa_cpus(device_index)
    if not cpus:
        return None
    try:
        allowed = os.sched_getaffinity(0)
        cpus = cpus & allowed
        if len(cpus) < 4:
            return None
        os.sched_setaffinity(0, cpus)
        return cpus
    except Exception:              # noqa: BLE001
        return None


class DataParallel:
    def __init__(self, device=None, backend=None, collective=None):
        """``collective``: who issues the gradient all-reduce -- "etm" (default: the library's own RCCL communicator,
        ``etm_comm_*`` / ``etm_allreduce_f32`` of include/etm_hip.h, enqueued on the current stream; HIP device tensors and the
        nccl (= RCCL) process-group backend only; rendezvous id carried by torch.distributed) or "torch" (torch.distributed's
        all_reduce).  The environment variable ETM_DP_COLLECTIVE is used when the argument is None.  If the library
        communicator cannot be created (CPU tensors, gloo backend, several ranks on one device, RCCL not loadable) the torch
        collective takes over with a message -- a transport choice, the summed gradients are the same."""
        self.collective = collective or os.environ.get("ETM_DP_COLLECTIVE") or "etm"     # an explicit argument wins over the environment
        if self.collective not in ("torch", "etm"):
            raise ValueError(f"collective must be 'torch' or 'etm', got {self.collective!r}")
        self._comm = None
        self._graph_collective = None      # can the library all-reduce be captured into a HIP graph? (decided by graph_collective_ok)
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.flat = None
        if self.world > 1 and not dist.is_initialized():
            if backend is None:
                backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
            # dmabuf IPC is the only form this host driver has; the runtime reads the variable when it starts, so the entry points
            # (train.py, bench.py) set it before their first device call -- say so if a caller did not
            if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0" and backend == "nccl":
                print("[etm.dist] HSA_ENABLE_IPC_MODE_LEGACY=0 is not in the environment: RCCL needs it set before the HIP runtime "
                      "starts (export it, or set it before the first torch.cuda call)", flush=True)
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device(device)
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)
        if self.world > 1 and self.collective == "etm":
            self._start_etm_comm()

    @property
    def active(self):
        return self.world > 1

    def shard(self, total: int):
        """Contiguous shard of ``total`` items for this rank -> (first, count); total must divide evenly."""
        if total % self.world != 0:
            raise ValueError(f"{total} items do not divide over {self.world} ranks")
        per = total // self.world
        return self.rank * per, per

    def broadcast_parameters(self, module: torch.nn.Module):
        if not self.active:
            return
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=0)

    def attach_flat_grads(self, params):
        """Allocate one flat fp32 buffer and make every ``p.grad`` a view into it."""
        params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
        off = 0
        for p in params:
            p.grad = self.flat[off: off + p.numel()].view_as(p)
            off += p.numel()
        return self.flat

    def _start_etm_comm(self):
        """Create the library communicator now (not inside the first optimisation step, which may be under graph capture), on the
        calling thread, and check it with a four-element all-reduce.  Every step of the hand-shake is SYMMETRIC -- all ranks issue
        the same torch.distributed collectives in the same order whatever happens locally -- so a failure on one rank (RCCL not
        loadable, CPU tensors, gloo backend, several ranks on one device) can never leave ranks in different collectives:

          1. local pre-check (library loads, RCCL resolves -- every rank draws a throw-away rendezvous id --, the device answers)
                                                  -> all_reduce(MIN) of "I can try"          (everybody leaves here together if not)
          2. rank 0 draws the rendezvous id       -> broadcast of [status byte | 128-byte id] (rank 0 ALWAYS broadcasts; a failure
                                                                                               travels as status 1 + a zeroed id)
          3. etm_comm_init + self-test all-reduce -> all_reduce(MIN) of "mine works"

        Everything that can fail LOCALLY inside etm_comm_init is exercised in step 1, so no rank enters the RCCL rendezvous alone; a
        peer that dies between steps 1 and 3 is covered by a watchdog (ETM_COMM_TIMEOUT_S, default 600 s: message + exit), because
        RCCL's bootstrap waits without a bound for a peer that never connects."""
        from . import lib as _lib
        on_dev = dist.get_backend() == "nccl"
        flag_dev = self.device if on_dev else "cpu"

        def agree(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=flag_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))

        why = None
        if self.device is None or torch.device(self.device).type != "cuda" or not on_dev:
            why = "needs HIP device tensors and the nccl (RCCL) backend"
        else:
            # EVERY local precondition of etm_comm_init is exercised here, before anybody enters the RCCL rendezvous (a rank that
            # failed locally inside step 3 would leave its peers waiting in ncclCommInitRank for a peer that never connects):
            # the library loads, RCCL is found and its symbols resolve (drawing a rendezvous id does exactly that -- every rank
            # draws one, only rank 0's is used), the device answers
            try:
                import ctypes
                with torch.cuda.device(torch.device(self.device)):
                    torch.cuda.synchronize()
                    rc = _lib.load().etm_comm_unique_id(ctypes.create_string_buffer(128))
                if rc != 0:
                    why = f"etm_comm_unique_id failed ({rc}): RCCL not usable on this rank"
            except Exception as exc:       # noqa: BLE001
                why = repr(exc)
        if not agree(why is None):                                        # step 1
            return self._use_torch_collective(why)
        import ctypes
        lib = _lib.load()
        dev = torch.device(self.device)
        msg = torch.zeros(1 + 128, dtype=torch.uint8)
        if self.rank == 0:
            buf = ctypes.create_string_buffer(128)
            rc = lib.etm_comm_unique_id(buf)
            if rc == 0:
                msg[1:] = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8)
            else:
                msg[0] = 1
                why = f"etm_comm_unique_id failed ({rc})"
        # watchdog for the rendezvous itself (a peer that died between step 1 and step 3): RCCL's bootstrap waits without a bound
        # for a peer that never connects, so after ETM_COMM_TIMEOUT_S (default 600) the process is ended with a message instead
        import threading
        limit = float(os.environ.get("ETM_COMM_TIMEOUT_S", "600"))

        def _give_up():
            print(f"[etm.dist] rank {self.rank}: RCCL rendezvous did not complete within {limit:.0f} s (a peer is gone?) -- aborting",
                  flush=True)
            os._exit(3)

        watchdog = threading.Timer(limit, _give_up)
        watchdog.daemon = True
        watchdog.start()
        try:      # everything from the start of the watchdog to the last hand-shake step: the timer never outlives this call
            msg = msg.to(dev)
            dist.broadcast(msg, src=0)                                        # step 2 (unconditional on every rank)
            msg = msg.cpu()
            ok = int(msg[0]) == 0
            if ok:
                try:
                    comm = ctypes.c_void_p()
                    with torch.cuda.device(dev):
                        _lib.check(lib.etm_comm_init(bytes(msg[1:].numpy().tobytes()), self.rank, self.world, ctypes.byref(comm)), "etm_comm_init")
                        self._comm = comm
                        probe = torch.ones(4, dtype=torch.float32, device=dev)
                        _lib.check(lib.etm_allreduce_f32(comm, probe.data_ptr(), probe.data_ptr(), 4, torch.cuda.current_stream(dev).cuda_stream),
                                   "etm_allreduce_f32")
                        torch.cuda.synchronize(dev)
                    if probe.tolist() != [float(self.world)] * 4:
                        ok, why = False, f"self-test all-reduce returned {probe.tolist()}"
                except Exception as exc:       # noqa: BLE001 -- any failure: use the framework's collective
                    ok, why = False, repr(exc)
            elif why is None:
                why = "rank 0 could not draw a rendezvous id"
            agreed = agree(ok)                                            # step 3
        finally:
            watchdog.cancel()
        if not agreed:
            return self._use_torch_collective(why)
        # step 4 (round 6): the same collective INSIDE a captured graph -- the optimisation step then is ONE graph per minibatch
        # (forward, backward, all-reduce, clip + AdamW) instead of two replays around a host call.  Symmetric like the steps above:
        # every rank captures and replays, then all agree; any failure leaves the three-call step in place.
        watchdog = threading.Timer(limit, _give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            # two phases, each followed by an agreement: a rank whose CAPTURE fails issues no collective, so nobody may replay (= issue
            # the captured all-reduce) before every rank is known to hold a graph
            captured = self._graph_selftest_capture()
            if agree(captured is not None):
                self._graph_collective = agree(self._graph_selftest_replay(captured))
            else:
                self._graph_collective = False
        finally:
            watchdog.cancel()
        if not self._graph_collective and self.rank == 0:
            print("[etm.dist] the library all-reduce could not be captured into a HIP graph on every rank: the optimisation step stays "
                  "graph A -> all-reduce -> graph B", flush=True)

    def _graph_selftest_capture(self):
        """Library all-reduce of 4 floats captured into a HIP graph (no communication happens during a capture).  -> (graph, src, buf) or
        None on any failure."""
        from . import lib as _lib
        if os.environ.get("ETM_DP_GRAPH_COLLECTIVE", "1") == "0" or self._comm is None:
            return None
        dev = torch.device(self.device)
        try:
            with torch.cuda.device(dev):
                src = torch.ones(4, dtype=torch.float32, device=dev)
                buf = torch.zeros(4, dtype=torch.float32, device=dev)
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                        buf.copy_(src)
                        _lib.check(_lib.load().etm_allreduce_f32(self._comm, buf.data_ptr(), buf.data_ptr(), 4,
                                                                 torch.cuda.current_stream(dev).cuda_stream), "etm_allreduce_f32")
                        buf.mul_(2.0)
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize(dev)
                return g, src, buf
        except Exception as exc:       # noqa: BLE001 -- capture not supported here: the eager collective stays
            print(f"[etm.dist] rank {self.rank}: graph capture of the library all-reduce failed ({exc!r})", flush=True)
            try:
                torch.cuda.synchronize(dev)
            except Exception:          # noqa: BLE001
                pass
            return None

    def _graph_selftest_replay(self, captured):
        """Two replays of the captured all-reduce with different inputs; every rank calls this (or none does)."""
        g, src, buf = captured
        dev = torch.device(self.device)
        try:
            ok = True
            with torch.cuda.device(dev):
                for k in (1.0, 3.0):
                    src.fill_(k)
                    g.replay()
                    torch.cuda.synchronize(dev)
                    ok = ok and buf.tolist() == [2.0 * k * self.world] * 4      # (the communicator has self.world ranks)
            return ok
        except Exception as exc:       # noqa: BLE001
            print(f"[etm.dist] rank {self.rank}: replay of the captured all-reduce failed ({exc!r})", flush=True)
            return False

    def _graph_selftest(self):
        captured = self._graph_selftest_capture()
        return captured is not None and self._graph_selftest_replay(captured)

    def agree(self, ok: bool) -> bool:
        """Logical AND of ``ok`` over the ranks (one small torch.distributed all-reduce; every rank must call it)."""
        if not self.active or not dist.is_initialized():
            return bool(ok)
        dev = self.device if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def graph_collective_ok(self):
        """True when the gradient all-reduce may be captured inside the optimisation step's graph: library communicator in use and
        its captured form self-tested on every rank (multi-rank: at construction; world size 1 with the collective forced on --
        tests --: here, on first use)."""
        if self.collective != "etm":
            return False
        if self._graph_collective is None:
            if self.world != 1:
                return False              # (multi-rank runs decide in _start_etm_comm, together)
            if self._comm is None:
                self._comm = self._single_rank_comm()
            self._graph_collective = self._graph_selftest()
        return bool(self._graph_collective)

    def _use_torch_collective(self, why):
        if self.rank == 0 or why is not None:
            print(f"[etm.dist] rank {self.rank}: library RCCL communicator not used ({why or 'another rank could not create it'}); "
                  "gradient all-reduce goes through torch.distributed", flush=True)
        if self._comm is not None:
            try:
                from . import lib as _lib
                _lib.load().etm_comm_destroy(self._comm)
            except Exception:          # noqa: BLE001
                pass
            self._comm = None
        self.collective = "torch"

    def all_reduce_slice(self, lo: int, hi: int, stream):
        """Sum ``flat[lo:hi]`` over the ranks on ``stream`` (a torch.cuda.Stream): the pieces of the overlapped data-parallel step
        (trainer.py: dp_overlap).  No averaging: the division by the world size rides in the optimiser's clip coefficient."""
        if not self.active or hi <= lo:
            return
        part = self.flat[lo:hi]
        if self.collective == "etm":
            from . import lib as _lib
            if self._comm is None:
                if self.world != 1:
                    raise RuntimeError("library communicator missing (DataParallel creates it at construction)")
                self._comm = self._single_rank_comm()
            _lib.check(_lib.load().etm_allreduce_f32(self._comm, part.data_ptr(), part.data_ptr(), part.numel(), stream.cuda_stream),
                       "etm_allreduce_f32")
        elif stream is not None and part.is_cuda:
            with torch.cuda.stream(stream):
                dist.all_reduce(part, op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(part, op=dist.ReduceOp.SUM)

    def all_reduce_grads(self, average=True):
        """Sum the flat gradient bucket over ranks (one RCCL all-reduce).  ``average=True`` also divides by the world size (one more
        launch over the bucket); the trainer passes False and hands ``grad_scale = 1 / world`` to the optimiser step instead, where
        the division rides in the clip coefficient (csrc/optim.hip)."""
        if not self.active:
            return
        if self.collective == "etm":
            from . import lib as _lib
            if self._comm is None:
                if self.world != 1:
                    raise RuntimeError("library communicator missing (DataParallel creates it at construction)")
                self._comm = self._single_rank_comm()     # world size 1 with the collective forced on (tests): no rendezvous needed
            rc = _lib.load().etm_allreduce_f32(self._comm, self.flat.data_ptr(), self.flat.data_ptr(), self.flat.numel(),
                                               torch.cuda.current_stream(self.flat.device).cuda_stream)
            _lib.check(rc, "etm_allreduce_f32")
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if average:
            self.flat.div_(self.world)

    def _single_rank_comm(self):
        import ctypes
        from . import lib as _lib
        lib = _lib.load()
        buf = ctypes.create_string_buffer(128)
        _lib.check(lib.etm_comm_unique_id(buf), "etm_comm_unique_id")
        comm = ctypes.c_void_p()
        with torch.cuda.device(torch.device(self.device)):
            _lib.check(lib.etm_comm_init(bytes(buf.raw), 0, 1, ctypes.byref(comm)), "etm_comm_init")
        return comm

    @property
    def grad_scale(self):
        """1 / world: what the optimiser step multiplies into the clip coefficient after ``all_reduce_grads(average=False)``."""
        return 1.0 / self.world if self.active else 1.0

    def merge_adv_stats(self, stats3: torch.Tensor) -> torch.Tensor:
        """Merge per-rank (count, mean, M2) into global statistics (Chan et al. pairwise update).  ``stats3`` is [3] or
        [k, 3] (k independent minibatches merged row-wise with ONE all-gather)."""
        if not self.active:
            return stats3
        gathered = [torch.empty_like(stats3) for _ in range(self.world)]
        dist.all_gather(gathered, stats3.contiguous())
        allst = torch.stack(gathered)               # [world, (k,) 3]
        n, mean, m2 = allst[..., 0], allst[..., 1], allst[..., 2]
        tot = n.sum(dim=0)
        gmean = (n * mean).sum(dim=0) / tot
        gm2 = (m2 + n * (mean - gmean) ** 2).sum(dim=0)
        return torch.stack([tot, gmean, gm2], dim=-1)

    def max_over_ranks(self, value: float) -> float:
        if not self.active:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        if self.active:
            dist.barrier()

    def close(self):
        if self._comm is not None:
            from . import lib as _lib
            _lib.load().etm_comm_destroy(self._comm)
            self._comm = None
        if self.active and dist.is_initialized():
            dist.destroy_process_group()
