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
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.flat = None
        if self.world > 1 and not dist.is_initialized():
            if backend is None:
                backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (the only form this host driver has)
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
        """Create the library communicator now (not inside the first optimisation step, which may be under graph capture) and
        check it with a one-element all-reduce; any failure selects the torch collective."""
        import threading
        out = {"why": "did not finish"}

        def create_and_probe():
            try:
                if self.device is None or torch.device(self.device).type != "cuda" or dist.get_backend() != "nccl":
                    out["why"] = "needs HIP device tensors and the nccl (RCCL) backend"
                    return
                from . import lib as _lib
                dev = torch.device(self.device)
                with torch.cuda.device(dev):
                    comm = self._etm_comm()
                    probe = torch.ones(4, dtype=torch.float32, device=dev)
                    rc = _lib.load().etm_allreduce_f32(comm, probe.data_ptr(), probe.data_ptr(), 4, torch.cuda.current_stream(dev).cuda_stream)
                    _lib.check(rc, "etm_allreduce_f32")
                    torch.cuda.synchronize(dev)
                out["why"] = None if probe.tolist() == [float(self.world)] * 4 else f"self-test all-reduce returned {probe.tolist()}"
            except Exception as exc:       # noqa: BLE001 -- any failure: use the framework's collective
                out["why"] = repr(exc)

        # bounded: a rendezvous that never completes on some node must not stall the job -- the torch collective takes over
        limit = float(os.environ.get("ETM_COMM_TIMEOUT_S", "120"))
        th = threading.Thread(target=create_and_probe, daemon=True)
        th.start()
        th.join(limit)
        why = f"not ready after {limit:.0f} s" if th.is_alive() else out["why"]
        # all ranks must agree on the transport
        flag = torch.tensor([0 if why is None else 1], dtype=torch.int32,
                            device=self.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) != 0:
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

    def _etm_comm(self):
        """Library-owned RCCL communicator (created on first use): rank 0 draws the rendezvous id, torch.distributed carries
        its 128 bytes to the other ranks, every rank joins with its current device."""
        if self._comm is None:
            import ctypes
            from . import lib as _lib
            lib = _lib.load()
            dev = torch.device(self.device)
            if dev.type != "cuda":
                raise RuntimeError("collective='etm' needs HIP device tensors (RCCL); use the torch collective for CPU / gloo runs")
            buf = ctypes.create_string_buffer(128)
            if self.rank == 0:
                _lib.check(lib.etm_comm_unique_id(buf), "etm_comm_unique_id")
            on_dev = dist.get_backend() == "nccl"
            t = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            t = t.to(dev) if on_dev else t
            dist.broadcast(t, src=0)
            raw = bytes(t.cpu().numpy().tobytes())
            comm = ctypes.c_void_p()
            with torch.cuda.device(dev):
                _lib.check(lib.etm_comm_init(raw, self.rank, self.world, ctypes.byref(comm)), "etm_comm_init")
            self._comm = comm
        return self._comm

    def all_reduce_grads(self):
        """Sum the flat gradient bucket over ranks and average (one RCCL all-reduce)."""
        if not self.active:
            return
        if self.collective == "etm":
            from . import lib as _lib
            rc = _lib.load().etm_allreduce_f32(self._etm_comm(), self.flat.data_ptr(), self.flat.data_ptr(), self.flat.numel(),
                                               torch.cuda.current_stream(self.flat.device).cuda_stream)
            _lib.check(rc, "etm_allreduce_f32")
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(self.world)

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
