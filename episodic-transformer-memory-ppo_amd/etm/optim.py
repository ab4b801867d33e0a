"""AdamW + global-norm clipping on flat fp32 arenas (upstream trainer.py:311-312: ``clip_grad_norm_`` + ``optimizer.step()``).

Every parameter becomes a view into one parameter arena and every ``.grad`` a view into one gradient arena of the same
layout (the gradient arena is also the message of the data-parallel all-reduce); the two AdamW moments are arenas too.  The
step is then two kernel launches (``etm_grad_sqnorm``, ``etm_adamw_clip``; csrc/optim.hip) whatever the number of parameters,
with the learning rate and the step counter on the device so that a captured HIP graph replays it under changing schedules.
``state_dict`` keys, shapes and values of the model are untouched (views share storage, nothing is renamed).
"""
import torch

from . import lib as _lib


class FlatAdamW:
    N_PARTIAL = 1024

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.params = [p for p in params if p.requires_grad]
        if not self.params or not self.params[0].is_cuda:
            raise RuntimeError("FlatAdamW needs parameters on the MI355X (HIP) device")
        dev = self.params[0].device
        self.device = dev
        self.total = sum(p.numel() for p in self.params)
        padded = (self.total + 3) // 4 * 4
        self.flat_params = torch.zeros(padded, dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(padded, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(padded, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(padded, dtype=torch.float32, device=dev)
        self.grad_views = []
        off = 0
        with torch.no_grad():
            for p in self.params:
                if p.dtype != torch.float32:
                    raise TypeError("FlatAdamW: float32 parameters only")
                n = p.numel()
                view = self.flat_params[off: off + n].view(p.shape)
                view.copy_(p.data)
                p.data = view                                   # the parameter now lives in the arena
                p.grad = self.flat_grads[off: off + n].view(p.shape)
                self.grad_views.append(p.grad)
                off += n
        self.lr_dev = torch.tensor(float(lr), dtype=torch.float32, device=dev)
        self._lr_host = float(lr)
        self.step_dev = torch.zeros((), dtype=torch.int64, device=dev)
        self.partial = torch.zeros(self.N_PARTIAL, dtype=torch.float32, device=dev)
        self.total_norm = torch.zeros((), dtype=torch.float32, device=dev)     # un-clipped gradient norm of the last step
        self.betas, self.eps, self.weight_decay = (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)

    def set_lr(self, lr):
        if float(lr) != self._lr_host:
            self.lr_dev.fill_(float(lr))
            self._lr_host = float(lr)

    def zero_grad(self):
        self.flat_grads.zero_()

    def step(self, max_grad_norm=0.0, grad_scale=1.0):
        """Clip the gradient arena to ``max_grad_norm`` (global L2 norm, the rule of ``clip_grad_norm_``; <= 0: no clipping) and
        apply one AdamW update.  Two launches on the current stream.  ``grad_scale``: the arena holds gradient / grad_scale
        (data parallel: the all-reduced sum, grad_scale = 1 / world); the scale rides in the clip coefficient."""
        lib = _lib.load()
        st = torch.cuda.current_stream(self.device).cuda_stream
        n = self.flat_params.numel()
        _lib.check(lib.etm_grad_sqnorm(self.flat_grads.data_ptr(), n, self.partial.data_ptr(), self.N_PARTIAL, self.step_dev.data_ptr(), st),
                   "etm_grad_sqnorm")
        _lib.check(lib.etm_adamw_clip(self.flat_params.data_ptr(), self.flat_grads.data_ptr(), self.exp_avg.data_ptr(),
                                      self.exp_avg_sq.data_ptr(), n, self.partial.data_ptr(), self.N_PARTIAL, self.lr_dev.data_ptr(),
                                      self.step_dev.data_ptr(), self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                      float(max_grad_norm), float(grad_scale), self.total_norm.data_ptr(), st), "etm_adamw_clip")
