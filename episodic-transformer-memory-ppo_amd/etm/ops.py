"""torch-facing wrappers (autograd glue) around the C ABI of libetm_hip.so.

PyTorch is plumbing here: it owns the device buffers and the stream; the math of the three hot ops runs in the
hand-written kernels.  Every entry point requires HIP device tensors and raises otherwise -- no CPU/eager fallback.
"""
import torch

from . import lib as _lib

_workspaces = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _need_dev(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("etm ops need tensors on the MI355X (HIP) device; there is no CPU path in this build "
                               "(the CPU oracle lives under oracle/ and is test-only)")


def _f32c(t, name):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


_frozen = set()
_retired = []


def workspace(nbytes, device, tag="ws"):
    """Scratch buffer of at least ``nbytes`` per (tag, device), grown on demand.  Once a HIP graph has captured a buffer's
    address (``freeze_workspaces``) that buffer is never freed: a later, larger request gets a NEW buffer and the old one is
    retired but kept alive, so replays of the captured graph keep writing to memory they own."""
    key = (tag, device.index)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None and key in _frozen:
            _retired.append(buf)
            _frozen.discard(key)
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def freeze_workspaces(device):
    """Called by the trainer right after a graph capture: the scratch buffers that exist now are referenced by the graph."""
    for key in _workspaces:
        if key[1] == device.index:
            _frozen.add(key)


class WindowSpec:
    """Where the memory-window rows of a batch live: X[n, l] = bank[ep[n], win[n, l], block, :].

    bank_ptr/ep_stride/row_stride/block_stride are in float32 elements.  ``ep`` may be None (ep[n] = n).
    """

    __slots__ = ("bank", "ep_stride", "row_stride", "block_stride", "ep", "win", "pidx", "mask", "N", "L", "pos_included", "row_stats",
                 "_gathered_stats")

    def __init__(self, bank, ep_stride, row_stride, block_stride, ep, win, pidx, mask):
        _need_dev(bank, ep, win, pidx, mask)
        if bank.dtype != torch.float32 or bank.stride(-1) != 1:
            raise TypeError("memory bank must be float32 with a contiguous feature dimension")
        for s in (ep_stride, row_stride, block_stride):
            if s % 4 != 0:
                raise ValueError("memory bank strides must be multiples of 4 floats (16-byte rows)")
        if bank.data_ptr() % 16 != 0:
            raise ValueError("memory bank must be 16-byte aligned")
        self.bank = bank
        self.ep_stride, self.row_stride, self.block_stride = int(ep_stride), int(row_stride), int(block_stride)
        self.N, self.L = int(win.shape[0]), int(win.shape[1])
        self.pos_included = False   # True: the bank rows already contain their positional rows (see Transformer.bank_with_positions)
        # round 5, pre-LN models: (mean, rstd) of every bank row [blocks, E, T, 2] (bank_row_stats; once per update: norm_kv's statistics
        # do not depend on its gain / bias) -- the window passes then gather their [N, L, 2] statistics instead of re-reading every
        # window row to compute them (472 MB per launch at config 5); None: computed per window row inside etm_window_fwd
        self.row_stats = None
        self._gathered_stats = {}
        self.ep = None if ep is None else ep.to(torch.int64).contiguous()
        self.win = win.to(torch.int64).contiguous()
        self.pidx = None if pidx is None else pidx.to(torch.int64).contiguous()
        if mask.dtype == torch.bool:
            m = mask.contiguous().view(torch.uint8)
        elif mask.dtype == torch.uint8:
            m = mask.contiguous()
        else:
            m = (mask != 0).to(torch.uint8)  # reference semantics: `mask == 0` is masked (transformer.py:66)
        self.mask = m

    @classmethod
    def from_bank(cls, bank, ep, win, pidx, mask):
        """bank: [E, T, nb, D] episode bank."""
        return cls(bank, bank.stride(0), bank.stride(1), bank.stride(2), ep, win, pidx, mask)

    @classmethod
    def from_windows(cls, memories, pidx, mask):
        """memories: pre-gathered windows [N, L, nb, D] (the reference's calling convention) or [N, L, D]."""
        n, l = memories.shape[0], memories.shape[1]
        win = torch.arange(l, device=memories.device, dtype=torch.int64).unsqueeze(0).expand(n, l)
        bstride = memories.stride(2) if memories.dim() == 4 else 0
        return cls(memories, memories.stride(0), memories.stride(1), bstride, None, win, pidx, mask)

    def block_ptr(self, block):
        return self.bank.data_ptr() + 4 * block * self.block_stride

    def window_stats(self, block):
        """[N, L, 2] LayerNorm statistics of this batch's window rows of ``block``, gathered from ``row_stats`` (None without it)."""
        if self.row_stats is None:
            return None
        got = self._gathered_stats.get("all")
        if got is None:                     # one gather for all blocks (the window indices are the same): 2 launches per step, not 2 per block
            ep = self.ep if self.ep is not None else torch.arange(self.N, device=self.win.device)
            got = self._gathered_stats["all"] = self.row_stats[:, ep.unsqueeze(1), self.win].contiguous()       # [blocks, N, L, 2]
        return got[block]


def bank_row_stats(bank, eps, out=None):
    """(mean, 1 / sqrt(var + eps)) of every row of a BLOCK-MAJOR episode bank view [E, T, blocks, D] (memory order [blocks][E][T][D],
    buffer.py) -> [blocks, E, T, 2] (etm_ln_row_stats: one pass over the used part of the bank, once per update)."""
    lib = _lib.load()
    E, T, nb, D = bank.shape
    rows = bank.permute(2, 0, 1, 3)                      # [blocks, E, T, D]: contiguous for a block-major bank
    if not rows.is_contiguous():
        rows = rows.contiguous()
    if out is None or out.shape != (nb, E, T, 2):
        out = torch.empty((nb, E, T, 2), dtype=torch.float32, device=bank.device)
    _lib.check(lib.etm_ln_row_stats(_ptr(rows), float(eps), _ptr(out), nb * E * T, D, _stream()), "etm_ln_row_stats")
    return out


class _MhaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, wk, wv, ln_g, ln_b, pos, spec, block, num_heads, ln_eps):
        lib = _lib.load()
        _need_dev(q, wk, wv, ln_g, ln_b, pos)
        q, wk, wv = _f32c(q, "q"), _f32c(wk, "wk"), _f32c(wv, "wv")
        ln_g, ln_b, pos = _f32c(ln_g, "ln_g"), _f32c(ln_b, "ln_b"), _f32c(pos, "pos")
        N, D = q.shape
        L, H = spec.L, int(num_heads)
        if N != spec.N:
            raise ValueError("query batch and window batch differ")
        need_grad = any(ctx.needs_input_grad[:6])
        ctx.set_materialize_grads(False)      # the attention weights are an output nobody differentiates: no zero tensor for them
        dev = q.device
        out = torch.empty((N, D), dtype=torch.float32, device=dev)
        att = torch.empty((N, H, L), dtype=torch.float32, device=dev)
        k_save = v_save = None
        if need_grad:
            k_save = torch.empty((N, L, D), dtype=torch.float32, device=dev)
            v_save = torch.empty((N, L, D), dtype=torch.float32, device=dev)
        ln_stats = torch.empty((N, L, 2), dtype=torch.float32, device=dev) if ln_g is not None else None
        pidx = spec.pidx if pos is not None else None
        rc = lib.etm_mha_fwd(spec.block_ptr(block), spec.ep_stride, spec.row_stride, _ptr(spec.ep), _ptr(spec.win), _ptr(pidx),
                             _ptr(spec.mask), _ptr(pos), _ptr(ln_g), _ptr(ln_b), float(ln_eps), _ptr(q), _ptr(wk), _ptr(wv),
                             _ptr(out), _ptr(att), _ptr(k_save), _ptr(v_save), _ptr(ln_stats), N, L, D, H, _stream())
        _lib.check(rc, "etm_mha_fwd")
        if need_grad:
            ctx.spec, ctx.block, ctx.H = spec, block, H
            ctx.has_ln, ctx.has_pos = ln_g is not None, pos is not None
            saved = [q, wk, wv, att, k_save, v_save]
            if ln_g is not None:
                saved += [ln_g, ln_b, ln_stats]
            if pos is not None:
                saved += [pos]
            ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(att)
        return out, att

    @staticmethod
    def backward(ctx, d_out, _d_att):
        if d_out is None:
            return (None,) * 10
        lib = _lib.load()
        spec, block, H = ctx.spec, ctx.block, ctx.H
        saved = list(ctx.saved_tensors)
        q, wk, wv, att, k_save, v_save = saved[:6]
        rest = saved[6:]
        ln_g = ln_b = ln_stats = pos = None
        if ctx.has_ln:
            ln_g, ln_b, ln_stats = rest[:3]
            rest = rest[3:]
        if ctx.has_pos:
            pos = rest[0]
        N, D = q.shape
        L = spec.L
        dev = q.device
        d_out = _f32c(d_out, "d_ctx")
        d_q = torch.empty_like(q)
        d_e = torch.empty((N, H, L), dtype=torch.float32, device=dev)
        d_wk = torch.empty_like(wk)
        d_wv = torch.empty_like(wv)
        want_ln = ctx.has_ln and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        want_pos = ctx.has_pos and ctx.needs_input_grad[5]
        d_ln_g = torch.zeros_like(ln_g) if want_ln else None
        d_ln_b = torch.zeros_like(ln_b) if want_ln else None
        d_pos = torch.zeros_like(pos) if want_pos else None
        nbytes = lib.etm_mha_bwd_workspace_bytes(N, L, D)
        ws = workspace(nbytes, dev, "mha_bwd")
        pidx = spec.pidx if pos is not None else None
        rc = lib.etm_mha_bwd(spec.block_ptr(block), spec.ep_stride, spec.row_stride, _ptr(spec.ep), _ptr(spec.win), _ptr(pidx),
                             _ptr(spec.mask), _ptr(pos), _ptr(ln_g), _ptr(ln_b), _ptr(q), _ptr(wk), _ptr(wv), _ptr(att),
                             _ptr(k_save), _ptr(v_save), _ptr(ln_stats), _ptr(d_out), _ptr(d_q), _ptr(d_e), _ptr(d_wk), _ptr(d_wv),
                             _ptr(d_ln_g), _ptr(d_ln_b), _ptr(d_pos), _ptr(ws), nbytes, N, L, D, H, _stream())
        _lib.check(rc, "etm_mha_bwd")
        return d_q, d_wk, d_wv, d_ln_g, d_ln_b, d_pos, None, None, None, None


class _WindowFn(torch.autograd.Function):
    """Folded window attention pass: u [H,N,D] -> (z [H,N,D], att [N,H,L]); see csrc/window_attn.hip."""

    @staticmethod
    def forward(ctx, u, ln_g, ln_b, pos, spec, block, ln_eps):
        lib = _lib.load()
        _need_dev(u, ln_g, ln_b, pos)
        u = _f32c(u, "u")
        ln_g_param, ln_b_param = ln_g, ln_b
        ln_g, ln_b, pos = _f32c(ln_g, "ln_g"), _f32c(ln_b, "ln_b"), _f32c(pos, "pos")
        H, N, D = u.shape
        L = spec.L
        if N != spec.N:
            raise ValueError("query batch and window batch differ")
        ctx.set_materialize_grads(False)      # (no zero tensor for the gradient of the attention weights)
        dev = u.device
        att = torch.empty((N, H, L), dtype=torch.float32, device=dev)
        z = torch.empty((H, N, D), dtype=torch.float32, device=dev)
        ln_stats, stats_ready = None, 0
        if ln_g is not None:
            ln_stats = spec.window_stats(block) if pos is None else None      # (row_stats were taken with the positional rows included)
            stats_ready = int(ln_stats is not None)
            if ln_stats is None:
                ln_stats = torch.empty((N, L, 2), dtype=torch.float32, device=dev)
        pidx = spec.pidx if pos is not None else None
        rc = lib.etm_window_fwd(spec.block_ptr(block), spec.ep_stride, spec.row_stride, _ptr(spec.ep), _ptr(spec.win), _ptr(pidx),
                                _ptr(spec.mask), _ptr(pos), _ptr(ln_g), _ptr(ln_b), float(ln_eps), _ptr(u), N * D, D, _ptr(att),
                                _ptr(z), N * D, D, _ptr(ln_stats), stats_ready, N, L, D, H, _stream())
        _lib.check(rc, "etm_window_fwd")
        if any(ctx.needs_input_grad[:4]):
            ctx.spec, ctx.block = spec, block
            ctx.has_ln, ctx.has_pos = ln_g is not None, pos is not None
            ctx.ln_params = (ln_g_param, ln_b_param) if ln_g is not None else None      # the parameters themselves (arena views by address)
            saved = [u, att]
            if ln_g is not None:
                saved += [ln_g, ln_b, ln_stats, z]      # (z: norm_kv's gradients are formed from the passes' outputs, round 6)
            if pos is not None:
                saved += [pos]
            ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(att)
        return z, att

    @staticmethod
    def backward(ctx, gz, _d_att):
        if gz is None:
            return (None,) * 7
        lib = _lib.load()
        spec, block = ctx.spec, ctx.block
        saved = list(ctx.saved_tensors)
        u, att = saved[:2]
        rest = saved[2:]
        ln_g = ln_b = ln_stats = pos = None
        z_fwd = None
        if ctx.has_ln:
            ln_g, ln_b, ln_stats, z_fwd = rest[:4]
            rest = rest[4:]
        if ctx.has_pos:
            pos = rest[0]
        H, N, D = u.shape
        L = spec.L
        dev = u.device
        gz = _f32c(gz, "d_z")
        d_e = torch.empty((N, H, L), dtype=torch.float32, device=dev)
        du = torch.empty((H, N, D), dtype=torch.float32, device=dev)
        pidx = spec.pidx if pos is not None else None
        rc = lib.etm_window_bwd(spec.block_ptr(block), spec.ep_stride, spec.row_stride, _ptr(spec.ep), _ptr(spec.win), _ptr(pidx),
                                _ptr(spec.mask), _ptr(pos), _ptr(ln_g), _ptr(ln_b), _ptr(ln_stats), _ptr(att), _ptr(gz), N * D, D,
                                _ptr(d_e), _ptr(du), N * D, D, N, L, D, H, _stream())
        _lib.check(rc, "etm_window_bwd")
        want_ln = ctx.has_ln and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        want_pos = ctx.has_pos and ctx.needs_input_grad[3]
        d_ln_g = d_ln_b = d_pos = None
        if _ln_grad_kernel and want_ln and not want_pos and D % 128 == 0 and D <= 512 and H <= 8 and L <= 128:
            # norm_kv's gain / bias gradients as per-workgroup partial rows (csrc/window_ln_grad.hip), summed by the grouped column-sum
            # reduction of the step, or -- without a collector / once it is full -- by the same kernel launched here.  NOT by torch's
            # column sum: its two-stage reduction (a memset node for its semaphore + the reduce kernel) returns wrong sums in some
            # replays of a captured graph on this runtime (profiles/r05/graph_reduce_hazard.txt).
            # Round 6 ("outputs", the default): from the window passes' OUTPUTS -- u, gz, du and the forward's z -- an elementwise
            # pass over four [H, N, D] tensors; round 5 ("rows"): a third pass over the gathered window rows (8 x the bytes).
            rows = lib.etm_window_ln_grad_rows(N)
            partial = torch.empty((rows, 2 * D), dtype=torch.float32, device=dev)
            if _ln_grad_kernel == "rows":
                rc = lib.etm_window_ln_grad(spec.block_ptr(block), spec.ep_stride, spec.row_stride, _ptr(spec.ep), _ptr(spec.win), _ptr(pidx),
                                            _ptr(pos), _ptr(ln_stats), _ptr(att), _ptr(d_e), _ptr(u), _ptr(gz), N * D, D, _ptr(partial),
                                            N, L, D, H, _stream())
                _lib.check(rc, "etm_window_ln_grad")
            else:
                rc = lib.etm_window_ln_grad_from_outputs(_ptr(u), _ptr(gz), _ptr(du), _ptr(z_fwd), _ptr(att), _ptr(d_e), _ptr(ln_g), _ptr(ln_b),
                                                         N * D, D, _ptr(partial), N, L, D, H, _stream())
                _lib.check(rc, "etm_window_ln_grad_from_outputs")
            col = DeferredDw.active
            params = getattr(ctx, "ln_params", None)
            if col is not None and params is not None and col.offer_colsum(partial, rows, 2 * D, [(0, D, params[0].data_ptr()), (D, D, params[1].data_ptr())]):
                return du, None, None, None, None, None, None
            sums = colsum_rows(partial, rows, 2 * D)
            return du, sums[:D], sums[D:], None, None, None, None
        if want_ln or want_pos:
            d_ln_g = torch.zeros_like(ln_g) if want_ln else None
            d_ln_b = torch.zeros_like(ln_b) if want_ln else None
            d_pos = torch.zeros_like(pos) if want_pos else None
            uw = torch.empty((2, N, H, D), dtype=torch.float32, device=dev)
            uw[0].copy_(u.transpose(0, 1))
            uw[1].copy_(gz.transpose(0, 1))
            rc = lib.etm_window_dx(spec.block_ptr(block), spec.ep_stride, spec.row_stride, _ptr(spec.ep), _ptr(spec.win), _ptr(pidx),
                                   _ptr(pos), _ptr(ln_g), _ptr(ln_b), _ptr(ln_stats), _ptr(att), _ptr(d_e), _ptr(uw), _ptr(d_ln_g),
                                   _ptr(d_ln_b), _ptr(d_pos), N, L, D, H, _stream())
            _lib.check(rc, "etm_window_dx")
        return du, d_ln_g, d_ln_b, d_pos, None, None, None


def colsum_rows(partial, P, C):
    """Column sums of the first ``C`` columns of ``partial`` [P, ld] by the library's fixed-order reduction (the summation tree of the
    grouped launch a ``DeferredDw`` collector would have used: same bits with and without a collector)."""
    import ctypes
    out = torch.empty(C, dtype=torch.float32, device=partial.device)
    one = lambda t, v: (t * 1)(v)
    _lib.check(_lib.load().etm_colsum_reduce_grouped(one(ctypes.c_void_p, _ptr(partial)), one(ctypes.c_int32, P), one(ctypes.c_int32, C),
                                                     one(ctypes.c_int32, partial.stride(0)), one(ctypes.c_void_p, _ptr(out)), 1, _stream()),
               "etm_colsum_reduce_grouped")
    return out


_ln_grad_kernel = "outputs"      # norm_kv's gain / bias gradients by csrc/window_ln_grad.hip: "outputs" (from the passes' outputs,
#                                    round 6) or "rows" (a pass over the window rows, round 5); False: the generic dX kernel, etm_window_dx


def set_ln_grad_kernel(on):
    """True / "outputs", "rows", or False (trainer.py: ``fused_ln_grad``)."""
    global _ln_grad_kernel
    if on not in (True, False, "outputs", "rows"):
        raise ValueError(f"fused_ln_grad must be true, false, 'outputs' or 'rows', got {on!r}")
    _ln_grad_kernel = "outputs" if on is True else on


ATTENTION_IMPLS = ("folded", "dense")
_default_impl = "folded"


def set_attention_impl(impl):
    """Select the kernel family behind ``mha``: 'folded' (one HBM-bound pass over the window, default) or 'dense' (the
    K/V projections of the window as fp32-MFMA contractions).  Both compute transformer.py:31-86 for a single query."""
    global _default_impl
    if impl not in ATTENTION_IMPLS:
        raise ValueError(f"attention impl must be one of {ATTENTION_IMPLS}, got {impl!r}")
    _default_impl = impl


def folded_supported(D, L, num_heads):
    hd = D // num_heads
    return D % 32 == 0 and hd % 2 == 0 and ((D <= 512 and L <= 128) or (D <= 1024 and L <= 64))


# ---------------------------------------------------------------------------------------------------------------
# Deferred weight gradients: dW = dy^T x of the dense layers, collected during backward and computed by ONE grouped launch
# (csrc/grouped_dw.hip) straight into the flat gradient arena.  Inactive (no collector) every layer computes its own dW as before.
class DeferredDw:
    """Context manager around ``loss.backward()``.  ``dest``: {parameter.data_ptr(): gradient view [out, in] in the arena}.
    Inside, the autograd functions below hand (dy, x, weight) over instead of multiplying; ``flush()`` (called on exit) runs the
    grouped kernel.  ``written``: data_ptr()s of the parameters whose gradient now sits in its arena view."""
    active = None
    last_flops = 0.0             # 2 * N * Ma * Nb summed over the problems of the most recent grouped launch (bench.py prices it with this)

    def __init__(self, dest):
        self.dest = dest
        self.rows = {}           # parameter data_ptr -> [(first row, rows)] already taken (a second use of the same rows is refused)
        self.items = []          # (A, B, C view, Ma, Nb, lda, ldb, ldc)
        self.colsums = []        # (partial sums, first column, rows P, columns C, row stride, destination view)
        self.conv_wgrads = []    # (pixel slices, slice count, dw view, db view, Cout, C, KH, KW) of the encoder layers
        self.written = set()
        self.N = None

    def __enter__(self):
        DeferredDw.active = self
        return self

    def __exit__(self, *exc):
        DeferredDw.active = None
        if exc[0] is None:
            self.flush()
        return False

    def offer(self, a, b, weight, row0=0, rows=None, a_col0=0):
        """C = weight.grad[row0 : row0 + rows] (all rows if None) = a[:, a_col0 : a_col0 + rows]^T b.  a [N, *], b [N, in] contiguous
        rows.  Returns True when the problem was taken (shape supported, destination known); else the caller multiplies itself."""
        view = self.dest.get(weight.data_ptr())
        if view is None or not (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2 and b.dim() == 2):
            return False
        n, nb = b.shape
        ma = view.shape[0] if rows is None else rows
        if a.stride(1) != 1 or b.stride(1) != 1 or a.shape[0] != n or (self.N is not None and n != self.N):
            return False
        lda, ldb, ldc = a.stride(0), b.stride(0), view.stride(0)
        lib = _lib.load()
        if not lib.etm_grouped_dw_supported(n, ma, nb, lda, ldb, ldc):       # (any number of problems: flush() launches them in chunks)
            return False
        c = view[row0: row0 + ma]
        if view.shape[1] != nb or (_ptr(b) % 16) or (_ptr(c) % 16) or ((_ptr(a) + 4 * a_col0) % 4):
            return False
        # the grouped kernel OVERWRITES its destination: a parameter used twice in one graph (tied weights, a layer applied twice)
        # keeps its first use here and the caller multiplies the second one itself (autograd then accumulates it, _train_body_a adds it)
        taken = self.rows.setdefault(weight.data_ptr(), [])
        if any(row0 < r0 + m and r0 < row0 + ma for r0, m in taken):
            return False
        taken.append((row0, ma))
        self.N = n
        self.items.append((a, b, c, ma, nb, lda, ldb, ldc, a_col0))
        self.written.add(weight.data_ptr())
        return True

    def offer_colsum(self, partial, P, ld, parts):
        """Second stage of column-sum gradients (LayerNorm weight / bias, linear bias): ``partial`` [P, ld] per-workgroup partial sums
        (a buffer of the caller's own, kept alive here), ``parts`` = [(first column, columns, parameter data_ptr)].  True: all
        destinations are known 1-D arena views and the sums will be written by the ONE reduction launch of ``flush()``."""
        views = [self.dest.get(ptr) for _, _, ptr in parts]
        if any(v is None or v.dim() != 1 or not v.is_contiguous() for v in views):
            return False
        if len(self.colsums) + len(parts) > _lib.load().etm_colsum_reduce_max_problems():
            return False
        for (c0, cols, ptr), v in zip(parts, views):
            if v.numel() != cols or ptr in self.written:
                return False
        for (c0, cols, ptr), v in zip(parts, views):
            self.colsums.append((partial, c0, P, cols, ld, v))
            self.written.add(ptr)
        return True

    def flush(self):
        import ctypes
        if self.conv_wgrads:
            k = len(self.conv_wgrads)
            vp = lambda j: (ctypes.c_void_p * k)(*[_ptr(it[j]) for it in self.conv_wgrads])
            ia = lambda j: (ctypes.c_int32 * k)(*[it[j] for it in self.conv_wgrads])
            _lib.check(_lib.load().etm_conv_wgrad_reduce_grouped(vp(0), ia(1), vp(2), vp(3), ia(4), ia(5), ia(6), ia(7), k, _stream()),
                       "etm_conv_wgrad_reduce_grouped")
            self.conv_wgrads = []
        if self.colsums:
            k = len(self.colsums)
            pp = (ctypes.c_void_p * k)(*[_ptr(it[0]) + 4 * it[1] for it in self.colsums])
            po = (ctypes.c_void_p * k)(*[_ptr(it[5]) for it in self.colsums])
            iP = (ctypes.c_int32 * k)(*[it[2] for it in self.colsums])
            iC = (ctypes.c_int32 * k)(*[it[3] for it in self.colsums])
            iL = (ctypes.c_int32 * k)(*[it[4] for it in self.colsums])
            _lib.check(_lib.load().etm_colsum_reduce_grouped(pp, iP, iC, iL, po, k, _stream()), "etm_colsum_reduce_grouped")
            self.colsums = []
        if not self.items:
            return
        lib = _lib.load()
        cap = lib.etm_grouped_dw_max_problems()          # problems per launch (the kernel-argument table: 84)
        for lo in range(0, len(self.items), cap):
            items = self.items[lo: lo + cap]
            k = len(items)
            pa = (ctypes.c_void_p * k)(*[_ptr(it[0]) + 4 * it[8] for it in items])
            pb = (ctypes.c_void_p * k)(*[_ptr(it[1]) for it in items])
            pc = (ctypes.c_void_p * k)(*[_ptr(it[2]) for it in items])
            dims = (ctypes.c_int32 * (5 * k))(*[v for it in items for v in it[3:8]])
            _lib.check(lib.etm_grouped_dw(pa, pb, pc, dims, k, self.N, _stream()), "etm_grouped_dw")
        DeferredDw.last_flops = float(sum(2.0 * self.N * it[3] * it[4] for it in self.items))
        self.items = []


def _offer_dw(a, b, weight, **kw):
    col = DeferredDw.active
    return col is not None and col.offer(a, b, weight, **kw)


class _LinearNoBiasFn(torch.autograd.Function):
    """y = x W^T (transformer.py:26-29 queries / fc_out, :115 fc without their epilogues): library GEMMs for y and dx; the weight
    gradient goes to the grouped launch when a DeferredDw collector is active."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return x.mm(weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        dx = g.mm(weight) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1] and not _offer_dw(g, x, weight):
            dw = g.t().mm(x)
        return dx, dw


def linear_nobias(x, weight):
    """F.linear(x, weight) for 2-D fp32 device tensors under autograd, weight gradient deferrable (see DeferredDw)."""
    if torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous():
        return _LinearNoBiasFn.apply(x, weight)
    return torch.nn.functional.linear(x, weight)


class _HeadFoldFn(torch.autograd.Function):
    """u[h] = q_h Wk_h ([N, hd] x [hd, D] per head): q [N, D], wk [D, D] -> u [H, N, D].  The batched library GEMMs read q and write
    dq through per-head strided views (batch stride hd, row stride D), so no head-major copy of q / dq is ever made."""

    @staticmethod
    def forward(ctx, q, wk, H):
        N, D = q.shape
        hd = D // H
        ctx.save_for_backward(q, wk)
        ctx.H = H
        return torch.bmm(q.view(N, H, hd).transpose(0, 1), wk.view(H, hd, D))

    @staticmethod
    def backward(ctx, du):
        q, wk = ctx.saved_tensors
        H = ctx.H
        N, D = q.shape
        hd = D // H
        du = du.contiguous()
        dq = dwk = None
        if ctx.needs_input_grad[0]:
            dq = torch.empty_like(q)
            torch.bmm(du, wk.view(H, hd, D).transpose(1, 2), out=dq.view(N, H, hd).transpose(0, 1))
        if ctx.needs_input_grad[1]:
            # dWk rows of head h = q_h^T du_h: H problems of the grouped launch (A = the head's columns of q, B = its plane of du)
            col = DeferredDw.active
            if col is not None and q.is_contiguous():
                n0 = len(col.items)
                ok = all(col.offer(q, du[h], wk, row0=h * hd, rows=hd, a_col0=h * hd) for h in range(H))
                if not ok:
                    del col.items[n0:]
                    col.written.discard(wk.data_ptr())
            else:
                ok = False
            if not ok:
                dwk = torch.bmm(q.view(N, H, hd).permute(1, 2, 0), du).view(D, D)
        return dq, dwk, None


class _HeadUnfoldFn(torch.autograd.Function):
    """ctx[:, h] = z_h Wv_h^T ([N, D] x [D, hd] per head): z [H, N, D], wv [D, D] -> ctx [N, D], written straight into its
    [N, H * hd] layout through a strided view (no transpose copy forward or backward)."""

    @staticmethod
    def forward(ctx, z, wv, H):
        _, N, D = z.shape
        hd = D // H
        ctx.save_for_backward(z, wv)
        ctx.H = H
        out = torch.empty((N, D), dtype=z.dtype, device=z.device)
        torch.bmm(z, wv.view(H, hd, D).transpose(1, 2), out=out.view(N, H, hd).transpose(0, 1))
        return out

    @staticmethod
    def backward(ctx, g):
        z, wv = ctx.saved_tensors
        H = ctx.H
        _, N, D = z.shape
        hd = D // H
        g = g.contiguous()
        gh = g.view(N, H, hd).transpose(0, 1)                      # [H, N, hd], strided
        dz = torch.bmm(gh, wv.view(H, hd, D)) if ctx.needs_input_grad[0] else None
        dwv = None
        if ctx.needs_input_grad[1]:
            # dWv rows of head h = g_h^T z_h: H problems of the grouped launch
            col = DeferredDw.active
            ok = False
            if col is not None and z.is_contiguous():
                n0 = len(col.items)
                ok = all(col.offer(g, z[h], wv, row0=h * hd, rows=hd, a_col0=h * hd) for h in range(H))
                if not ok:
                    del col.items[n0:]
                    col.written.discard(wv.data_ptr())
            if not ok:
                dwv = torch.bmm(gh.transpose(1, 2), z).view(D, D)
        return dz, dwv, None


def mha(q, wk, wv, spec, block, num_heads, ln_g=None, ln_b=None, pos=None, ln_eps=1e-5, impl=None):
    """Window attention of one block.  q [N,D] projected queries -> (ctx [N,D] before fc_out, attention [N,H,L])."""
    impl = _default_impl if impl is None else impl
    if impl not in ATTENTION_IMPLS:
        raise ValueError(f"attention impl must be one of {ATTENTION_IMPLS}, got {impl!r}")
    N, D = q.shape
    H = int(num_heads)
    if impl == "dense" or not folded_supported(D, spec.L, H):
        return _MhaFn.apply(q, wk, wv, ln_g, ln_b, pos, spec, block, num_heads, ln_eps)
    hd = D // H
    # u[h] = q_h Wk_h and ctx_h = z_h Wv_h^T: [N,hd] x [hd,D] per head (library GEMMs; autograd supplies d q, d Wk, d Wv)
    if q.is_cuda and q.is_contiguous() and wk.is_contiguous() and wv.is_contiguous():
        u = _HeadFoldFn.apply(q, wk, H)
        z, att = _WindowFn.apply(u, ln_g, ln_b, pos, spec, block, ln_eps)
        return _HeadUnfoldFn.apply(z, wv, H), att
    u = torch.bmm(q.view(N, H, hd).transpose(0, 1), wk.view(H, hd, D))
    z, att = _WindowFn.apply(u, ln_g, ln_b, pos, spec, block, ln_eps)
    ctx = torch.bmm(z, wv.view(H, hd, D).transpose(1, 2)).transpose(0, 1).reshape(N, D)
    return ctx, att


def attn_cached(q, kv_spec, block, num_heads, want_att=False):
    """Inference-only window attention over cached projections.  ``kv_spec`` addresses a cache [W, T, blocks, 2D]
    (K | V per row).  q [N, D] projected queries -> ctx [N, D] (and attention [N, H, L] if asked)."""
    lib = _lib.load()
    _need_dev(q)
    if torch.is_grad_enabled() and q.requires_grad:
        raise RuntimeError("attn_cached is the rollout (no-grad) path; training uses ops.mha")
    q = _f32c(q, "q")
    N, D = q.shape
    L, H = kv_spec.L, int(num_heads)
    ctx = torch.empty((N, D), dtype=torch.float32, device=q.device)
    att = torch.empty((N, H, L), dtype=torch.float32, device=q.device) if want_att else None
    rc = lib.etm_attn_cached(kv_spec.block_ptr(block), kv_spec.ep_stride, kv_spec.row_stride, _ptr(kv_spec.ep), _ptr(kv_spec.win),
                             _ptr(kv_spec.mask), _ptr(q), _ptr(ctx), _ptr(att), N, L, D, H, _stream())
    _lib.check(rc, "etm_attn_cached")
    return ctx, att


def reset_rows(dst, init, step):
    """dst[w] = init for every w with step[w] == 0 (dst [W, ...], init [...], step [W] int64), in place."""
    lib = _lib.load()
    _need_dev(dst, init, step)
    if not dst.is_contiguous() or not init.is_contiguous() or step.dtype != torch.int64:
        raise TypeError("reset_rows needs contiguous float32 tensors and an int64 step vector")
    _lib.check(lib.etm_reset_rows(_ptr(dst), _ptr(init), _ptr(step), dst.shape[0], init.numel(), _stream()), "etm_reset_rows")
    return dst


def rollout_window(step, mask_table, index_table, t_dev, mask_t, win_t, st_mask, st_idx, t_row=None, reset=None, w_off=0,
                   latch=None):
    """Window-table lookup of one rollout step + staging (all outputs preallocated, in place).  Optional riders of the same
    launch: ``t_row`` (int64 scalar tensor) receives the staging row t, ``reset = (cache [W, ...], init [...])`` performs
    ``reset_rows(cache, init, step)``, ``latch = (ss [2, W], copy [2, W])`` copies the uploaded (episode step, slot) block --
    ``step`` must be ``ss[0]`` -- into the buffer the tail of the step indexes.  Worker groups: ``mask_t`` / ``win_t`` / ``step`` (and the cache) cover the W workers of
    the group, the staging arrays ``st_mask`` / ``st_idx`` [S, W_total, L] all of them; ``w_off`` is the group's first worker."""
    lib = _lib.load()
    W, L = win_t.shape
    stage_w = st_idx.shape[1]
    rdst = rinit = None
    relems = 0
    if reset is not None:
        rdst, rinit = reset
        if not rdst.is_contiguous() or not rinit.is_contiguous() or rdst.shape[0] != W:
            raise TypeError("reset needs a contiguous cache [W, ...] and a contiguous initial row")
        relems = rinit.numel()
    lat = None
    if latch is not None:
        ss, lat = latch
        if (ss.shape != (2, W) or lat.shape != (2, W) or not ss.is_contiguous() or not lat.is_contiguous()
                or ss.dtype != torch.int64 or lat.dtype != torch.int64 or step.data_ptr() != ss.data_ptr()):
            raise TypeError("latch needs contiguous int64 [2, W] blocks and step = ss[0]")
    _lib.check(lib.etm_rollout_window(_ptr(step), _ptr(mask_table), _ptr(index_table), _ptr(t_dev), _ptr(mask_t), _ptr(win_t),
                                      st_mask.data_ptr() + w_off * L * st_mask.element_size(),
                                      st_idx.data_ptr() + w_off * L * st_idx.element_size(), _ptr(t_row), _ptr(lat), _ptr(rdst), _ptr(rinit),
                                      relems, W, L, stage_w, _stream()), "etm_rollout_window")


def rollout_sample(logits, value, uniforms, forced, t_dev, actions, st_actions, st_logp, st_values):
    """Categorical sampling + staging of one rollout step for a single-branch policy (in place; increments t_dev).
    ``forced`` (optional): time-major int64 table [S, W]; entries >= 0 replace the sample of that (step, worker)."""
    lib = _lib.load()
    W, A = logits.shape
    logits, value = _f32c(logits, "logits"), _f32c(value, "value")
    _lib.check(lib.etm_rollout_sample(_ptr(logits), _ptr(value), _ptr(uniforms), _ptr(forced), _ptr(t_dev), _ptr(actions),
                                      _ptr(st_actions), _ptr(st_logp), _ptr(st_values), W, A, _stream()), "etm_rollout_sample")


_policy_sync = {}


def rollout_policy(h2, policy_head, value_head, uniforms, forced, t_dev, actions, st_actions, st_logp, st_values,
                   host_actions=None, host_flag=None, h_bias=None, w_off=0):
    """``rollout_heads`` + ``rollout_sample`` in one launch (single-branch policy); ``host_actions`` / ``host_flag``: pinned
    int64 tensors that receive the actions and then the incremented step counter (the host spins on the flag).  ``forced``
    (optional): time-major int64 table [S, W_total]; entries >= 0 replace the sample of that (step, worker)."""
    lib = _lib.load()
    W, A = h2.shape[0], policy_head.weight.shape[0]
    hid = h2.shape[1] // 2
    h2 = _f32c(h2, "h")
    ha = 0 if host_actions is None else host_actions.data_ptr()
    hf = 0 if host_flag is None else host_flag.data_ptr()
    sync = _policy_sync.get(t_dev.data_ptr())       # arrival counter of the launch's workgroups, one per step counter
    if sync is None:
        sync = _policy_sync[t_dev.data_ptr()] = torch.zeros(1, dtype=torch.int32, device=h2.device)
    # worker groups: h2 / actions / forced / host_actions cover this group's W workers; uniforms and the staging arrays are
    # [S, W_total(, 1)] and are addressed from the group's first worker ``w_off``
    stage_w = st_values.shape[1]
    off = lambda t: None if t is None else t.data_ptr() + w_off * t.element_size()
    _lib.check(lib.etm_rollout_policy(_ptr(h2), _ptr(h_bias), _ptr(policy_head.weight), _ptr(policy_head.bias), _ptr(value_head.weight),
                                      _ptr(value_head.bias), off(uniforms), off(forced), _ptr(t_dev), _ptr(actions), off(st_actions),
                                      off(st_logp), off(st_values), ha, hf, _ptr(sync), W, A, hid, stage_w, _stream()),
               "etm_rollout_policy")


def rollout_trxl_group_ok(fused_group, W, L, hid, A):
    """Does the group form of the step kernel (etm_rollout_trxl_group, csrc/rollout_group.hip) take a worker group of W workers of
    this model?  ``fused_group``: ``ActorCriticModel._rfg`` (None: the model has no group packings)."""
    if fused_group is None:
        return False
    return bool(_lib.load().etm_rollout_trxl_group_supported(fused_group["D"], fused_group["H"], L, hid, A, fused_group["nb"], W, fused_group["gtrxl"]))


def rollout_trxl_scratch(W, D, H, nb, device, group=False):
    """Zeroed scratch of one worker group for ``rollout_trxl`` (launch counter, error word, exchange slots); ``group``: for the
    group form of the kernel."""
    lib = _lib.load()
    nbytes = lib.etm_rollout_trxl_group_scratch_bytes(nb) if group else lib.etm_rollout_trxl_scratch_bytes(W, D, H, nb)
    return torch.zeros((nbytes + 7) // 8, dtype=torch.int64, device=device)


def rollout_trxl_error(scratch):
    """Non-zero if a team member of the last launches timed out waiting for a partner (device scalar; no synchronisation)."""
    return scratch[1]


def rollout_trxl_clear_error(scratch):
    """Reset the error word of the step kernel's scratch area (after the caller has dealt with a reported time-out)."""
    scratch[1].zero_()


def rollout_trxl_supported(D, H, L, hid, A, nb):
    """Does ``rollout_trxl`` handle these shapes (etm_rollout_trxl_supported)?"""
    return bool(_lib.load().etm_rollout_trxl_supported(D, H, L, hid, A, nb))


def rollout_trxl(h_in, fused, kv, win_t, mask_t, items, policy_head, value_head, uniforms, forced, t_dev, actions, st_actions, st_logp,
                 st_values, scratch, host_actions=None, host_flag=None, w_off=0, tail=None, h_bias=None, window=None):
    """Transformer + hidden / output heads + sampling of one rollout step of a worker group in one launch (etm_rollout_trxl).
    ``fused``: the transposed fixed-address weight copies of ``ActorCriticModel.refresh_rollout_weights`` (dict with the host
    pointer table ``blocks``); ``kv`` the group's K | V cache [W, T, blocks, 2D]; ``scratch`` from ``rollout_trxl_scratch``; the
    staging arguments as in ``rollout_policy``.  ``window`` = (ss, mask_table, index_table, st_mask, st_idx, latch, t_row, kv_init):
    the launch does the step's window lookup (and the cache reset of workers at episode step 0) itself -- no ``rollout_window``
    in front of it.  ``tail`` = (wkv [blocks, D, 2D], pos [T, D] or None, step_l [W], slot_l [W],
    bank [slots, T, blocks, D]): after the action hand-over the same launch writes the new memory items into
    ``bank[slot_l, step_l]`` and their K | V projection into ``kv[w, step_l]``."""
    lib = _lib.load()
    h_in = _f32c(h_in, "h_in")
    h_splits = 0
    if h_bias is not None:        # h_in = [splits, W, D] slice sums of rollout_hidden_partial
        h_splits = h_in.shape[0]
        h_in_shape = h_in.shape[1:]
    else:
        h_in_shape = h_in.shape
    w_args = (0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)
    if window is not None:        # (ss [2, W], mask_table, index_table, st_mask, st_idx, latch [2, W], t_row, kv_init or None)
        ss, mask_table, index_table, st_mask, st_idx, latch, t_row, kv_init = window
        Lw = win_t.shape[1]
        w_args = (_ptr(ss), _ptr(mask_table), _ptr(index_table), st_mask.data_ptr() + w_off * Lw * st_mask.element_size(),
                  st_idx.data_ptr() + w_off * Lw * st_idx.element_size(), _ptr(latch), _ptr(t_row), _ptr(mask_t), _ptr(win_t),
                  0 if kv_init is None else _ptr(kv_init), index_table.shape[0])
    t_args = (0, 0, 0, 0, 0, 0, 0, 0)
    if tail is not None:
        wkv, pos, step_l, slot_l, bank = tail
        if not (wkv.is_contiguous() and bank.stride(3) == 1 and (pos is None or pos.is_contiguous())):
            raise ValueError("rollout_trxl tail: wkv / pos contiguous and a bank with contiguous feature rows expected")
        # bank [slots, T, blocks, D] by STRIDES: the episode bank is block-major in memory (buffer.py), upstream's layout as a view
        t_args = (_ptr(wkv), 0 if pos is None else _ptr(pos), _ptr(step_l), _ptr(slot_l), _ptr(bank), bank.stride(0), bank.stride(1),
                  bank.stride(2))
    W, D = h_in_shape
    L = win_t.shape[1]
    A, hid = policy_head.weight.shape
    sync = _policy_sync.get(t_dev.data_ptr())
    if sync is None:
        sync = _policy_sync[t_dev.data_ptr()] = torch.zeros(1, dtype=torch.int32, device=h_in.device)
    stage_w = st_values.shape[1]
    off = lambda t: None if t is None else t.data_ptr() + w_off * t.element_size()
    ha = 0 if host_actions is None else host_actions.data_ptr()
    hf = 0 if host_flag is None else host_flag.data_ptr()
    # ``fused`` with the group packings (ActorCriticModel._rfg, "group": True) selects the group form of the kernel: same arguments
    entry, name = (lib.etm_rollout_trxl_group, "etm_rollout_trxl_group") if fused.get("group") else (lib.etm_rollout_trxl, "etm_rollout_trxl")
    _lib.check(entry(_ptr(h_in), _ptr(fused["emb_t"]), _ptr(fused["emb_b"]), fused["blocks"], fused["nb"], _ptr(kv), kv.stride(0),
                                    kv.stride(1), _ptr(win_t), _ptr(mask_t), _ptr(items), _ptr(fused["heads_t"]), _ptr(fused["heads_b"]),
                                    _ptr(policy_head.weight), _ptr(policy_head.bias), _ptr(value_head.weight), _ptr(value_head.bias),
                                    off(uniforms), off(forced), _ptr(t_dev), _ptr(actions), off(st_actions), off(st_logp), off(st_values),
                                    ha, hf, _ptr(sync), float(fused["eps"]), _ptr(scratch), scratch.numel() * 8, *t_args,
                                    0 if h_bias is None else _ptr(h_bias), h_splits, *w_args, int(fused.get("pre_ln", 0)),
                                    int(fused.get("gtrxl", 0)), W, D, fused["H"], L, hid, A, stage_w, _stream()),
               name)


def gather_rows(fields, idx):
    """``[t.index_select(0, idx) for t in fields]`` in one launch (etm_gather_rows): the per-sample fields of a minibatch.
    Tensors whose rows are not a multiple of 4 bytes (or not contiguous) go through index_select."""
    import ctypes
    lib = _lib.load()
    n = idx.numel()
    outs = [None] * len(fields)
    sel = []
    for k, t in enumerate(fields):
        row = t[0].numel() * t.element_size() if t.shape[0] > 0 else 0
        if t.is_contiguous() and 0 < row <= 4096 and row % 4 == 0 and t.shape[0] == fields[0].shape[0] and len(sel) < 16:
            outs[k] = torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            sel.append((k, row))
        else:
            outs[k] = t.index_select(0, idx)
    if sel:
        m = len(sel)
        src = (ctypes.c_void_p * m)(*[fields[k].data_ptr() for k, _ in sel])
        dst = (ctypes.c_void_p * m)(*[outs[k].data_ptr() for k, _ in sel])
        rb = (ctypes.c_int64 * m)(*[r for _, r in sel])
        _lib.check(lib.etm_gather_rows(src, dst, rb, m, _ptr(idx), n, fields[sel[0][0]].shape[0], _stream()), "etm_gather_rows")
    return outs


def rollout_hidden_partial(x, wt, out=None):
    """K-slice partial sums of ``x [W, F] @ wt [F, D]`` (etm_rollout_hidden_partial): [splits, W, D]; the consumer
    (``rollout_trxl(h_bias=...)``) adds the slices, the bias and the ReLU."""
    lib = _lib.load()
    x = _f32c(x, "features")
    W, F = x.shape
    D = wt.shape[1]
    splits = lib.etm_rollout_hidden_splits(F)
    if splits <= 0:
        raise ValueError(f"rollout_hidden_partial: unsupported feature size {F}")
    if out is None:
        out = torch.empty((splits, W, D), dtype=torch.float32, device=x.device)
    _lib.check(lib.etm_rollout_hidden_partial(_ptr(x), _ptr(wt), _ptr(out), W, F, D, _stream()), "etm_rollout_hidden_partial")
    return out


def rollout_conv3_hidden_supported(conv3, hi, wi, d):
    return bool(_lib.load().etm_rollout_conv3_hidden_supported(conv3.in_channels, hi, wi, conv3.out_channels, conv3.kernel_size[0],
                                                               conv3.kernel_size[1], conv3.stride[0], d))


def rollout_conv3_hidden(x2, w3k, b3, hid_t, out=None):
    """Last encoder layer + lin_hidden's partial sums of a rollout step in one launch (etm_rollout_conv3_hidden): ``x2`` [W, Hi, Wi, 64]
    NHWC, ``w3k`` [576, 64], ``hid_t`` [64 * Ho * Wo, D] -> [Ho * Wo, W, D] (``rollout_trxl(h_bias=...)`` adds the rows)."""
    lib = _lib.load()
    x2 = _f32c(x2, "x2")
    W, hi, wi, _ = x2.shape
    D = hid_t.shape[1]
    npix = (hi - 2) * (wi - 2)
    if out is None:
        out = torch.empty((npix, W, D), dtype=torch.float32, device=x2.device)
    _lib.check(lib.etm_rollout_conv3_hidden(_ptr(x2), _ptr(w3k), _ptr(b3), _ptr(hid_t), _ptr(out), W, hi, wi, D, _stream()),
               "etm_rollout_conv3_hidden")
    return out


def rollout_heads(h2, branch, value_head):
    """logits [W,A] and value [W] from h2 = [relu(lin_policy(h)) | relu(lin_value(h))] ([W, 2*hid]); no-grad path."""
    lib = _lib.load()
    h2 = _f32c(h2, "h2")
    W, hid = h2.shape[0], h2.shape[1] // 2
    A = branch.weight.shape[0]
    logits = torch.empty((W, A), dtype=torch.float32, device=h2.device)
    value = torch.empty((W,), dtype=torch.float32, device=h2.device)
    _lib.check(lib.etm_rollout_heads(_ptr(h2), _ptr(branch.weight), _ptr(branch.bias), _ptr(value_head.weight), _ptr(value_head.bias),
                                     _ptr(logits), _ptr(value), W, A, hid, _stream()), "etm_rollout_heads")
    return logits, value


def gru_gate(x, y, wy, ux, ug, bg):
    """GTrXL gate on the no-grad path: wy = [Wr;Wz;Wg] ([3D,D]), ux = [Ur;Uz] ([2D,D]), ug [D,D], bg [D].
    Three library GEMMs + two elementwise kernels."""
    lib = _lib.load()
    x, y = _f32c(x, "x"), _f32c(y, "y")
    N, D = x.shape
    a = torch.nn.functional.linear(y, wy)
    b = torch.nn.functional.linear(x, ux)
    rx = torch.empty_like(x)
    z = torch.empty_like(x)
    _lib.check(lib.etm_gru_gate_rz(_ptr(a), _ptr(b), _ptr(bg), _ptr(x), _ptr(rx), _ptr(z), N, D, _stream()), "etm_gru_gate_rz")
    c = torch.nn.functional.linear(rx, ug)
    out = torch.empty_like(x)
    _lib.check(lib.etm_gru_gate_out(_ptr(a), _ptr(c), _ptr(z), _ptr(x), _ptr(out), N, D, _stream()), "etm_gru_gate_out")
    return out


def add_layernorm(a, b, norm, out=None, bias=None, relu=False):
    """LayerNorm(act(a + bias) + b) with ``norm``'s affine parameters; forward only (rollout path).  ``bias`` / ``relu`` fold
    the epilogue of the linear layer that produced ``a`` (which then runs as a plain, per-shape tuned library GEMM)."""
    lib = _lib.load()
    _need_dev(a, b)
    a, b = _f32c(a, "a"), _f32c(b, "b")
    N, D = a.shape
    if out is None:
        out = torch.empty_like(a)
    elif out.shape != a.shape or out.dtype != torch.float32 or not out.is_contiguous():
        raise TypeError("add_layernorm: out must be a contiguous float32 tensor of the input shape")
    _lib.check(lib.etm_add_layernorm(_ptr(a), _ptr(bias), 1 if relu else 0, _ptr(b), _ptr(norm.weight), _ptr(norm.bias), float(norm.eps),
                                     _ptr(out), N, D, _stream()), "etm_add_layernorm")
    return out


class _FusedLayerNormFn(torch.autograd.Function):
    """y = LayerNorm(act(a + bias) + res) * gamma + beta with hand-written forward and backward (csrc/block_train.hip)."""

    @staticmethod
    def forward(ctx, a, bias, res, gamma, beta, relu, eps, fork=False):
        lib = _lib.load()
        _need_dev(a, bias, res, gamma, beta)
        a, bias, res = _f32c(a, "a"), _f32c(bias, "bias"), _f32c(res, "res")
        gamma, beta = _f32c(gamma, "gamma"), _f32c(beta, "beta")
        N, D = a.shape
        need = any(ctx.needs_input_grad[:5])
        y = torch.empty_like(a)
        s = torch.empty_like(a) if need else None
        stats = torch.empty((N, 2), dtype=torch.float32, device=a.device) if need else None
        _lib.check(lib.etm_ln_train_fwd(_ptr(a), _ptr(bias), 1 if relu else 0, _ptr(res), _ptr(gamma), _ptr(beta), float(eps), _ptr(y),
                                        _ptr(s), _ptr(stats), N, D, _stream()), "etm_ln_train_fwd")
        if need:
            ctx.relu, ctx.has_bias, ctx.has_res = bool(relu), bias is not None, res is not None
            ctx.param_ptrs = (gamma.data_ptr(), beta.data_ptr(), bias.data_ptr() if bias is not None else 0)
            ctx.save_for_backward(s, stats, gamma, a if relu else None, bias if relu else None)
        ctx.fork = bool(fork)
        if fork:         # the output twice (one storage): the gradients of its two consumers arrive separately and are added on load
            ctx.set_materialize_grads(False)
            return y, y.detach()
        return y

    @staticmethod
    def backward(ctx, dy, dy2=None):
        lib = _lib.load()
        s, stats, gamma, a, bias = ctx.saved_tensors
        if dy is None:
            dy, dy2 = dy2, None
        if dy is None:
            return (None,) * 8
        dy = _f32c(dy, "dy")
        dy2 = _f32c(dy2, "dy2")
        N, D = s.shape
        ds = torch.empty_like(s)
        da = torch.empty_like(s) if ctx.relu else None
        nbytes = lib.etm_ln_train_bwd_workspace_bytes(N, D)
        d_a = da if ctx.relu else ds
        col = DeferredDw.active
        if col is not None:
            # the three column sums go to the collector's ONE reduction launch, straight into the parameters' arena views
            g_ptr, b_ptr, bias_ptr = ctx.param_ptrs
            parts = [(0, D, g_ptr), (D, D, b_ptr)] + ([(2 * D, D, bias_ptr)] if ctx.has_bias else [])
            part = torch.empty(nbytes // 4, dtype=torch.float32, device=s.device)
            if col.offer_colsum(part, lib.etm_ln_train_bwd_partial_rows(N), 3 * D, parts):
                _lib.check(lib.etm_ln_train_bwd(_ptr(dy), _ptr(dy2), _ptr(s), _ptr(stats), _ptr(gamma), _ptr(a), _ptr(bias), 1 if ctx.relu else 0,
                                                _ptr(ds), _ptr(da), None, _ptr(part), nbytes, N, D, _stream()), "etm_ln_train_bwd")
                return d_a, None, (ds if ctx.has_res else None), None, None, None, None, None
        sums = torch.empty((3, D), dtype=torch.float32, device=s.device)
        ws = workspace(nbytes, s.device, "ln_bwd")
        _lib.check(lib.etm_ln_train_bwd(_ptr(dy), _ptr(dy2), _ptr(s), _ptr(stats), _ptr(gamma), _ptr(a), _ptr(bias), 1 if ctx.relu else 0, _ptr(ds),
                                        _ptr(da), _ptr(sums), _ptr(ws), nbytes, N, D, _stream()), "etm_ln_train_bwd")
        return d_a, (sums[2] if ctx.has_bias else None), (ds if ctx.has_res else None), sums[0], sums[1], None, None, None


def fused_layernorm(a, norm, bias=None, res=None, relu=False, fork=False):
    """``norm(act(a + bias) + res)`` (``norm``: an nn.LayerNorm over the last dimension of the [N, D] input) as one forward and
    one backward kernel (+ a tiny fixed-order column-sum kernel): the bias / ReLU of the linear layer that produced ``a`` and the
    residual add ride along, so that layer runs as a plain GEMM.  Training path (differentiable); the rollout uses
    ``add_layernorm``.  ``fork``: returns the result TWICE (two tensors, one storage) for its two consumers -- the next GEMM and
    the next residual branch (transformer.py:143-149, :160-170) -- so that their gradients reach the backward kernel separately
    and are added on load, not by a launch of autograd's."""
    if fork:
        return _FusedLayerNormFn.apply(a, bias, res, norm.weight, norm.bias, relu, norm.eps, True)
    return _FusedLayerNormFn.apply(a, bias, res, norm.weight, norm.bias, relu, norm.eps)


class _GruGateFn(torch.autograd.Function):
    """GTrXL gate (transformer.py:287-298): out = (1 - z) x + z tanh(Wg y + Ug (r x)), r = sigmoid(Wr y + Ur x),
    z = sigmoid(Wz y + Uz x - bg).  Three concatenated library GEMMs + two kernels forward; six GEMMs + two kernels (+ the
    column-sum kernel for d bg) backward, instead of six small GEMMs + ~10 element-wise launches forward and twice that backward."""

    @staticmethod
    def forward(ctx, x, y, wr, ur, wz, uz, wg, ug, bg, wy=None, ux=None, fork=False):
        lib = _lib.load()
        _need_dev(x, y, wr, ur, wz, uz, wg, ug, bg)
        x, y = _f32c(x, "x"), _f32c(y, "y")
        N, D = x.shape
        if wy is None or ux is None:                   # (the caller may hand the concatenations over: transformer.py:_pack_gate_weights)
            wy = torch.cat((wr, wz, wg), dim=0)        # [3D, D]
            ux = torch.cat((ur, uz), dim=0)            # [2D, D]
        A = torch.mm(y, wy.t())
        B = torch.mm(x, ux.t())
        r, z, rx = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        _lib.check(lib.etm_gate_train_rz(_ptr(A), _ptr(B), _ptr(bg), _ptr(x), _ptr(r), _ptr(z), _ptr(rx), N, D, _stream()), "etm_gate_train_rz")
        C = torch.mm(rx, ug.t())
        hh, out = torch.empty_like(x), torch.empty_like(x)
        _lib.check(lib.etm_gate_train_out(_ptr(A), _ptr(C), _ptr(z), _ptr(x), _ptr(hh), _ptr(out), N, D, _stream()), "etm_gate_train_out")
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x, y, r, z, rx, hh, wy, ux, ug)
            ctx.gate_weights = (wr, ur, wz, uz, wg)      # (the parameters themselves: DeferredDw looks their arena views up by address)
            ctx.bg_ptr = bg.data_ptr()
        if fork:         # the output twice (one storage) for its two consumers (pre-LN blocks: the next LayerNorm and the next gate's
            ctx.set_materialize_grads(False)   # residual input): their gradients arrive separately and gate_bwd1 adds them on load
            return out, out.detach()
        return out

    @staticmethod
    def backward(ctx, dout, dout2=None):
        lib = _lib.load()
        x, y, r, z, rx, hh, wy, ux, ug = ctx.saved_tensors
        if dout is None:
            dout, dout2 = dout2, None
        if dout is None:
            return (None,) * 12
        dout = _f32c(dout, "dout")
        dout2 = _f32c(dout2, "dout2")
        N, D = x.shape
        dev = x.device
        dA = torch.empty((N, 3 * D), dtype=torch.float32, device=dev)
        dB = torch.empty((N, 2 * D), dtype=torch.float32, device=dev)
        dx1, dbg = torch.empty_like(x), None
        nbytes = lib.etm_gate_train_bwd_workspace_bytes(N, D)
        col = DeferredDw.active
        part = torch.empty(nbytes // 4, dtype=torch.float32, device=dev) if col is not None else None
        if col is not None and col.offer_colsum(part, lib.etm_gate_train_bwd_partial_rows(N), D, [(0, D, ctx.bg_ptr)]):
            ws = part                                   # d bg's second stage rides in the collector's one reduction launch
        else:
            dbg = torch.empty((D,), dtype=torch.float32, device=dev)
            ws = workspace(nbytes, dev, "gate_bwd")
        _lib.check(lib.etm_gate_train_bwd1(_ptr(dout), _ptr(dout2), _ptr(z), _ptr(hh), _ptr(x), _ptr(dA), _ptr(dB), _ptr(dx1), _ptr(dbg), _ptr(ws), nbytes,
                                           N, D, _stream()), "etm_gate_train_bwd1")
        dC = dA[:, 2 * D:]                              # d pre_h, a strided view (row stride 3D): GEMM operand in place
        drx = torch.mm(dC, ug)
        dx2 = torch.empty_like(x)
        _lib.check(lib.etm_gate_train_bwd2(_ptr(drx), _ptr(x), _ptr(r), _ptr(dx1), _ptr(dA), _ptr(dB), _ptr(dx2), N, D, _stream()),
                   "etm_gate_train_bwd2")
        dy = torch.mm(dA, wy)
        dx = dx2.addmm_(dB, ux)                         # (in place: the out-of-place form first copies dx2, a 3 MB device copy per gate)
        # Weight gradients.  Round 5: with a DeferredDw collector active (the trainer's backward pass) the six D x D products go to the
        # grouped fp32-MFMA launch (csrc/grouped_dw.hip) like every other dense layer's -- column blocks of dA / dB as the left
        # operands (a_col0), written straight into the arena views.  Two reasons: 24 library GEMMs fewer per minibatch step at
        # config 5, and accuracy -- the library's [3D, N] x [N, D] product walks the N samples in ONE fp32 chain (measured 2.8e-6 of
        # the tensor norm against float64 at N = 1,686: the worst tensors of the kink-free parity test were the gate matrices), the
        # grouped kernel in four chains summed pairwise (7e-7).
        wr, ur, wz, uz, wg = ctx.gate_weights
        took = [_offer_dw(dA, y, wr, a_col0=0), _offer_dw(dB, x, ur, a_col0=0), _offer_dw(dA, y, wz, a_col0=D),
                _offer_dw(dB, x, uz, a_col0=D), _offer_dw(dA, y, wg, a_col0=2 * D), _offer_dw(dA, rx, ug, a_col0=2 * D)]
        dwy = torch.mm(dA.t(), y) if not (took[0] and took[2] and took[4]) else None     # [3D, D] = d [Wr; Wz; Wg]
        dux = torch.mm(dB.t(), x) if not (took[1] and took[3]) else None                 # [2D, D] = d [Ur; Uz]
        pick = lambda t, full, lo: None if t else full[lo: lo + D]
        return (dx, dy, pick(took[0], dwy, 0), pick(took[1], dux, 0), pick(took[2], dwy, D), pick(took[3], dux, D),
                pick(took[4], dwy, 2 * D), None if took[5] else torch.mm(dC.t(), rx), dbg, None, None, None)


def gru_gate_train(gate, x, y, packed=None, fork=False):
    """Differentiable GTrXL gate of ``gate`` (a transformer.GRUGate) on [N, D] inputs.  ``packed``: ([Wr; Wz; Wg], [Ur; Uz]) of the
    gate's CURRENT weights when the caller has them concatenated already.  ``fork``: returns the result TWICE (two tensors, one
    storage) for its two consumers; their gradients are added by the gate's backward kernel on load instead of by an extra launch."""
    wy, ux = packed if packed is not None else (None, None)
    return _GruGateFn.apply(x, y, gate.Wr.weight, gate.Ur.weight, gate.Wz.weight, gate.Uz.weight, gate.Wg.weight, gate.Ug.weight, gate.bg, wy, ux,
                            fork)


def conv_pack_weights(weight2d):
    """[Cout, K] (K in the order that matches the input layout) -> the MFMA-fragment order etm_conv_relu reads
    (include/etm_hip.h): packed[g, t, half, col, j] = w[t * 32 + col, g * 8 + half * 4 + j]; returned as [Cout, K]."""
    Cout, K = weight2d.shape
    if Cout % 32 or K % 8:
        raise ValueError("conv_pack_weights needs Cout % 32 == 0 and K % 8 == 0")
    w = weight2d.reshape(Cout // 32, 32, K // 8, 2, 4)          # [t, col, g, half, j]
    return w.permute(2, 0, 3, 1, 4).contiguous().view(Cout, K)   # [g, t, half, col, j]


def conv_relu(x, weight2d, bias, C, H, W, KH, KW, S, in_nhwc, out_nchw, index=None, rows=None):
    """relu(conv2d(x) + bias) on the no-grad path (see etm_conv_relu).  ``weight2d``: ``conv_pack_weights`` of the [Cout, K]
    weights in the K order that matches the input layout.  With ``index`` (int64 device scalar) ``x`` is a stack [S, N, ...]
    and the layer reads x[index] -- the row is chosen on the device, so a captured graph can walk a staging array; ``rows =
    (lo, hi)`` restricts it to images lo..hi-1 of that row (a worker group).
    Returns NHWC [N,Ho,Wo,Cout] or NCHW [N,Cout,Ho,Wo]."""
    lib = _lib.load()
    _need_dev(x, weight2d, bias, index)
    x = _f32c(x, "x")
    stride = 0
    if index is not None:
        if index.dtype != torch.int64 or index.numel() != 1:
            raise TypeError("conv_relu: index must be an int64 device scalar")
        stride = x[0].numel()
        N = x.shape[1]
    else:
        N = x.shape[0]
    base = x.data_ptr()
    if rows is not None:          # only images [lo, hi) of every row of the stack (a worker group)
        if index is None:
            raise TypeError("conv_relu: rows needs index (stacked input)")
        lo, hi = rows
        base += lo * x[0, 0].numel() * 4
        N = hi - lo
    Cout = weight2d.shape[0]
    Ho, Wo = (H - KH) // S + 1, (W - KW) // S + 1
    shape = (N, Cout, Ho, Wo) if out_nchw else (N, Ho, Wo, Cout)
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    rc = lib.etm_conv_relu(base, _ptr(index), stride, _ptr(weight2d), _ptr(bias), _ptr(out), N, C, H, W, Cout, KH, KW, S,
                           1 if in_nhwc else 0, 1 if out_nchw else 0, _stream())
    _lib.check(rc, "etm_conv_relu")
    return out


def conv_pack_dgrad_weights(weight, stride):
    """conv weight [Cout, C, KH, KW] -> the S*S stride-parity class blocks etm_conv_train_dgrad reads (include/etm_hip.h):
    class (py, px) packs Wd[c][(a*T + j)*Cout + co] = w[co][c][py + S a][px + S (T-1-j)], T = KH / S."""
    Cout, C, KH, KW = weight.shape
    S = int(stride)
    T = KH // S
    blocks = []
    for py in range(S):
        for px in range(S):
            sub = weight[:, :, py::S, px::S].flip(3)                   # [Cout, C, a, j]
            blocks.append(conv_pack_weights(sub.permute(1, 2, 3, 0).reshape(C, T * T * Cout)))
    return torch.stack(blocks).contiguous()


def encoder_train_supported(obs_shape, convs, batch=None):
    """Can the hand-written training kernels run this encoder?  (model.py:40-56 geometry with 84 x 84 or similar inputs.)
    ``batch``: images per call -- the kernels index output pixels with 24 bits (N * Ho * Wo < 2^24 per layer, padded to the
    backward-data kernel's image unit of 1024) and the source with 32-bit element offsets; larger minibatches take the library path."""
    c, h, w = obs_shape
    for conv in convs:
        if batch is not None:
            kh, s = conv.kernel_size[0], conv.stride[0]
            padded = (batch + 1023) // 1024 * 1024
            if (batch * h * w * c >= 2 ** 31 or batch * ((h - kh) // s + 1) * ((w - kh) // s + 1) >= 2 ** 24      # forward / backward-weight
                    or (conv is not convs[0] and padded * (h // s) * (w // s) >= 2 ** 24)):                      # backward-data
                return False
        kh, kw = conv.kernel_size
        s = conv.stride[0]
        if (kh != kw or conv.stride[0] != conv.stride[1] or conv.padding != (0, 0) or conv.dilation != (1, 1) or conv.groups != 1
                or conv.out_channels not in (32, 64) or (kw * c) % 8 or (w * c) % 4 or (s * c) % 4 or (kh * kw * c) % 32
                or h < kh or w < kw):
            return False
        if conv is not convs[0] and (kh % s or h % s or w % s or c not in (32, 64)):     # backward-data of the layers above the first
            return False
        h, w, c = (h - kh) // s + 1, (w - kw) // s + 1, conv.out_channels
    return True


_encoder_products = "bf16x3"     # default of encoder_train for callers that name none: "bf16x3" = the encoder's products on the bf16 matrix
#                                    pipe at fp32 accuracy (csrc/conv_b3.hip, round 6); "fp32" = v_mfma_f32_32x32x2_f32 (csrc/conv_train.hip and
#                                    the LDS-resident forms).  The model passes its own ``encoder_products`` with every call.


def set_encoder_products(kind):
    """Default product form of ``encoder_train`` calls that pass none ("bf16x3" / "fp32"); tools use it, the trainer does not."""
    global _encoder_products
    if kind not in ("bf16x3", "fp32"):
        raise ValueError(f"encoder_products must be 'bf16x3' or 'fp32', got {kind!r}")
    _encoder_products = kind


_B3_LAYERS = {(3, 84, 84, 32, 8, 4), (32, 20, 20, 64, 4, 2), (64, 9, 9, 64, 3, 1)}      # (C, H, W, Cout, K, S) of model.py:29-31 on 84 x 84


def conv_b3_pack(weights, dgrad, strides):
    """``weights[i]`` [Cout, C, K, K] -> the three bf16 planes of its forward (``dgrad[i]`` 0) or backward-data (1) operand in
    fragment order (etm_conv_b3_pack, one launch for all entries): int16 tensors of 3 * numel."""
    import ctypes
    lib = _lib.load()
    n = len(weights)
    outs = [torch.empty(3 * w.numel(), dtype=torch.int16, device=w.device) for w in weights]
    vp = lambda ts: (ctypes.c_void_p * n)(*[_ptr(t) for t in ts])
    ia = lambda vs: (ctypes.c_int32 * n)(*vs)
    _lib.check(lib.etm_conv_b3_pack(vp(weights), vp(outs), ia(dgrad), ia([w.shape[0] for w in weights]), ia([w.shape[1] for w in weights]),
                                    ia([w.shape[2] for w in weights]), ia(strides), n, _stream()), "etm_conv_b3_pack")
    return outs


class _EncoderFn(torch.autograd.Function):
    """The three relu(conv2d) layers of model.py:90-92 on NHWC activations: 3 forward launches; backward = 1 mask/layout kernel,
    3 weight-gradient kernels (+ their fixed-order reductions) and 2 data-gradient kernels, with bias, ReLU, ReLU masks and
    bias gradients fused in (csrc/conv_train.hip).  Returns the features NHWC-flattened [N, Ho*Wo*Cout]: the consumer permutes
    the columns of ITS weight (a 4.8 MB copy at config 3) instead of the features being transposed to upstream's (c, h, w)
    flatten order (model.py:94) and back."""

    @staticmethod
    def forward(ctx, x_nhwc, w1, b1, w2, b2, w3, b3, strides, index=None, products=None):
        lib = _lib.load()
        _need_dev(x_nhwc, w1, b1, w2, b2, w3, b3)
        x = _f32c(x_nhwc, "obs")
        st = _stream()
        acts, shapes = [x], []
        n, h, w, c = x.shape
        x_images = n
        if index is not None:          # batch image i = x[index[i]]: the minibatch gather rides in the first layer's loads
            n = index.numel()
        # both packings of the three layers' weights in ONE launch; the backward-data ones ride in the context
        import ctypes
        layers = ((w1, b1, strides[0]), (w2, b2, strides[1]), (w3, b3, strides[2]))
        wts = [_f32c(wt.detach(), "conv weight") for wt, _, _ in layers]
        geo, hh, ww = [], h, w
        for wt, _, s in layers:
            geo.append((wt.shape[1], hh, ww, wt.shape[0], wt.shape[2], s))
            hh, ww = (hh - wt.shape[2]) // s + 1, (ww - wt.shape[3]) // s + 1
        use_b3 = (products or _encoder_products) == "bf16x3" and all(g in _B3_LAYERS for g in geo) and all(wt.shape[2] == wt.shape[3] for wt in wts)
        if use_b3:      # the five operands (three forward, two backward-data) split and packed in ONE launch
            packs = conv_b3_pack(wts + wts[1:], [0, 0, 0, 1, 1], [l[2] for l in layers] + [l[2] for l in layers[1:]])
            dgrad_packs = [None] + packs[3:]
        else:
            packs = [torch.empty(wt.numel(), dtype=torch.float32, device=x.device) for wt in wts]
            dgrad_packs = [None] + [torch.empty(wt.numel(), dtype=torch.float32, device=x.device) for wt in wts[1:]]
            vp = lambda ts: (ctypes.c_void_p * 3)(*[_ptr(t) for t in ts])
            ia = lambda vs: (ctypes.c_int32 * 3)(*vs)
            _lib.check(lib.etm_conv_pack_weights_grouped(vp(wts), vp(packs), vp(dgrad_packs), ia([wt.shape[0] for wt in wts]),
                                                         ia([wt.shape[1] for wt in wts]), ia([wt.shape[2] for wt in wts]),
                                                         ia([wt.shape[3] for wt in wts]), ia([l[2] for l in layers]), 3, st),
                       "etm_conv_pack_weights_grouped")
        ctx.b3 = use_b3
        relu_bits = []      # (b3: the ReLU pattern of every layer's output, one bit per element -- what backward-data needs of it)
        for i, (wt, bs, s) in enumerate(layers):
            cout, _, kh, kw = wt.shape
            ho, wo = (h - kh) // s + 1, (w - kw) // s + 1
            y = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
            if use_b3:
                bits = torch.empty((n, ho, wo, cout // 32), dtype=torch.int32, device=x.device)
                relu_bits.append(bits)
                _lib.check(lib.etm_conv_b3_fwd(_ptr(acts[-1]), _ptr(index) if i == 0 else None, _ptr(packs[i]), _ptr(_f32c(bs.detach(), "bias")),
                                               _ptr(y), _ptr(bits), n, c, h, w, cout, kh, kw, s, st), "etm_conv_b3_fwd")
            else:
                _lib.check(lib.etm_conv_train_fwd(_ptr(acts[-1]), _ptr(index) if i == 0 else None, x_images, _ptr(packs[i]),
                                                  _ptr(_f32c(bs.detach(), "bias")), _ptr(y), n, c, h, w, cout, kh, kw, s, 0, st), "etm_conv_train_fwd")
            shapes.append((c, h, w, cout, kh, kw, s, ho, wo))
            acts.append(y)
            h, w, c = ho, wo, cout
        ctx.param_ptrs = tuple(t.data_ptr() for t in (w1, b1, w2, b2, w3, b3))
        ctx.shapes = shapes
        ctx.save_for_backward(acts[0], acts[1], acts[2], acts[3], dgrad_packs[1], dgrad_packs[2], index,
                              *(relu_bits if use_b3 else (None, None, None)))
        return acts[3].view(n, -1)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x0, y1, y2, y3, pd2, pd3, index, bits1, bits2, bits3 = ctx.saved_tensors
        st = _stream()
        dev = x0.device
        n = y1.shape[0]
        g = _f32c(g, "d_features")
        c3, h3, w3_, cout3, _, _, _, ho3, wo3 = ctx.shapes[2]
        if ctx.b3:      # the last layer's ReLU backward rides in the fills of its two consumers (pattern words of y3 from the forward pass)
            dy, dy_bits = g, bits3
        else:
            dy, dy_bits = torch.empty((n, ho3, wo3, cout3), dtype=torch.float32, device=dev), None
            _lib.check(lib.etm_relu_mask(_ptr(g), _ptr(y3), _ptr(dy), dy.numel(), st), "etm_relu_mask")
        grads = [None] * 6
        inputs = (x0, y1, y2)
        dgrad_packs = (None, pd2, pd3)
        # with a collector that knows the six parameters' arena views the three slice reductions become ONE launch at its flush
        col = DeferredDw.active
        dests = None
        if col is not None:
            dests = [col.dest.get(ptr) for ptr in ctx.param_ptrs]
            if any(v is None or not v.is_contiguous() for v in dests) or any(ptr in col.written for ptr in ctx.param_ptrs):
                dests = None
        deferred = []
        for i in (2, 1, 0):
            c, h, w, cout, kh, kw, s, ho, wo = ctx.shapes[i]
            K = kh * kw * c
            if ctx.b3:
                slices = lib.etm_conv_b3_wgrad_slices(n, c, h, w, cout, kh, kw, s)
                nbytes = slices * (K * cout + cout) * 4
            else:
                slices = lib.etm_conv_train_wgrad_slices(n, c, h, w, cout, kh, kw, s)
                nbytes = lib.etm_conv_train_wgrad_workspace_bytes(n, c, h, w, cout, kh, kw, s)
            if dests is not None:
                ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)      # (lives until the collector's flush)
                buf = None
                deferred.append((ws, slices, dests[2 * i], dests[2 * i + 1], cout, c, kh, kw))
            else:
                ws = workspace(nbytes, dev, "conv_wgrad")
                buf = torch.empty(K * cout + cout, dtype=torch.float32, device=dev)
            if ctx.b3:      # the slices on the bf16 matrix pipe (csrc/conv_b3_wgrad.hip); without a collector their reduction follows at once
                _lib.check(lib.etm_conv_b3_wgrad(_ptr(inputs[i]), _ptr(index) if i == 0 else None, _ptr(dy), _ptr(dy_bits), _ptr(ws), nbytes, n, c, h, w,
                                                 cout, kh, kw, s, st), "etm_conv_b3_wgrad")
                if buf is not None:
                    import ctypes
                    one = lambda ct, v: (ct * 1)(v)
                    _lib.check(lib.etm_conv_wgrad_reduce_grouped(one(ctypes.c_void_p, _ptr(ws)), one(ctypes.c_int32, slices), one(ctypes.c_void_p, _ptr(buf)),
                                                                 one(ctypes.c_void_p, buf.data_ptr() + K * cout * 4), one(ctypes.c_int32, cout),
                                                                 one(ctypes.c_int32, c), one(ctypes.c_int32, kh), one(ctypes.c_int32, kw), 1, st),
                               "etm_conv_wgrad_reduce_grouped")
            else:
                _lib.check(lib.etm_conv_train_wgrad(_ptr(inputs[i]), _ptr(index) if i == 0 else None, _ptr(dy), _ptr(buf), _ptr(ws), nbytes, n, c, h, w,
                                                    cout, kh, kw, s, st), "etm_conv_train_wgrad")
            if buf is not None:
                grads[2 * i] = buf[: K * cout].view(cout, c, kh, kw)
                grads[2 * i + 1] = buf[K * cout:]
            if i > 0:
                dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dev)
                if ctx.b3:
                    _lib.check(lib.etm_conv_b3_dgrad(_ptr(dy), _ptr(dy_bits), _ptr(dgrad_packs[i]), None, _ptr((None, bits1, bits2)[i]), _ptr(dx), n, c, h, w,
                                                     cout, kh, kw, s, st), "etm_conv_b3_dgrad")
                else:
                    _lib.check(lib.etm_conv_train_dgrad(_ptr(dy), _ptr(dgrad_packs[i]), _ptr(inputs[i]), _ptr(dx), n, c, h, w, cout, kh, kw, s, st),
                               "etm_conv_train_dgrad")
                dy, dy_bits = dx, None      # (the data gradient comes out masked: a pre-activation gradient)
        if deferred:
            col.conv_wgrads.extend(deferred)
            col.written.update(ctx.param_ptrs)
        return (None, *grads, None, None, None)


def encoder_train(obs_nhwc, conv1, conv2, conv3, index=None, products=None):
    """Differentiable encoder forward on an NHWC observation batch [N, H, W, C] -- or, with ``index`` (int64 [n]), on the images
    ``obs_nhwc[index]`` without gathering them first: features [N, Ho * Wo * Cout], NHWC-flattened
    (``linear_relu_nhwc`` is the following linear layer on that column order).  Gradients flow to the
    convolution weights and biases (observations need none)."""
    if products not in (None, "bf16x3", "fp32"):
        raise ValueError(f"products must be 'bf16x3' or 'fp32', got {products!r}")
    return _EncoderFn.apply(obs_nhwc, conv1.weight, conv1.bias, conv2.weight, conv2.bias, conv3.weight, conv3.bias,
                            (conv1.stride[0], conv2.stride[0], conv3.stride[0]), index, products)


class ReplayAfterWarmup:
    """``fn()`` -- device work on fixed-address operands without host synchronisation (a weight repacking, a cache refresh: dozens of
    small launches once per update) -- run eagerly for its first ``warm`` calls and from a captured graph afterwards (round 6:
    refresh_rollout_weights 2.3 -> 0.3 ms per update at config 5).  ``after``: host-side bookkeeping that must follow every execution,
    captured or not.  A failed capture keeps the eager form for good; ``enabled`` False never captures."""

    def __init__(self, fn, device, warm=2, after=None, enabled=True, what="refresh"):
        self.fn, self.device, self.warm, self.after, self.enabled, self.what = fn, device, warm, after, enabled, what
        self.calls, self.graph, self.failed = 0, None, False

    def __call__(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.calls += 1
            capture = (self.enabled and not self.failed and self.calls > self.warm and torch.device(self.device).type == "cuda"
                       and not torch.cuda.is_current_stream_capturing())
            if capture:
                try:
                    torch.cuda.synchronize(self.device)
                    g = torch.cuda.CUDAGraph()
                    with torch.no_grad(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self.fn()
                    self.graph = g
                    g.replay()
                except Exception as exc:       # noqa: BLE001
                    import sys
                    self.failed, self.graph = True, None
                    torch.cuda.synchronize(self.device)
                    print(f"[etm] {self.what} stays eager (capture failed: {exc!r})", file=sys.stderr, flush=True)
                    self.fn()
            else:
                self.fn()
        if self.after is not None:
            self.after()


def host_view(t):
    """numpy array over the memory of the contiguous float32 device tensor ``t`` at its HOST address (large-BAR systems map the
    device's memory into the process: the pointer is the same) -- WRITE-ONLY use: host reads through the BAR are uncached and
    slow, and the device's caches know nothing of them.  Call ``host_direct_write_ok`` first."""
    import ctypes
    import numpy as np
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise TypeError("host_view: a contiguous float32 device tensor is needed")
    buf = (ctypes.c_float * t.numel()).from_address(t.data_ptr())
    return np.ctypeslib.as_array(buf).reshape(tuple(t.shape))


_direct_ok = {}
_direct_mode = {}      # device index -> etm_host_direct_write_init's answer (2: with an HDP flush register, 1: without, 0: not usable)


def host_direct_write_ok(device):
    """Can the host write rows straight into this device's memory (etm_host_direct_write_init: large BAR + HDP flush register),
    and does a written pattern come back through the device (self-test, once per device)?"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _direct_ok:
        ok = False
        try:
            lib = _lib.load()
            mode = lib.etm_host_direct_write_init(idx)
            _direct_mode[idx] = mode
            if mode >= 1:
                import numpy as np
                # 48 rounds of "host rewrites a small buffer whose lines the previous kernel has just read, a kernel reads it
                # again": what a rollout does with its staging rows (tools/microbench/bar_stale.hip is the long form: 12,000
                # plain launches and graph replays, none stale)
                probe = torch.zeros(1024, dtype=torch.float32, device=device)
                view = host_view(probe)
                base = np.arange(1024, dtype=np.float32) * 0.5
                seen = []
                for k in range(48):
                    torch.cuda.synchronize(device)
                    view[:] = base + float(k + 1)
                    lib.etm_host_store_fence(idx)
                    seen.append(probe + 0.0)
                got = torch.stack(seen).cpu().numpy()
                ok = bool(all(np.array_equal(got[k], base + float(k + 1)) for k in range(48)))
        except Exception:      # noqa: BLE001 -- anything odd: keep pinned memory + uploads
            ok = False
        _direct_ok[idx] = ok
    return _direct_ok[idx]


def upload(dst, src_pinned, stream):
    """Asynchronous pinned-host -> device copy of a contiguous block on ``stream`` (a torch.cuda.Stream)."""
    lib = _lib.load()
    nbytes = src_pinned.numel() * src_pinned.element_size()
    if not (dst.is_contiguous() and src_pinned.is_contiguous()) or dst.numel() * dst.element_size() != nbytes:
        raise TypeError("upload needs contiguous blocks of equal size")
    _lib.check(lib.etm_upload(dst.data_ptr(), src_pinned.data_ptr(), nbytes, stream.cuda_stream), "etm_upload")


_fused_linear_relu = None  # None: untested, True/False after the first call


class _LinearReluFn(torch.autograd.Function):
    """y = relu(x W^T + b) with the element-wise part of the backward in two launches (etm_relu_bwd_colsum: ReLU mask and
    bias gradient in one pass + the fixed-order column-sum reduction) instead of a mask multiply and a framework reduction."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        if _fused_linear_relu:        # bias + ReLU in the GEMM's epilogue (one launch; verified against the two-op form by linear_relu)
            y = torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
        else:
            y = torch.addmm(bias, x, weight.t())
            torch.relu_(y)
        ctx.save_for_backward(x, weight, y)
        ctx.bias_ptr = bias.data_ptr()
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        lib = _lib.load()
        g = _f32c(g, "grad")
        n, c = g.shape
        gm = torch.empty_like(g)
        nbytes = lib.etm_relu_bwd_colsum_workspace_bytes(n, c)
        col = DeferredDw.active
        db = None
        part = torch.empty(nbytes // 4, dtype=torch.float32, device=g.device) if col is not None else None
        if col is not None and col.offer_colsum(part, lib.etm_relu_bwd_colsum_partial_rows(n), c, [(0, c, ctx.bias_ptr)]):
            # the bias gradient's second stage rides in the collector's one reduction launch
            _lib.check(lib.etm_relu_bwd_colsum(_ptr(g), _ptr(y), _ptr(gm), None, _ptr(part), nbytes, n, c, _stream()), "etm_relu_bwd_colsum")
        else:
            db = torch.empty(c, dtype=torch.float32, device=g.device)
            ws = workspace(nbytes, g.device, "relu_bwd")
            _lib.check(lib.etm_relu_bwd_colsum(_ptr(g), _ptr(y), _ptr(gm), _ptr(db), _ptr(ws), nbytes, n, c, _stream()), "etm_relu_bwd_colsum")
        dx = gm.mm(weight) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1] and not _offer_dw(gm, x, weight):
            dw = gm.t().mm(x)
        return dx, dw, db


class _LinearBiasFn(torch.autograd.Function):
    """y = x W^T + b (transformer.py:29 fc_out on the paths that keep its bias: pre-LN / gated blocks).  nn.Linear's backward leaves
    the bias gradient to the framework's column sum -- two stages with a semaphore, which this runtime does not replay reliably from a
    captured graph (profiles/r05/graph_reduce_hazard.txt) and which costs two launches; here it is etm_relu_bwd_colsum without a mask,
    its second stage in the collector's one reduction launch; the weight gradient goes to the grouped launch."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.bias_ptr = bias.data_ptr()
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        lib = _lib.load()
        g = _f32c(g, "grad")
        n, c = g.shape
        db = None
        if ctx.needs_input_grad[2]:
            nbytes = lib.etm_relu_bwd_colsum_workspace_bytes(n, c)
            col = DeferredDw.active
            part = torch.empty(nbytes // 4, dtype=torch.float32, device=g.device) if col is not None else None
            if col is not None and col.offer_colsum(part, lib.etm_relu_bwd_colsum_partial_rows(n), c, [(0, c, ctx.bias_ptr)]):
                _lib.check(lib.etm_relu_bwd_colsum(_ptr(g), None, None, None, _ptr(part), nbytes, n, c, _stream()), "etm_relu_bwd_colsum")
            else:
                db = torch.empty(c, dtype=torch.float32, device=g.device)
                ws = workspace(nbytes, g.device, "relu_bwd")
                _lib.check(lib.etm_relu_bwd_colsum(_ptr(g), None, None, _ptr(db), _ptr(ws), nbytes, n, c, _stream()), "etm_relu_bwd_colsum")
        dx = g.mm(weight) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1] and not _offer_dw(g, x, weight):
            dw = g.t().mm(x)
        return dx, dw, db


def linear_bias(lin, x):
    """``lin(x)`` (an nn.Linear with bias) for 2-D fp32 device tensors under autograd: library GEMMs for y and dx, the bias gradient by
    the library's column sums, both parameter gradients deferrable (see DeferredDw).  Anything else: ``lin(x)``."""
    if (torch.is_grad_enabled() and lin.bias is not None and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous()):
        return _LinearBiasFn.apply(x, lin.weight, lin.bias)
    return lin(x)


class _LinearReluNhwcFn(torch.autograd.Function):
    """relu(x Wp^T + b) for NHWC-flattened features x [N, HW * C] and a weight kept in upstream's (c, h, w) column order
    (model.py:94): Wp = the weight with its columns in (h, w, c) order.  The permutation lives INSIDE the node: one permuting copy
    of the weight forward, and backward the weight gradient is permuted straight into the parameter's arena view when a DeferredDw
    collector knows it (one permuting copy instead of a permute-back copy plus the gradient packing copy)."""

    @staticmethod
    def forward(ctx, x, weight, bias, channels):
        out, feat = weight.shape
        hw = feat // channels
        wp = weight.detach().view(out, channels, hw).transpose(1, 2).reshape(out, feat)       # [out, (h, w, c)]
        if _fused_linear_relu:
            y = torch._addmm_activation(bias, x, wp.t(), use_gelu=False)
        else:
            y = torch.addmm(bias, x, wp.t())
            torch.relu_(y)
        ctx.save_for_backward(x, wp, y)
        ctx.bias_ptr, ctx.weight_ptr, ctx.channels = bias.data_ptr(), weight.data_ptr(), channels
        return y

    @staticmethod
    def backward(ctx, g):
        x, wp, y = ctx.saved_tensors
        lib = _lib.load()
        g = _f32c(g, "grad")
        n, c = g.shape
        out, feat = wp.shape
        hw = feat // ctx.channels
        gm = torch.empty_like(g)
        nbytes = lib.etm_relu_bwd_colsum_workspace_bytes(n, c)
        col = DeferredDw.active
        db = None
        part = torch.empty(nbytes // 4, dtype=torch.float32, device=g.device) if col is not None else None
        if col is not None and col.offer_colsum(part, lib.etm_relu_bwd_colsum_partial_rows(n), c, [(0, c, ctx.bias_ptr)]):
            _lib.check(lib.etm_relu_bwd_colsum(_ptr(g), _ptr(y), _ptr(gm), None, _ptr(part), nbytes, n, c, _stream()), "etm_relu_bwd_colsum")
        else:
            db = torch.empty(c, dtype=torch.float32, device=g.device)
            ws = workspace(nbytes, g.device, "relu_bwd")
            _lib.check(lib.etm_relu_bwd_colsum(_ptr(g), _ptr(y), _ptr(gm), _ptr(db), _ptr(ws), nbytes, n, c, _stream()), "etm_relu_bwd_colsum")
        dx = gm.mm(wp) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            dwp = gm.t().mm(x).view(out, hw, ctx.channels).transpose(1, 2)                # [out, c, hw] view of the (h, w, c) result
            dest = col.dest.get(ctx.weight_ptr) if col is not None else None
            if dest is not None and dest.is_contiguous() and ctx.weight_ptr not in col.written:
                dest.view(out, ctx.channels, hw).copy_(dwp)
                col.written.add(ctx.weight_ptr)
            else:
                dw = dwp.reshape(out, feat)
        return dx, dw, db, None


def linear_relu_nhwc(x, weight, bias, channels):
    """relu(F.linear(x, nhwc_columns(weight, channels), bias)) as one autograd node (see _LinearReluNhwcFn)."""
    return _LinearReluNhwcFn.apply(x, weight, bias, channels)


def linear_relu_train(x, weight, bias):
    """relu(F.linear(x, weight, bias)) for 2-D fp32 device tensors under autograd (see _LinearReluFn)."""
    return _LinearReluFn.apply(x, weight, bias)


def linear_relu(lin, x, out=None):
    """relu(lin(x)).  In the no-grad rollout path the ReLU rides in the GEMM epilogue (hipBLASLt through
    ``torch._addmm_activation``), saving one launch per layer; with autograd enabled it is the plain two-op form.
    ``out`` (no-grad only): contiguous destination."""
    global _fused_linear_relu

    def plain():
        if (torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and lin.bias is not None and out is None
                and x.is_contiguous()):
            return _LinearReluFn.apply(x, lin.weight, lin.bias)
        y = torch.relu(torch.nn.functional.linear(x, lin.weight, lin.bias))
        return y if out is None else out.copy_(y)

    if torch.is_grad_enabled() or _fused_linear_relu is False or x.dim() != 2:
        return plain()
    if _fused_linear_relu is None:
        if torch.cuda.is_current_stream_capturing():
            return plain()
        try:
            y = torch._addmm_activation(lin.bias, x, lin.weight.t(), use_gelu=False)
            ref = torch.relu(torch.nn.functional.linear(x, lin.weight, lin.bias))
            _fused_linear_relu = bool(torch.allclose(y, ref, atol=1e-5, rtol=1e-5))
        except Exception:
            _fused_linear_relu = False
        return plain()
    if out is not None:
        return torch._addmm_activation(lin.bias, x, lin.weight.t(), use_gelu=False, out=out)
    return torch._addmm_activation(lin.bias, x, lin.weight.t(), use_gelu=False)


def gae(rewards, dones, values, last_value, gamma, lamda, out=None):
    """Generalized advantage estimation on device; [W,S] row-major like the reference buffer."""
    lib = _lib.load()
    _need_dev(rewards, dones, values, last_value)
    rewards, values, last_value = _f32c(rewards, "rewards"), _f32c(values, "values"), _f32c(last_value, "last_value")
    d = dones.contiguous()
    d = d.view(torch.uint8) if d.dtype == torch.bool else d.to(torch.uint8)
    W, S = values.shape
    if out is None:
        out = torch.empty_like(values)
    elif not out.is_contiguous() or out.dtype != torch.float32:
        raise TypeError("advantages output must be contiguous float32")
    g32 = float(torch.tensor(float(gamma), dtype=torch.float32))
    gl32 = float(torch.tensor(float(gamma) * float(lamda), dtype=torch.float32))  # python-float product, then fp32 (buffer.py:111)
    rc = lib.etm_gae(_ptr(rewards), _ptr(d), _ptr(values), _ptr(last_value), g32, gl32, _ptr(out), W, S, _stream())
    _lib.check(rc, "etm_gae")
    return out


def adv_stats(adv):
    """(count, mean, M2) of a flat advantage vector as a 3-element device tensor."""
    lib = _lib.load()
    _need_dev(adv)
    adv = _f32c(adv.reshape(-1), "adv")
    stats = torch.empty(3, dtype=torch.float32, device=adv.device)
    ws_bytes = lib.etm_adv_stats_workspace_bytes(adv.numel())
    if ws_bytes > 0:       # large N: per-chunk statistics on many workgroups + a fixed-order merge
        ws = workspace(ws_bytes, adv.device, tag="adv_stats")
        _lib.check(lib.etm_adv_stats_ws(_ptr(adv), adv.numel(), _ptr(stats), _ptr(ws), ws_bytes, _stream()), "etm_adv_stats_ws")
        return stats
    _lib.check(lib.etm_adv_stats(_ptr(adv), adv.numel(), _ptr(stats), _stream()), "etm_adv_stats")
    return stats


class _PpoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, value, actions, old_logp, adv, old_value, stats3, branch, n_branches, clip, vf_coef, beta,
                include_value, dyn=None):
        lib = _lib.load()
        _need_dev(logits, value, actions, old_logp, adv, old_value, stats3)
        logits = _f32c(logits, "logits")
        value = _f32c(value, "value")
        adv, old_value = _f32c(adv, "adv"), _f32c(old_value, "old_value")
        if actions.dtype != torch.int64 or old_logp.dtype != torch.float32:
            raise TypeError("actions must be int64 and old log-probs float32")
        actions, old_logp = actions.contiguous(), old_logp.contiguous()
        N, A = logits.shape
        B = actions.shape[1] if actions.dim() == 2 else 1
        dev = logits.device
        out8 = torch.empty(8, dtype=torch.float32, device=dev)
        d_logits = torch.empty_like(logits)
        d_value = torch.empty_like(value)
        nbytes = lib.etm_ppo_loss_workspace_bytes(N)
        ws = workspace(nbytes, dev, "ppo")
        rc = lib.etm_ppo_loss(_ptr(logits), actions.data_ptr() + 8 * branch, B, old_logp.data_ptr() + 4 * branch, B, _ptr(adv),
                              _ptr(old_value), _ptr(value), _ptr(stats3), float(clip), float(vf_coef), float(beta),
                              1.0 / (N * n_branches), 1.0 / N, 1.0 / N, 1 if include_value else 0, _ptr(out8), _ptr(d_logits),
                              _ptr(d_value), _ptr(ws), nbytes, _ptr(dyn), N, A, _stream())
        _lib.check(rc, "etm_ppo_loss")
        ctx.save_for_backward(d_logits, d_value)
        ctx.include_value = include_value
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(out8)
        return out8[2].clone(), out8

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        if g_loss is None:
            return (None,) * 14
        d_logits, d_value = ctx.saved_tensors
        gv = d_value * g_loss if ctx.include_value else None
        return d_logits * g_loss, gv, None, None, None, None, None, None, None, None, None, None, None, None


class _HeadsLossFn(torch.autograd.Function):
    """Hidden heads (bias + ReLU), policy branch, value head, PPO loss and the backward pass down to the hidden heads' pre-activations
    as one kernel pass (csrc/heads_loss.hip); the two hidden GEMMs h W^T (forward) and gm W (backward) stay library GEMMs, the hidden
    heads' weight gradients go to the grouped launch.  Gradients are those of ``loss`` itself (unit upstream gradient) times
    ``g_loss``."""

    @staticmethod
    def forward(ctx, h, wlp, blp, wlv, blv, wb, bb, wv, bv, actions, old_logp, adv, old_value, stats3, clip, vf_coef, beta, dyn, unit_grad):
        lib = _lib.load()
        _need_dev(h, wlp, blp, wlv, blv, wb, bb, wv, bv, actions, old_logp, adv, old_value, stats3)
        h = _f32c(h, "h")
        N, hid, A = h.shape[0], wlp.shape[0], wb.shape[0]
        dev = h.device
        pre_p, pre_v = h.mm(wlp.t()), h.mm(wlv.t())
        gm_p, gm_v = torch.empty_like(pre_p), torch.empty_like(pre_v)
        row = lib.etm_heads_loss_row_floats(hid, A)
        sums = torch.empty(row, dtype=torch.float32, device=dev)
        out8 = torch.empty(8, dtype=torch.float32, device=dev)
        nbytes = lib.etm_heads_loss_workspace_bytes(N, hid, A)
        ws = workspace(nbytes, dev, "heads_loss")
        actions, old_logp = actions.contiguous(), old_logp.contiguous()
        B = actions.shape[1] if actions.dim() == 2 else 1
        rc = lib.etm_heads_loss(_ptr(pre_p), _ptr(pre_v), _ptr(blp), _ptr(blv), _ptr(wb), _ptr(bb), _ptr(wv), _ptr(bv), _ptr(actions), B,
                                _ptr(old_logp), B, _ptr(_f32c(adv, "adv")), _ptr(_f32c(old_value, "old_value")), _ptr(stats3), float(clip),
                                float(vf_coef), float(beta), 1.0 / N, 1.0 / N, 1.0 / N, _ptr(dyn), _ptr(gm_p), _ptr(gm_v), _ptr(sums), _ptr(out8),
                                0, 0, _ptr(ws), nbytes, N, hid, A, _stream())
        _lib.check(rc, "etm_heads_loss")
        ctx.save_for_backward(h, wlp, wlv, gm_p, gm_v, sums)
        ctx.dims = (hid, A)
        ctx.unit = bool(unit_grad)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(out8)
        return out8[2].clone(), out8

    @staticmethod
    def backward(ctx, g_loss, _g_stats):
        if g_loss is None:
            return (None,) * 19
        h, wlp, wlv, gm_p, gm_v, sums = ctx.saved_tensors
        hid, A = ctx.dims
        unit = ctx.unit                 # the caller promises loss.backward() on the returned loss itself: g_loss == 1, nothing to scale
        dh = None
        if ctx.needs_input_grad[0]:
            dh = gm_p.mm(wlp)
            dh.addmm_(gm_v, wlv)
            if not unit:
                dh = dh * g_loss
        dwlp = dwlv = None
        if ctx.needs_input_grad[1] and not (unit and _offer_dw(gm_p, h, wlp)):
            dwlp = gm_p.t().mm(h) if unit else gm_p.t().mm(h) * g_loss
        if ctx.needs_input_grad[3] and not (unit and _offer_dw(gm_v, h, wlv)):
            dwlv = gm_v.t().mm(h) if unit else gm_v.t().mm(h) * g_loss
        s = sums if unit else sums * g_loss
        o = (3 + A) * hid
        return (dh, dwlp, s[:hid], dwlv, s[hid:2 * hid], s[3 * hid:o].view(A, hid), s[o:o + A], s[2 * hid:3 * hid].view(1, hid),
                s[o + A:o + A + 1], None, None, None, None, None, None, None, None, None, None)


def heads_loss_supported(h, lin_policy, branch):
    return (h.is_cuda and h.dim() == 2 and h.dtype == torch.float32
            and bool(_lib.load().etm_heads_loss_supported(h.shape[0], lin_policy.weight.shape[0], branch.weight.shape[0])))


def heads_ppo_loss(h, lin_policy, lin_value, branch, value_head, actions, old_logp, adv, old_value, clip, vf_coef, beta, stats3=None, dyn=None,
                   unit_grad=False):
    """model.py:101-110 + the PPO loss of ``ppo_loss`` for a single-branch policy, from the transformer output ``h`` [N, D]:
    -> (loss scalar with grad, stats[6]).  ``unit_grad=True``: the caller runs ``backward()`` on the returned loss itself (upstream
    gradient 1): no scaling launches, weight gradients of the hidden heads deferrable.  See _HeadsLossFn."""
    if stats3 is None:
        stats3 = adv_stats(adv)
    if dyn is not None and (dyn.dtype != torch.float64 or dyn.numel() != 2 or not dyn.is_cuda):
        raise TypeError("heads_ppo_loss: dyn must be a float64 device tensor (clip, beta)")
    loss, st = _HeadsLossFn.apply(h, lin_policy.weight, lin_policy.bias, lin_value.weight, lin_value.bias, branch.weight, branch.bias,
                                  value_head.weight, value_head.bias, actions, old_logp, adv, old_value, stats3, clip, vf_coef, beta, dyn,
                                  unit_grad)
    return loss, st[:6]


def ppo_loss(logits_list, value, actions, old_logp, adv, old_value, clip, vf_coef, beta, stats3=None, dyn=None):
    """PPO loss over all action branches.  Returns (loss scalar with grad, stats[6] device tensor in trainer.py:318-323 order).
    ``dyn``: optional float64 device tensor (clip, beta) read by the kernels at run time instead of the two scalars."""
    if stats3 is None:
        stats3 = adv_stats(adv)
    if dyn is not None and (dyn.dtype != torch.float64 or dyn.numel() != 2 or not dyn.is_cuda):
        raise TypeError("ppo_loss: dyn must be a float64 device tensor (clip, beta)")
    nb = len(logits_list)
    loss, st = _PpoLossFn.apply(logits_list[0], value, actions, old_logp, adv, old_value, stats3, 0, nb, clip, vf_coef, beta, True, dyn)
    if nb == 1:
        return loss, st[:6]
    pol, ent, kl, cf = st[0], st[3], st[4], st[5]
    total = loss
    for b in range(1, nb):
        l_b, s_b = _PpoLossFn.apply(logits_list[b], value, actions, old_logp, adv, old_value, stats3, b, nb, clip, vf_coef, beta, False, dyn)
        total = total + l_b
        pol, ent, kl, cf = pol + s_b[0], ent + s_b[3], kl + s_b[4], cf + s_b[5]
    return total, torch.stack([pol, st[1], total.detach(), ent, kl, cf])
