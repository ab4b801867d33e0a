"""TransformerXL-style episodic memory encoder on the MI355X kernels.

Same public classes, constructor arguments, forward signatures and ``state_dict`` keys as the upstream
``transformer.py`` (MultiHeadAttention :8-86, TransformerBlock :88-172, SinusoidalPosition :174-186,
Transformer :188-253, GRUGate :255-298), but the window attention of every block -- gather of the memory window,
positional rows, ``norm_kv``, K/V projection, masked softmax, attention-weighted sum -- is ONE fused HIP kernel
(``etm.ops.mha``) that reads the episodic memory bank in place.  The small [N, D] x [D, D] maps around it (query
projection, ``fc_out``, feed-forward, GRU gates) are library GEMMs.

Besides the upstream calling convention (pre-gathered ``memories`` [N, L, blocks, D]) every level also accepts a
``WindowSpec`` that addresses the windows inside the whole-episode bank, which is what the trainer uses so the
[N, L, blocks, D] tensor is never materialised.
"""
import math

import torch
from torch import nn

from etm import ops
from etm.ops import WindowSpec
from utils import Module


class MultiHeadAttention(nn.Module):
    """Single-query multi-head attention over a memory window (no dropout)."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        if embed_dim % num_heads != 0:
            raise AssertionError("Embedding dimension needs to be divisible by the number of heads")
        self.embed_dim, self.num_heads, self.head_size = embed_dim, num_heads, embed_dim // num_heads
        self.values = nn.Linear(embed_dim, embed_dim, bias=False)
        self.keys = nn.Linear(embed_dim, embed_dim, bias=False)
        self.queries = nn.Linear(embed_dim, embed_dim, bias=False)
        self.fc_out = nn.Linear(embed_dim, embed_dim)

    def attend(self, query, spec: WindowSpec, block=0, pos=None, norm_kv=None, raw=False):
        """query [N, D] -> (output [N, D], attention [N, H, L]); window rows come from ``spec``.  ``raw``: the output is
        ``ctx fc_out.weight^T`` WITHOUT ``fc_out.bias`` (the caller folds the bias into the kernel that follows)."""
        q = ops.linear_nobias(query, self.queries.weight)
        ln_g = ln_b = None
        eps = 1e-5
        if norm_kv is not None:
            ln_g, ln_b, eps = norm_kv.weight, norm_kv.bias, norm_kv.eps
        ctx, att = ops.mha(q, self.keys.weight, self.values.weight, spec, block, self.num_heads, ln_g, ln_b, pos, eps)
        if raw:
            return ops.linear_nobias(ctx, self.fc_out.weight), att
        return ops.linear_bias(self.fc_out, ctx), att

    def forward(self, values, keys, queries, mask):
        """Upstream signature: values/keys [N, L, D], queries [N, 1, D], mask [N, L] -> ([N, 1, D], [N, H, 1, L])."""
        if keys.data_ptr() != values.data_ptr() or keys.shape != values.shape or keys.stride() != values.stride():
            raise NotImplementedError("the fused kernel projects keys and values from the same memory window "
                                      "(the only way the upstream model calls it, transformer.py:249)")
        if queries.shape[1] != 1:
            raise NotImplementedError("episodic attention is single-query (query_len == 1)")
        if mask is None:
            mask = torch.ones(values.shape[:2], dtype=torch.bool, device=values.device)
        spec = WindowSpec.from_windows(values, None, mask)
        out, att = self.attend(queries[:, 0], spec)
        return out.unsqueeze(1), att.unsqueeze(2)


class GRUGate(nn.Module):
    """GTrXL gating unit: r, z gates and candidate from six bias-free maps (+ gate bias ``bg``)."""

    def __init__(self, input_dim: int, bg: float = 0.0):
        super().__init__()
        for name in ("Wr", "Ur", "Wz", "Uz", "Wg", "Ug"):
            lin = nn.Linear(input_dim, input_dim, bias=False)
            setattr(self, name, lin)
        self.bg = nn.Parameter(torch.full([input_dim], float(bg)))
        for name in ("Wr", "Ur", "Wz", "Uz", "Wg", "Ug"):
            nn.init.xavier_uniform_(getattr(self, name).weight)

    def refresh_rollout_weights(self):
        """Concatenated gate maps for the fused no-grad path; buffers keep their address (captured rollout graph)."""
        with torch.no_grad():
            wy = torch.cat((self.Wr.weight, self.Wz.weight, self.Wg.weight), dim=0)
            ux = torch.cat((self.Ur.weight, self.Uz.weight), dim=0)
            if getattr(self, "_wy", None) is None or self._wy.device != wy.device:
                self._wy, self._ux = wy.contiguous(), ux.contiguous()
            else:
                self._wy.copy_(wy)
                self._ux.copy_(ux)
            self._wver = self._versions()

    def _versions(self):
        return tuple(m.weight._version for m in (self.Wr, self.Wz, self.Wg, self.Ur, self.Uz))

    def forward(self, x, y, fork=False):
        """``fork`` (fused training path only): the result twice -- see ops.gru_gate_train."""
        if not torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and getattr(self, "_wy", None) is not None:
            if not torch.cuda.is_current_stream_capturing() and self._wver != self._versions():
                self.refresh_rollout_weights()       # weights moved since the copies were made (optimizer step, load)
            return ops.gru_gate(x, y, self._wy, self._ux, self.Ug.weight, self.bg)
        if torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and x.shape[1] <= 1024:
            # training: concatenated GEMMs + fused forward / backward kernels; the concatenations come packed (one launch for all
            # gates of the model, Transformer._pack_gate_weights) when the enclosing forward pass made them, else from two cat launches
            packed, self._train_cat = getattr(self, "_train_cat", None), None
            return ops.gru_gate_train(self, x, y, packed, fork)
        r = torch.sigmoid(self.Wr(y) + self.Ur(x))
        z = torch.sigmoid(self.Wz(y) + self.Uz(x) - self.bg)
        cand = torch.tanh(self.Wg(y) + self.Ug(r * x))
        out = (1 - z) * x + z * cand
        return (out, out) if fork else out


class TransformerBlock(Module):
    def __init__(self, embed_dim, num_heads, config):
        super().__init__()
        self.attention = MultiHeadAttention(embed_dim, num_heads)
        self.use_gtrxl = bool(config["gtrxl"]) if "gtrxl" in config else False
        if self.use_gtrxl:
            self.gate1 = GRUGate(embed_dim, config["gtrxl_bias"])
            self.gate2 = GRUGate(embed_dim, config["gtrxl_bias"])
        self.layer_norm = config["layer_norm"]
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)
        if self.layer_norm == "pre":
            self.norm_kv = nn.LayerNorm(embed_dim)
        self.fc = nn.Sequential(nn.Linear(embed_dim, embed_dim), nn.ReLU())

    def forward_window(self, h, spec: WindowSpec, block=0, pos=None, h_res=None, fork=False):
        """h [N, D] query state; returns (new state [N, D], attention [N, H, L]).  ``h_res``: the second copy of ``h`` for the
        residual branch when the producer forked it (ops.fused_layernorm / ops.gru_gate_train); ``fork``: return the new state forked
        for the next block, (state, attention, state for the residual) -- both only on the fused training paths (post-LN without
        gates, pre-LN with gates; the other layouts return None for the third element)."""
        pre = self.layer_norm == "pre"
        fused = self._fused_train(h)
        q_in = (ops.fused_layernorm(h, self.norm1) if fused else self.norm1(h)) if pre else h
        if fused and self.layer_norm == "post" and not self.use_gtrxl:
            # training, post-LN: fc_out and fc run as plain GEMMs; bias, ReLU, residual add and LayerNorm (and all of their
            # gradients) are one forward and one backward kernel each (transformer.py:143-149, :160-170)
            # norm1's output feeds fc AND the residual of norm2 (and the block's output the next block's query projection and
            # residual): forked, so that the two gradients are added by the LayerNorm backward kernel on load
            att_raw, att_w = self.attention.attend(q_in, spec, block, pos, None, raw=True)
            x, x_res = ops.fused_layernorm(att_raw, self.norm1, bias=self.attention.fc_out.bias, res=h if h_res is None else h_res, fork=True)
            fc = self.fc[0]
            f_raw = ops.linear_nobias(x, fc.weight)
            if fork:
                out, out_res = ops.fused_layernorm(f_raw, self.norm2, bias=fc.bias, res=x_res, relu=True, fork=True)
                return out, att_w, out_res
            return ops.fused_layernorm(f_raw, self.norm2, bias=fc.bias, res=x_res, relu=True), att_w
        att_out, att_w = self.attention.attend(q_in, spec, block, pos, self.norm_kv if pre else None)
        if fused and pre and self.use_gtrxl:
            # training, pre-LN + gates: a gate's output feeds the next LayerNorm AND the next gate's residual input (transformer.py:
            # 143-149, :160-170 with :287-298) -- forked, the two gradients are added by the gate's backward kernel on load
            x, x_res = self.gate1(h if h_res is None else h_res, att_out, fork=True)
            f = ops.linear_relu(self.fc[0], ops.fused_layernorm(x, self.norm2))
            if fork:
                out, out_res = self.gate2(x_res, f, fork=True)
                return out, att_w, out_res
            return self.gate2(x_res, f), att_w
        out = self._after_attention(h if h_res is None else h_res, att_out)
        return (out, att_w, None) if fork else (out, att_w)

    @staticmethod
    def _fused_train(h):
        return torch.is_grad_enabled() and h.is_cuda and h.dim() == 2 and h.shape[1] <= 1024

    def forward_cached(self, h, kv_spec: WindowSpec, block=0, out=None):
        """Rollout path: attention over cached K/V projections (no grad).  ``out``: contiguous destination of the result."""
        q_in = self.norm1(h) if self.layer_norm == "pre" else h
        ctx, _ = ops.attn_cached(self.attention.queries(q_in), kv_spec, block, self.attention.num_heads)
        if self.layer_norm == "post" and not self.use_gtrxl and not torch.is_grad_enabled():
            # rollout, post-LN without gates: the two linear layers run WITHOUT their bias / ReLU epilogue (plain library GEMMs,
            # tuned per shape) and the epilogues ride in the residual + LayerNorm kernels that follow them
            fc_out, fc = self.attention.fc_out, self.fc[0]
            x = ops.add_layernorm(torch.nn.functional.linear(ctx, fc_out.weight), h, self.norm1, bias=fc_out.bias)
            return ops.add_layernorm(torch.nn.functional.linear(x, fc.weight), x, self.norm2, out=out, bias=fc.bias, relu=True)
        return self._after_attention(h, self.attention.fc_out(ctx), out)

    def _after_attention(self, h, att_out, out=None):
        pre, post = self.layer_norm == "pre", self.layer_norm == "post"
        if post and not self.use_gtrxl and not torch.is_grad_enabled():
            # no-grad path with materialised att_out: residual + LayerNorm and Linear + ReLU are one launch each
            x = ops.add_layernorm(att_out, h, self.norm1)
            return ops.add_layernorm(ops.linear_relu(self.fc[0], x), x, self.norm2, out=out)
        fused = self._fused_train(h)
        ln = (lambda t, norm: ops.fused_layernorm(t, norm)) if fused else (lambda t, norm: norm(t))
        x = self.gate1(h, att_out) if self.use_gtrxl else att_out + h
        if post:
            x = ln(x, self.norm1)
        f = ops.linear_relu(self.fc[0], ln(x, self.norm2) if pre else x)
        res = self.gate2(x, f) if self.use_gtrxl else f + x
        if post:
            res = ln(res, self.norm2)
        return res if out is None else out.copy_(res)

    def forward(self, value, key, query, mask):
        """Upstream signature: value/key [N, L, D] (same tensor), query [N, 1, D], mask [N, L]."""
        if key.data_ptr() != value.data_ptr():
            raise NotImplementedError("keys and values must be the same memory window")
        spec = WindowSpec.from_windows(value, None, mask)
        out, att = self.forward_window(query[:, 0], spec)
        return out.unsqueeze(1), att.unsqueeze(2)


class SinusoidalPosition(nn.Module):
    """Reversed absolute sinusoid ("relative" option): row i of the table encodes position seq_len - 1 - i."""

    def __init__(self, dim, min_timescale=2.0, max_timescale=1e4):
        super().__init__()
        freqs = torch.arange(0, dim, min_timescale)
        self.register_buffer("inv_freqs", max_timescale ** (-freqs / dim))

    def forward(self, seq_len):
        seq = torch.arange(seq_len - 1, -1, -1.0, device=self.inv_freqs.device)
        ang = seq.unsqueeze(1) * self.inv_freqs.unsqueeze(0)
        return torch.cat((ang.sin(), ang.cos()), dim=-1)


class Transformer(nn.Module):
    def __init__(self, config, input_dim, max_episode_steps) -> None:
        super().__init__()
        self.config = config
        self.num_blocks = config["num_blocks"]
        self.embed_dim = config["embed_dim"]
        self.num_heads = config["num_heads"]
        self.max_episode_steps = max_episode_steps
        self.activation = nn.ReLU()
        self.linear_embedding = nn.Linear(input_dim, self.embed_dim)
        nn.init.orthogonal_(self.linear_embedding.weight, math.sqrt(2))
        self.pos_kind = config["positional_encoding"]
        if self.pos_kind == "relative":
            self.pos_embedding = SinusoidalPosition(dim=self.embed_dim)
            # table is evaluated once on the host with the same expression as upstream (bit-identical rows)
            self.register_buffer("_pos_table", self.pos_embedding(max_episode_steps).contiguous(), persistent=False)
        elif self.pos_kind == "learned":
            self.pos_embedding = nn.Parameter(torch.randn(self.max_episode_steps, self.embed_dim))
        self.transformer_blocks = nn.ModuleList(
            [TransformerBlock(self.embed_dim, self.num_heads, config) for _ in range(self.num_blocks)])

    def _pos(self):
        if self.pos_kind == "relative":
            return self._pos_table
        if self.pos_kind == "learned":
            return self.pos_embedding
        return None

    def _pack_gate_weights(self):
        """[Wr; Wz; Wg] and [Ur; Uz] of every GRU gate (transformer.py:287-298: the operands of the gates' concatenated GEMMs) into
        buffers that keep their address, by ONE multi-tensor copy for the whole model -- 2 cat launches per gate and step otherwise
        (16 at BASELINE config 5).  The buffers are saved for backward; TWO of them alternate, so a second grad-enabled forward pass
        before the backward pass of the first (auxiliary loss, twin evaluation) computes with its own copy; a third one overwrites
        the first's, which autograd reports (version counter) instead of computing with the wrong operands.  Returns the gates
        whose ``_train_cat`` was set (``forward_window`` clears them again whatever happens in between)."""
        gates = [g for blk in self.transformer_blocks if blk.use_gtrxl for g in (blk.gate1, blk.gate2)]
        if not gates:
            return gates
        D, dev = self.embed_dim, gates[0].Wr.weight.device
        bufs = getattr(self, "_gate_cat", None)
        if bufs is None or bufs[0].device != dev or bufs[0].shape[0] != len(gates):
            bufs = self._gate_cat = [torch.empty((len(gates), 5 * D, D), dtype=torch.float32, device=dev) for _ in range(2)]
            self._gate_cat_views = [[v for i, _ in enumerate(gates) for v in (b[i, :D], b[i, D: 2 * D], b[i, 2 * D: 3 * D],
                                                                              b[i, 3 * D: 4 * D], b[i, 4 * D:])] for b in bufs]
            self._gate_cat_turn = 0
        k = self._gate_cat_turn = 1 - self._gate_cat_turn
        buf = bufs[k]
        with torch.no_grad():
            torch._foreach_copy_(self._gate_cat_views[k], [m.weight for g in gates for m in (g.Wr, g.Wz, g.Wg, g.Ur, g.Uz)])
        for i, g in enumerate(gates):
            g._train_cat = (buf[i, : 3 * D], buf[i, 3 * D:])
        return gates

    def forward_window(self, h, spec: WindowSpec, want_items=True):
        """h [N, input_dim]; windows addressed by ``spec``.  Returns (h [N, D], new memory items [N, blocks, D]) -- the items
        are None with ``want_items=False`` (the optimisation phase never stores them)."""
        packed = self._pack_gate_weights() if (torch.is_grad_enabled() and h.is_cuda and h.dim() == 2) else []
        try:
            h = ops.linear_relu(self.linear_embedding, h)
            pos = None if spec.pos_included else self._pos()
            items = []
            h_res = None
            last = len(self.transformer_blocks) - 1
            for i, blk in enumerate(self.transformer_blocks):
                if want_items:
                    items.append(h.detach())
                if i < last:       # the state forked for the next block's two consumers (fused post-LN training path; None otherwise)
                    h, _, h_res = blk.forward_window(h, spec, i, pos, h_res=h_res, fork=True)
                else:
                    h, _ = blk.forward_window(h, spec, i, pos, h_res=h_res)
        finally:
            # a gate that did not consume its packed operands (eager path at D > 1024, an exception above) must not find a pointer into
            # the shared buffer in a later, non-window forward()
            for g in packed:
                g._train_cat = None
        return h, (torch.stack(items, dim=1) if want_items else None)

    def bank_with_positions(self, bank):
        """[E, T, blocks, D] episode bank with every row's positional row already added (same fp32 add the kernel would
        do per use).  Valid whenever a window's positional index equals its row index -- always the case in the
        optimisation phase (upstream trainer.py:271-274) -- and the table is not trainable.  Returns None otherwise."""
        if self.pos_kind != "relative":
            return None
        return bank + self._pos_table[None, : bank.shape[1], None, :]

    # ---------------------------------------------------------------- rollout K/V cache (weights frozen while sampling)
    def kv_projection_weights(self):
        """[blocks, D, 2D]: per block [Wk ; Wv]^T, plus norm_kv gains/biases [blocks, D] (or None)."""
        blocks = self.transformer_blocks
        w = torch.stack([torch.cat((b.attention.keys.weight, b.attention.values.weight), dim=0).t() for b in blocks])
        if blocks[0].layer_norm == "pre":
            g = torch.stack([b.norm_kv.weight for b in blocks])
            bb = torch.stack([b.norm_kv.bias for b in blocks])
            return w.contiguous(), g, bb, blocks[0].norm_kv.eps
        return w.contiguous(), None, None, 0.0

    def project_memory(self, items, pos_rows, weights):
        """items [M, blocks, D] memory items, pos_rows [M, D] (or None) -> cached projections [M, blocks, 2D] (K | V)."""
        w, g, bb, eps = weights
        x = items if pos_rows is None else items + pos_rows.unsqueeze(1)
        if g is not None:
            x = torch.nn.functional.layer_norm(x, (x.shape[-1],), None, None, eps) * g + bb
        return torch.bmm(x.transpose(0, 1), w).transpose(0, 1)

    def forward_cached(self, h, kv_spec: WindowSpec, items_out=None):
        """Rollout path of ``forward_window``: attention reads the K/V cache addressed by ``kv_spec``.
        ``items_out`` [blocks, N, D] (block-major, preallocated): every block's input -- the new memory items -- is produced
        directly in it (no stack / copy launches); it is then returned instead of the [N, blocks, D] stack."""
        nb = len(self.transformer_blocks)
        if items_out is not None:
            h = ops.linear_relu(self.linear_embedding, h, out=items_out[0])
            for i, blk in enumerate(self.transformer_blocks):
                h = blk.forward_cached(h, kv_spec, i, out=items_out[i + 1] if i + 1 < nb else None)
            return h, items_out
        h = ops.linear_relu(self.linear_embedding, h)
        items = []
        for i, blk in enumerate(self.transformer_blocks):
            items.append(h.detach())
            h = blk.forward_cached(h, kv_spec, i)
        return h, torch.stack(items, dim=1)

    def forward(self, h, memories, mask, memory_indices):
        """Upstream signature: memories [N, L, blocks, D] pre-gathered, mask [N, L], memory_indices [N, L] (long)."""
        return self.forward_window(h, WindowSpec.from_windows(memories, memory_indices, mask))
