"""Actor-critic with the episodic-memory transformer (API of upstream model.py:10-166).

``forward(obs, memory, memory_mask, memory_indices)`` keeps the upstream convention (pre-gathered memory windows);
``forward_banked(obs, spec)`` is the trainer's entry: windows are read in place from the episode bank.
"""
import math

import numpy as np
import torch
from torch import nn
from torch.distributions import Categorical
from torch.nn import functional as F

from etm import ops
from etm import lib as etm_lib
from etm.ops import WindowSpec
from transformer import Transformer


class IndexedObservations:
    """A minibatch of visual observations as (all observations of the update in NHWC memory order [n, H, W, C], int64 row
    indices): what ``buffer.samples_flat["obs"][mini_batch_indices]`` (buffer.py:84-91) denotes, without materialising it."""

    def __init__(self, bank_nhwc, index):
        self.bank, self.index = bank_nhwc, index


class ActorCriticModel(nn.Module):
    def __init__(self, config, observation_space, action_space_shape, max_episode_length):
        super().__init__()
        self.hidden_size = config["hidden_layer_size"]
        self.memory_layer_size = config["transformer"]["embed_dim"]
        self.observation_space_shape = tuple(observation_space.shape)
        self.max_episode_length = max_episode_length
        self.visual = len(self.observation_space_shape) > 1
        self.channels_last = True          # visual observations are kept NHWC for the optimisation phase (trainer._observations_channels_last)
        self.fused_encoder = bool(config.get("fused_rollout_encoder", True))
        self.train_encoder = bool(config.get("fused_train_encoder", True))     # False: library convolutions in the optimisation phase
        # round 6: "bf16x3" = the encoder's fp32 products as six bf16 MFMA products of exactly split operands (csrc/conv_b3*.hip; error
        # against float64 below the fp32-MFMA kernels'), "fp32" = the fp32-MFMA kernels of rounds 2 - 3.  Per model, handed to every call.
        self.encoder_products = str(config.get("encoder_products", "bf16x3"))
        if self.encoder_products not in ("bf16x3", "fp32"):
            raise ValueError(f"encoder_products must be 'bf16x3' or 'fp32', got {self.encoder_products!r}")
        self.fused_rollout_block = bool(config.get("fused_rollout_block", True))   # False: one launch per GEMM / LayerNorm / attention
        # round 5: GRU-gated layouts -- the step kernel of a worker GROUP (csrc/rollout_group.hip: workers as the rows of every product,
        # every matrix read once per group and step); False keeps the per-worker teams of csrc/rollout_fused.hip
        self.rollout_group_kernel = bool(config.get("rollout_group_kernel", True))
        self._rf = None
        self._rfg = None
        self._train_encoder_ok = None
        if self.visual:
            c = self.observation_space_shape[0]
            self.conv1 = nn.Conv2d(c, 32, 8, 4)
            self.conv2 = nn.Conv2d(32, 64, 4, 2, 0)
            self.conv3 = nn.Conv2d(64, 64, 3, 1, 0)
            for conv in (self.conv1, self.conv2, self.conv3):
                nn.init.orthogonal_(conv.weight, math.sqrt(2))
            self.conv_out_size = self.get_conv_output(self.observation_space_shape)
            feat = self.conv_out_size
        else:
            feat = self.observation_space_shape[0]
        self.lin_hidden = nn.Linear(feat, self.memory_layer_size)
        nn.init.orthogonal_(self.lin_hidden.weight, math.sqrt(2))
        self.transformer = Transformer(config["transformer"], self.memory_layer_size, self.max_episode_length)
        self.lin_policy = nn.Linear(self.memory_layer_size, self.hidden_size)
        nn.init.orthogonal_(self.lin_policy.weight, math.sqrt(2))
        self.lin_value = nn.Linear(self.memory_layer_size, self.hidden_size)
        nn.init.orthogonal_(self.lin_value.weight, math.sqrt(2))
        self.policy_branches = nn.ModuleList()
        for num_actions in action_space_shape:
            branch = nn.Linear(in_features=self.hidden_size, out_features=num_actions)
            nn.init.orthogonal_(branch.weight, math.sqrt(0.01))
            self.policy_branches.append(branch)
        self.value = nn.Linear(self.hidden_size, 1)
        nn.init.orthogonal_(self.value.weight, 1)

    # ------------------------------------------------------------------ forward
    # ------------------------------------------------------------------ rollout encoder (no grad): MFMA conv + bias + ReLU
    def _fused_encoder_ok(self, obs):
        if not (self.visual and self.fused_encoder and obs.is_cuda and not torch.is_grad_enabled()):
            return False
        _, _, hh, ww = obs.shape
        h1, w1 = (hh - 8) // 4 + 1, (ww - 8) // 4 + 1
        return ww % 4 == 0 and obs.shape[1] * 1 >= 1 and h1 >= 4 and w1 >= 4 and ((h1 - 4) // 2 + 1) >= 3 and ((w1 - 4) // 2 + 1) >= 3

    def refresh_rollout_weights(self):
        """(Re)build the weight copies / packings the rollout kernels read.  Buffers keep their address (the captured rollout graph
        reads them); called by the trainer at the start of every rollout and lazily whenever a weight's version counter moved.
        Round 6: dozens of small launches (2.3 ms per update for the gated layouts) -- from its third call on a graph replay when the
        trainer set ``graph_refresh`` (ops.ReplayAfterWarmup); the version bookkeeping follows every execution."""
        r = getattr(self, "_refresh_replay", None)
        if r is None:
            r = self._refresh_replay = ops.ReplayAfterWarmup(self._refresh_rollout_weights_now, self.lin_policy.weight.device,
                                                             after=self._mark_weight_versions, enabled=bool(getattr(self, "graph_refresh", False)),
                                                             what="refresh_rollout_weights")
        r.enabled = bool(getattr(self, "graph_refresh", False)) and r.device == self.lin_policy.weight.device
        r()

    def _mark_weight_versions(self):
        for mod in self.modules():
            if mod is not self and hasattr(mod, "_versions"):
                mod._wver = mod._versions()
        if self.visual:
            self._wver = (self.conv1.weight._version, self.conv2.weight._version, self.conv3.weight._version)

    def _refresh_rollout_weights_now(self):
        for mod in self.modules():
            if mod is not self and hasattr(mod, "refresh_rollout_weights"):
                mod.refresh_rollout_weights()
        with torch.no_grad():
            # concatenated hidden heads [lin_policy ; lin_value]: fixed-address copies read by the captured rollout graph
            w = torch.cat((self.lin_policy.weight, self.lin_value.weight), dim=0)
            b = torch.cat((self.lin_policy.bias, self.lin_value.bias), dim=0)
            if getattr(self, "_w_heads", None) is None or self._w_heads.device != w.device:
                self._w_heads, self._b_heads = w.contiguous(), b.contiguous()
                self._heads_lin = type("HeadsLinear", (), {})()
                self._heads_lin.weight, self._heads_lin.bias = self._w_heads, self._b_heads
            else:
                self._w_heads.copy_(w)
                self._b_heads.copy_(b)
        self._refresh_fused_block_weights()
        if not self.visual:
            return
        with torch.no_grad():
            # encoder weights in the K order of each layer's input layout (layer 1 reads NCHW: native order; layers 2, 3 read
            # NHWC: (ky, kx, c)), packed in MFMA fragment order (ops.conv_pack_weights)
            for name, conv, nhwc in (("_w1p", self.conv1, False), ("_w2p", self.conv2, True), ("_w3p", self.conv3, True)):
                w4 = conv.weight.permute(0, 2, 3, 1) if nhwc else conv.weight
                perm = ops.conv_pack_weights(w4.reshape(conv.out_channels, -1))
                buf = getattr(self, name, None)
                if buf is None or buf.shape != perm.shape or buf.device != perm.device:
                    setattr(self, name, perm.contiguous())
                else:
                    buf.copy_(perm)
            # the last layer as [(ky, kx, c), co]: what the fused conv3 + lin_hidden launch of a rollout step reads (ops.rollout_conv3_hidden)
            w3k = self.conv3.weight.permute(2, 3, 1, 0).reshape(-1, self.conv3.out_channels)
            if getattr(self, "_w3k", None) is None or self._w3k.shape != w3k.shape or self._w3k.device != w3k.device:
                self._w3k = w3k.contiguous()
            else:
                self._w3k.copy_(w3k)
            self._wver = (self.conv1.weight._version, self.conv2.weight._version, self.conv3.weight._version)

    def rollout_block_fusable(self):
        """Post-LN blocks without gates, one action branch, shapes inside etm_rollout_trxl's support: the trainer may run the
        transformer, the heads and the sampling of a rollout step as one kernel."""
        t = self.transformer
        blk = t.transformer_blocks[0]
        d = t.embed_dim
        if not (self.fused_rollout_block and blk.layer_norm in ("post", "pre") and len(self.policy_branches) == 1
                and t.linear_embedding.in_features == d):
            return False
        return ops.rollout_trxl_supported(d, t.num_heads, t.config["memory_length"], self.hidden_size,
                                          self.policy_branches[0].out_features, t.num_blocks)

    def _refresh_fused_block_weights(self):
        """Transposed ([in, out]) fixed-address copies of the matrices etm_rollout_trxl walks, and the host table of their
        device pointers (built once: the buffers keep their addresses, so captured graphs stay valid)."""
        if not self.lin_policy.weight.is_cuda or not self.rollout_block_fusable():
            self._rf = self._rfg = None
            return
        import ctypes
        t = self.transformer
        d = t.embed_dim
        team = etm_lib.load().etm_rollout_trxl_team(t.num_heads)
        merged = bool(etm_lib.load().etm_rollout_trxl_gate_merged(d, t.num_heads))

        def blocked(w_t):
            """[K, OUT] (transposed weight) -> member-blocked [P, K, OUT / P]: a team member's columns as one contiguous run."""
            k, out = w_t.shape
            return w_t.reshape(k, team, out // team).permute(1, 0, 2)

        with torch.no_grad():
            fresh = {"emb_t": blocked(t.linear_embedding.weight.t()),
                     "heads_t": blocked(torch.cat((self.lin_policy.weight, self.lin_value.weight), dim=0).t())}
            for i, blk in enumerate(t.transformer_blocks):
                fresh[f"wq_t{i}"] = blocked(blk.attention.queries.weight.t())
                fresh[f"wo_t{i}"] = blk.attention.fc_out.weight.t()          # K-split product: rows of a member are contiguous as they are
                fresh[f"wfc_t{i}"] = blocked(blk.fc[0].weight.t())
                if blk.use_gtrxl:    # GRU gates: [Wr | Wz | Wg]^T and [Ur | Uz]^T member-blocked, packed as the kernel expects for this shape
                    for gi, gate in ((1, blk.gate1), (2, blk.gate2)):
                        for key, names in (("wy", ("Wr", "Wz", "Wg")), ("ux", ("Ur", "Uz"))):
                            parts = torch.stack([blocked(getattr(gate, n).weight.t()) for n in names], dim=1)     # [P, j, D, DS]
                            fresh[f"g{gi}{key}_t{i}"] = parts.permute(0, 2, 1, 3) if merged else parts            # merged: [P, D, j, DS]
                        fresh[f"g{gi}ug_t{i}"] = blocked(gate.Ug.weight.t())
            if self.visual and self.lin_hidden.weight.shape[0] % 32 == 0:
                fresh["hid_t"] = self.lin_hidden.weight.t()     # [features, D]: etm_rollout_hidden_partial
            rf = getattr(self, "_rf", None)
            if rf is None or rf["emb_t"].device != self.lin_policy.weight.device:
                rf = {k: v.contiguous() for k, v in fresh.items()}
                rf["emb_b"], rf["heads_b"] = t.linear_embedding.bias, None
                ptrs = []
                for i, blk in enumerate(t.transformer_blocks):       # 19 pointers per block (include/etm_hip.h, etm_rollout_trxl)
                    ptrs += [rf[f"wq_t{i}"], rf[f"wo_t{i}"], blk.attention.fc_out.bias, blk.norm1.weight, blk.norm1.bias,
                             rf[f"wfc_t{i}"], blk.fc[0].bias, blk.norm2.weight, blk.norm2.bias]
                    for gi, gname in ((1, "gate1"), (2, "gate2")):
                        if blk.use_gtrxl:
                            ptrs += [rf[f"g{gi}wy_t{i}"], rf[f"g{gi}ux_t{i}"], rf[f"g{gi}ug_t{i}"], getattr(blk, gname).bg]
                        else:
                            ptrs += [None] * 4
                    ptrs += [blk.norm_kv.weight, blk.norm_kv.bias] if blk.layer_norm == "pre" else [None, None]
                rf["_keep"] = ptrs
                rf["blocks"] = (ctypes.c_void_p * len(ptrs))(*[None if p is None else p.data_ptr() for p in ptrs])
                rf["nb"], rf["H"], rf["eps"] = t.num_blocks, t.num_heads, t.transformer_blocks[0].norm1.eps
                rf["pre_ln"], rf["gtrxl"] = int(t.transformer_blocks[0].layer_norm == "pre"), int(t.transformer_blocks[0].use_gtrxl)
                self._rf = rf
            else:
                for k, v in fresh.items():
                    rf[k].copy_(v)
            self._rf["heads_b"] = self._b_heads          # concatenated hidden-head bias (refreshed above)
        self._refresh_group_block_weights()

    def _refresh_group_block_weights(self):
        """The packings etm_rollout_trxl_group reads (csrc/rollout_group.hip; GRU-gated blocks, groups of <= 8 workers): every
        [in, out] map transposed and COLUMN-BLOCKED over the launch's 32 workgroups -- [32][in][out / 32], a workgroup's slice of a
        map is one contiguous run --, the gates' maps of y / of x as [32][3][D][D / 32] / [32][2][D][D / 32], the hidden heads as
        [32][NCH][D][CH].  Fixed-address copies next to ``_rf`` (captured graphs read them); ``_rfg`` is None when the shape is not
        the group kernel's (the per-worker kernel then runs the gated layout as before)."""
        import ctypes
        t = self.transformer
        blk0 = t.transformer_blocks[0]
        d = t.embed_dim
        lib = etm_lib.load()
        A = self.policy_branches[0].out_features
        # (the kernel's [W, D] input rows and its D x D embedding slice assume linear_embedding.in_features == embed_dim)
        if not (self.rollout_group_kernel and self._rf is not None and blk0.use_gtrxl and t.linear_embedding.in_features == d
                and lib.etm_rollout_trxl_group_supported(d, t.num_heads, t.config["memory_length"], self.hidden_size, A, t.num_blocks, 1, 1)
                and lib.etm_rollout_trxl_team(t.num_heads) == t.num_heads):
            self._rfg = None
            return
        WG = lib.etm_rollout_trxl_group_grid()

        def colblock(w_t):
            """[K, OUT] (transposed weight) -> [32][K][OUT / 32]"""
            k, out = w_t.shape
            return w_t.reshape(k, WG, out // WG).permute(1, 0, 2)

        cbh = 2 * self.hidden_size // WG
        nch = (cbh + 15) // 16
        with torch.no_grad():
            heads_t = torch.cat((self.lin_policy.weight, self.lin_value.weight), dim=0).t()               # [D, 2 hid]
            fresh = {"emb_t": colblock(t.linear_embedding.weight.t()),
                     "heads_t": heads_t.reshape(d, WG, nch, cbh // nch).permute(1, 2, 0, 3)}              # [32][NCH][D][CH]
            for i, blk in enumerate(t.transformer_blocks):
                fresh[f"wq_t{i}"] = colblock(blk.attention.queries.weight.t())
                fresh[f"wo_t{i}"] = colblock(blk.attention.fc_out.weight.t())
                fresh[f"wfc_t{i}"] = colblock(blk.fc[0].weight.t())
                # the first gate's maps of y read a = fc_out(ctx) and nothing else does: W a = (W Wo) ctx + W bo.  Folded here, in
                # float64, once per refresh; the kernel multiplies the context rows with the folded maps and adds the folded bias rows
                # (one product and one all-gather fewer per block)
                wo64, bo64 = blk.attention.fc_out.weight.double(), blk.attention.fc_out.bias.double()
                for gi, gate in ((1, blk.gate1), (2, blk.gate2)):
                    maps = [getattr(gate, n).weight for n in ("Wr", "Wz", "Wg")]
                    if gi == 1:
                        folded = torch.stack([colblock((m.double() @ wo64).float().t()) for m in maps], dim=1)          # [32][3][D][CB]
                        bias = torch.stack([(m.double() @ bo64).float() for m in maps])                               # [3][D]
                        fresh[f"g{gi}wy_t{i}"] = torch.cat((folded.reshape(-1), bias.reshape(-1)))
                    else:
                        fresh[f"g{gi}wy_t{i}"] = torch.stack([colblock(m.t()) for m in maps], dim=1)
                    fresh[f"g{gi}ux_t{i}"] = torch.stack([colblock(getattr(gate, n).weight.t()) for n in ("Ur", "Uz")], dim=1)
                    fresh[f"g{gi}ug_t{i}"] = colblock(gate.Ug.weight.t())
            rg = getattr(self, "_rfg", None)
            if rg is None or rg["emb_t"].device != self.lin_policy.weight.device:
                rg = {k: v.contiguous() for k, v in fresh.items()}
                rg["emb_b"] = t.linear_embedding.bias
                ptrs = []
                for i, blk in enumerate(t.transformer_blocks):       # the 19 pointers per block of etm_rollout_trxl, group packings
                    ptrs += [rg[f"wq_t{i}"], rg[f"wo_t{i}"], blk.attention.fc_out.bias, blk.norm1.weight, blk.norm1.bias,
                             rg[f"wfc_t{i}"], blk.fc[0].bias, blk.norm2.weight, blk.norm2.bias]
                    for gi, gname in ((1, "gate1"), (2, "gate2")):
                        ptrs += [rg[f"g{gi}wy_t{i}"], rg[f"g{gi}ux_t{i}"], rg[f"g{gi}ug_t{i}"], getattr(blk, gname).bg]
                    ptrs += [blk.norm_kv.weight, blk.norm_kv.bias] if blk.layer_norm == "pre" else [None, None]
                rg["_keep"] = ptrs
                rg["blocks"] = (ctypes.c_void_p * len(ptrs))(*[None if q is None else q.data_ptr() for q in ptrs])
                rg["nb"], rg["H"], rg["eps"], rg["D"] = t.num_blocks, t.num_heads, blk0.norm1.eps, d
                rg["pre_ln"], rg["gtrxl"], rg["group"] = int(blk0.layer_norm == "pre"), 1, True
                self._rfg = rg
            else:
                for k, v in fresh.items():
                    rg[k].copy_(v)
            self._rfg["heads_b"] = self._b_heads
            if "hid_t" in self._rf:
                self._rfg["hid_t"] = self._rf["hid_t"]

    def _encode_fused(self, obs, obs_index=None, obs_rows=None, features_only=False):
        if getattr(self, "_w2p", None) is None or (not torch.cuda.is_current_stream_capturing()
                                                     and self._wver != (self.conv1.weight._version, self.conv2.weight._version,
                                                                        self.conv3.weight._version)):
            self.refresh_rollout_weights()
        n, c, hh, ww = obs.shape[-4:]      # with obs_index: obs is a stack [S, N, C, H, W] and the layer reads obs[obs_index]
        if obs_rows is not None:
            n = obs_rows[1] - obs_rows[0]
        x = ops.conv_relu(obs, self._w1p, self.conv1.bias, c, hh, ww, 8, 8, 4, False, False, index=obs_index, rows=obs_rows)  # -> NHWC
        h1, w1 = x.shape[1], x.shape[2]
        x = ops.conv_relu(x, self._w2p, self.conv2.bias, 32, h1, w1, 4, 4, 2, True, False)
        h2, w2 = x.shape[1], x.shape[2]
        if features_only == "conv2":          # the caller runs the last layer together with lin_hidden (ops.rollout_conv3_hidden)
            return x
        x = ops.conv_relu(x, self._w3p, self.conv3.bias, 64, h2, w2, 3, 3, 1, True, True)                                # -> NCHW
        if features_only:
            return x.reshape(n, -1)
        return ops.linear_relu(self.lin_hidden, x.reshape(n, -1))

    def _encode(self, obs, obs_index=None, obs_rows=None):
        """Observation encoder.  ``obs_index`` (int64 device scalar, fused rollout encoder only): ``obs`` is a time-major
        stack [S, N, C, H, W] and row obs[obs_index] is encoded (the row is selected on the device).  ``obs`` may be an
        ``IndexedObservations(bank_nhwc, index)``: the minibatch = bank_nhwc[index], gathered inside the first encoder layer."""
        if isinstance(obs, IndexedObservations):
            if self.visual and self.train_encoder and torch.is_grad_enabled():
                if self._train_encoder_ok is None:
                    self._train_encoder_ok = ops.encoder_train_supported(self.observation_space_shape, (self.conv1, self.conv2, self.conv3))
                if self._train_encoder_ok and ops.encoder_train_supported(self.observation_space_shape, (self.conv1, self.conv2, self.conv3),
                                                                            batch=int(obs.index.numel())):
                    feats = ops.encoder_train(obs.bank, self.conv1, self.conv2, self.conv3, index=obs.index, products=self.encoder_products)
                    if getattr(self, "_keep_encoder_features", False):      # data-parallel overlap: the backward pass is cut here
                        self._encoder_features = feats                      # (trainer._train_body_a1; released by _train_body_a2)
                    return ops.linear_relu_nhwc(feats, self.lin_hidden.weight, self.lin_hidden.bias, self.conv3.out_channels)
            obs = obs.bank.index_select(0, obs.index).permute(0, 3, 1, 2)      # NCHW view of the gathered NHWC rows
        if obs_index is not None:
            if not self._fused_encoder_ok(obs[0]):
                raise RuntimeError("obs_index needs the fused rollout encoder (visual observations, no grad)")
            return self._encode_fused(obs, obs_index, obs_rows)
        h = obs
        if self._fused_encoder_ok(obs):
            return self._encode_fused(obs)
        if self.visual and self.train_encoder and obs.is_cuda and torch.is_grad_enabled() and obs.dim() == 4:
            if self._train_encoder_ok is None:
                self._train_encoder_ok = ops.encoder_train_supported(self.observation_space_shape, (self.conv1, self.conv2, self.conv3))
            if self._train_encoder_ok and ops.encoder_train_supported(self.observation_space_shape, (self.conv1, self.conv2, self.conv3),
                                                                        batch=int(obs.shape[0])):
                # optimisation phase: the three relu(conv2d) layers forward and backward on the hand-written MFMA kernels
                # (NHWC activations; the trainer hands over an NCHW view of NHWC memory, which permutes back for free)
                feats = ops.encoder_train(obs.permute(0, 2, 3, 1), self.conv1, self.conv2, self.conv3, products=self.encoder_products)      # (h, w, c) flatten order
                return ops.linear_relu_nhwc(feats, self.lin_hidden.weight, self.lin_hidden.bias, self.conv3.out_channels)
        if self.visual:
            if self.channels_last and h.is_cuda:
                # NHWC activations: MIOpen's implicit-GEMM kernels run without layout transposes (1.35 vs 2.1 ms for
                # forward+backward of the three convolutions at N = 2048); weights, logical shapes and state_dict are unchanged
                h = h.contiguous(memory_format=torch.channels_last)
            h = F.relu(self.conv1(h))
            h = F.relu(self.conv2(h))
            h = F.relu(self.conv3(h))
            h = h.reshape(h.shape[0], -1)
        return ops.linear_relu(self.lin_hidden, h)

    def forward_state(self, obs, spec: WindowSpec, want_items=False):
        """Encoder + transformer: -> (h [N, D] in front of the hidden heads (model.py:100), new memory items or None)."""
        return self.transformer.forward_window(self._encode(obs), spec, want_items)

    def forward_logits(self, obs, spec: WindowSpec, want_items=True):
        """-> (list of raw logits per branch, value [N], new memory items [N, blocks, D] or None)."""
        h, memory = self.transformer.forward_window(self._encode(obs), spec, want_items)
        h_policy = ops.linear_relu(self.lin_policy, h)
        h_value = ops.linear_relu(self.lin_value, h)
        value = self.value(h_value).reshape(-1)
        return [branch(h_policy) for branch in self.policy_branches], value, memory

    def rollout_heads_fusable(self):
        """Single-branch policy with the concatenated hidden heads built: the trainer may run output heads + sampling as one
        kernel on ``forward_hidden_cached``'s result."""
        return len(self.policy_branches) == 1 and getattr(self, "_w_heads", None) is not None and not torch.is_grad_enabled()

    def forward_hidden_cached(self, obs, kv_spec: WindowSpec, items_out=None, obs_index=None, raw=False, obs_rows=None):
        """Rollout path up to the hidden heads: -> (h2 [N, 2*hidden] = [relu(lin_policy(h)) | relu(lin_value(h))], memory).
        ``raw=True``: h2 are the PRE-activations (plain GEMM without epilogue); the caller applies relu(h2 + self._b_heads)
        (``ops.rollout_policy(h_bias=...)`` does it on the fly)."""
        h, memory = self.transformer.forward_cached(self._encode(obs, obs_index, obs_rows), kv_spec, items_out)
        if raw:
            return F.linear(h, self._w_heads), memory
        return ops.linear_relu(self._heads_lin, h), memory

    def forward_logits_cached(self, obs, kv_spec: WindowSpec, items_out=None, obs_index=None, obs_rows=None):
        """Rollout path (no grad): like ``forward_logits`` but attention reads the per-worker K/V cache.  With ``items_out``
        [blocks, N, D] the new memory items are written there (block-major) and returned in that layout."""
        h, memory = self.transformer.forward_cached(self._encode(obs, obs_index, obs_rows), kv_spec, items_out)
        if len(self.policy_branches) == 1 and getattr(self, "_w_heads", None) is not None and not torch.is_grad_enabled():
            # [lin_policy ; lin_value] as ONE GEMM (+ReLU epilogue), then both output heads in one small kernel
            h2 = ops.linear_relu(self._heads_lin, h)
            logits, value = ops.rollout_heads(h2, self.policy_branches[0], self.value)
            return [logits], value, memory
        h_policy = ops.linear_relu(self.lin_policy, h)
        h_value = ops.linear_relu(self.lin_value, h)
        value = self.value(h_value).reshape(-1)
        return [branch(h_policy) for branch in self.policy_branches], value, memory

    def forward_banked(self, obs, spec: WindowSpec):
        logits, value, memory = self.forward_logits(obs, spec)
        return [Categorical(logits=l, validate_args=False) for l in logits], value, memory

    def forward(self, obs, memory, memory_mask, memory_indices):
        """Upstream signature; memory [N, L, blocks, D] is the pre-gathered window."""
        return self.forward_banked(obs, WindowSpec.from_windows(memory, memory_indices, memory_mask))

    def get_conv_output(self, shape) -> int:
        with torch.no_grad():
            o = self.conv3(self.conv2(self.conv1(torch.zeros(1, *shape))))
        return int(np.prod(o.size()))

    # ------------------------------------------------------------------ gradient monitoring (keys of upstream :128-151)
    def _grad_groups(self):
        groups = {}
        if self.visual:
            groups["encoder"] = [self.conv1, self.conv2, self.conv3]
        groups["linear_layer"] = [self.lin_hidden]
        for i, block in enumerate(self.transformer.transformer_blocks):
            groups["transformer_block_" + str(i)] = [block]
        for i, head in enumerate(self.policy_branches):
            groups["policy_head_" + str(i)] = [head]
        groups["lin_policy"] = [self.lin_policy]
        groups["value"] = [self.lin_value, self.value]
        groups["model"] = [self, self.value]
        return groups

    def grad_norms_device(self):
        """{key: 0-dim device tensor}: no host synchronisation (the trainer syncs once per update)."""
        out = {}
        for key, modules in self._grad_groups().items():
            grads = [p.grad.reshape(-1) for m in modules for p in m.parameters() if p.grad is not None]
            out[key] = torch.linalg.norm(torch.cat(grads)) if grads else None
        return out

    def get_grad_norm(self):
        return {k: (v.item() if v is not None else None) for k, v in self.grad_norms_device().items()}
