"""Oracle (CPU, fp32): functional restatement of the reference model path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Parity of this file is
pinned by ``tests/golden/*.npz`` (generated from the real reference by
``tests/golden/make_golden.py``) through ``tests/test_oracle_golden.py``.

All functions take ``sd``: a flat ``{name: fp32 tensor}`` dict that uses the
reference's ``state_dict`` key names, so the same weights
can be loaded into the reference, the oracle and the MI355X build.

Citations are into ``/root/reference``.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- tables
def window_tables(memory_length: int, max_episode_length: int):
    """Mask table [L, L] (float 0/1) and sliding-window index table [T, L] (int64).

    Restates trainer.py:78 (``tril(ones, diagonal=-1)``) and trainer.py:88-90
    (L-1 copies of ``[0..L-1]`` followed by one window per start ``0..T-L``).
    Built with integer loops on purpose: this is the exact/bit-exact contract.
    """
    L, T = int(memory_length), int(max_episode_length)
    if T < L:
        raise ValueError("max_episode_length must be >= memory_length (trainer.py:89 stacks range(T-L+1) windows)")
    mask = torch.zeros((L, L), dtype=torch.float32)
    for row in range(L):
        mask[row, :row] = 1.0
    idx = torch.empty((T, L), dtype=torch.int64)
    for step in range(T):
        first = 0 if step < L - 1 else step - (L - 1)
        idx[step] = torch.arange(first, first + L, dtype=torch.int64)
    return mask, idx


def rollout_window(step: int, memory_length: int):
    """(mask row id, first window index) used at episode step ``step`` while sampling.

    trainer.py:165-166: mask row = clip(step, 0, L-1); indices = table[step].
    """
    L = memory_length
    return min(max(step, 0), L - 1), (0 if step < L - 1 else step - (L - 1))


def last_value_window(step: int, memory_length: int):
    """Window used by ``get_last_value`` (trainer.py:230-232): [clip(step-L,0), clip(step,L))."""
    L = memory_length
    start = max(step - L, 0)
    end = max(step, L)
    return start, end


def sinusoid_table(embed_dim: int, seq_len: int, min_timescale: float = 2.0, max_timescale: float = 1e4):
    """transformer.py:174-186.  Row i encodes position ``seq_len-1-i``; [sin | cos] halves."""
    freqs = torch.arange(0, embed_dim, min_timescale)
    inv_freqs = max_timescale ** (-freqs / embed_dim)
    seq = torch.arange(seq_len - 1, -1, -1.0)
    ang = seq[:, None] * inv_freqs[None, :]
    return torch.cat((ang.sin(), ang.cos()), dim=-1)


def gather_window(memory: torch.Tensor, indices: torch.Tensor):
    """utils.py:52-75 ``batched_index_select(memory, 1, indices)``: [N,T,...],[N,L] -> [N,L,...]."""
    n = memory.shape[0]
    rows = torch.arange(n)[:, None].expand_as(indices)
    return memory[rows, indices]


# --------------------------------------------------------------------------- layers
def _ln(sd, prefix, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def mha(sd, prefix: str, num_heads: int, values, keys, queries, mask):
    """transformer.py:31-86.  values/keys [N,L,D], queries [N,Q,D], mask [N,L] (0 => masked).

    Quirks kept: scale is sqrt(embed_dim) (Q1, :69); ``-1e20`` fill happens
    *before* the scale (Q2, :66); returns (out [N,Q,D], attention [N,H,Q,L]).
    """
    n, klen, d = keys.shape
    qlen = queries.shape[1]
    hd = d // num_heads
    v = (values @ sd[prefix + ".values.weight"].t()).reshape(n, values.shape[1], num_heads, hd)
    k = (keys @ sd[prefix + ".keys.weight"].t()).reshape(n, klen, num_heads, hd)
    q = (queries @ sd[prefix + ".queries.weight"].t()).reshape(n, qlen, num_heads, hd)
    energy = torch.einsum("nqhd,nkhd->nhqk", q, k)
    if mask is not None:
        energy = energy.masked_fill(mask[:, None, None, :] == 0, float("-1e20"))
    attention = torch.softmax(energy / (d ** (1 / 2)), dim=3)
    ctx = torch.einsum("nhql,nlhd->nqhd", attention, v).reshape(n, qlen, d)
    out = ctx @ sd[prefix + ".fc_out.weight"].t() + sd[prefix + ".fc_out.bias"]
    return out, attention


def gru_gate(sd, prefix: str, x, y):
    """transformer.py:287-298 (GTrXL gate; six bias-free DxD maps and ``bg``)."""
    lin = lambda name, t: t @ sd[prefix + "." + name + ".weight"].t()
    r = torch.sigmoid(lin("Wr", y) + lin("Ur", x))
    z = torch.sigmoid(lin("Wz", y) + lin("Uz", x) - sd[prefix + ".bg"])
    h = torch.tanh(lin("Wg", y) + lin("Ug", r * x))
    return (1 - z) * x + z * h


def block(sd, prefix: str, tcfg: dict, value, key, query, mask):
    """transformer.py:117-172.  ``tcfg`` is the YAML ``transformer`` dict."""
    ln = tcfg["layer_norm"]
    gated = bool(tcfg.get("gtrxl", False))
    if ln == "pre":
        query_ = _ln(sd, prefix + ".norm1", query)
        value = _ln(sd, prefix + ".norm_kv", value)
        key = value
    else:
        query_ = query
    att, att_w = mha(sd, prefix + ".attention", tcfg["num_heads"], value, key, query_, mask)
    h = gru_gate(sd, prefix + ".gate1", query, att) if gated else att + query
    if ln == "post":
        h = _ln(sd, prefix + ".norm1", h)
    h_ = _ln(sd, prefix + ".norm2", h) if ln == "pre" else h
    fwd = torch.relu(h_ @ sd[prefix + ".fc.0.weight"].t() + sd[prefix + ".fc.0.bias"])
    out = gru_gate(sd, prefix + ".gate2", h, fwd) if gated else fwd + h
    if ln == "post":
        out = _ln(sd, prefix + ".norm2", out)
    return out, att_w


def transformer(sd, tcfg: dict, max_episode_steps: int, h, memories, mask, memory_indices, prefix="transformer"):
    """transformer.py:222-253.  memories [N,L,nb,D]; returns (h [N,D], new memory [N,nb,D], att list)."""
    h = torch.relu(h @ sd[prefix + ".linear_embedding.weight"].t() + sd[prefix + ".linear_embedding.bias"])
    pe = tcfg["positional_encoding"]
    if pe == "relative":
        table = sinusoid_table(tcfg["embed_dim"], max_episode_steps)
        memories = memories + table[memory_indices].unsqueeze(2)
    elif pe == "learned":
        memories = memories + sd[prefix + ".pos_embedding"][memory_indices].unsqueeze(2)
    new_items, atts = [], []
    for i in range(tcfg["num_blocks"]):
        new_items.append(h.detach())
        h, a = block(sd, f"{prefix}.transformer_blocks.{i}", tcfg, memories[:, :, i], memories[:, :, i], h.unsqueeze(1), mask)
        atts.append(a)
        h = h.squeeze()
        if h.dim() == 1:  # Q6: N == 1
            h = h.unsqueeze(0)
    return h, torch.stack(new_items, dim=1), atts


def actor_critic(sd, config: dict, obs, memory, memory_mask, memory_indices, max_episode_length: int):
    """model.py:71-112.  Returns (list of logits per branch, value [N], new memory [N,nb,D]).

    The reference wraps the logits in ``Categorical(logits=...)``; the oracle
    returns raw logits (normalisation is done where it is consumed).
    """
    h = obs
    if obs.dim() > 2:
        h = torch.relu(F.conv2d(h, sd["conv1.weight"], sd["conv1.bias"], stride=4))
        h = torch.relu(F.conv2d(h, sd["conv2.weight"], sd["conv2.bias"], stride=2))
        h = torch.relu(F.conv2d(h, sd["conv3.weight"], sd["conv3.bias"], stride=1))
        h = h.reshape(h.shape[0], -1)
    h = torch.relu(h @ sd["lin_hidden.weight"].t() + sd["lin_hidden.bias"])
    h, new_mem, _ = transformer(sd, config["transformer"], max_episode_length, h, memory, memory_mask, memory_indices)
    h_pi = torch.relu(h @ sd["lin_policy.weight"].t() + sd["lin_policy.bias"])
    h_v = torch.relu(h @ sd["lin_value.weight"].t() + sd["lin_value.bias"])
    value = (h_v @ sd["value.weight"].t() + sd["value.bias"]).reshape(-1)
    logits = []
    j = 0
    while f"policy_branches.{j}.weight" in sd:
        logits.append(h_pi @ sd[f"policy_branches.{j}.weight"].t() + sd[f"policy_branches.{j}.bias"])
        j += 1
    return logits, value, new_mem


# --------------------------------------------------------------------------- init
def init_state_dict(config: dict, obs_shape, action_space_shape, max_episode_length: int, seed: int = 0):
    """Fresh parameters with the reference's initialisers and key names.

    model.py:27-69 (orthogonal gains sqrt2 / sqrt0.01 / 1), transformer.py:205-220
    (orthogonal embedding, ``randn`` learned positions), transformer.py:262-285
    (xavier-uniform gate maps, ``bg`` fill).  torch-default (kaiming-uniform)
    for everything the reference leaves at its default.  The draw order differs
    from the reference's constructor order; distributions are the same.
    """
    g = torch.Generator().manual_seed(seed)
    tcfg = config["transformer"]
    d, hid = tcfg["embed_dim"], config["hidden_layer_size"]
    sd = {}

    def default_linear(name, out_f, in_f, bias=True):
        bound = 1.0 / math.sqrt(in_f)
        sd[name + ".weight"] = (torch.rand((out_f, in_f), generator=g) * 2 - 1) * bound
        if bias:
            sd[name + ".bias"] = (torch.rand((out_f,), generator=g) * 2 - 1) * bound

    def orthogonal_(name, gain):
        w = sd[name]
        flat = torch.randn((w.shape[0], w[0].numel()), generator=g)
        transposed = flat.shape[0] < flat.shape[1]
        if transposed:
            flat = flat.t()
        q, r = torch.linalg.qr(flat)
        q = q * torch.sign(torch.diagonal(r))
        if transposed:
            q = q.t()
        sd[name] = (gain * q).reshape(w.shape).contiguous()

    def xavier(name, n):
        bound = math.sqrt(6.0 / (n + n))
        sd[name] = (torch.rand((n, n), generator=g) * 2 - 1) * bound

    if len(obs_shape) > 1:
        for nm, (co, ci, k) in (("conv1", (32, obs_shape[0], 8)), ("conv2", (64, 32, 4)), ("conv3", (64, 64, 3))):
            fan_in = ci * k * k
            bound = 1.0 / math.sqrt(fan_in)
            sd[nm + ".weight"] = torch.empty((co, ci, k, k))
            sd[nm + ".bias"] = (torch.rand((co,), generator=g) * 2 - 1) * bound
            orthogonal_(nm + ".weight", math.sqrt(2))
        hh = (obs_shape[1] - 8) // 4 + 1
        ww = (obs_shape[2] - 8) // 4 + 1
        hh, ww = (hh - 4) // 2 + 1, (ww - 4) // 2 + 1
        hh, ww = hh - 2, ww - 2
        in_feat = 64 * hh * ww
    else:
        in_feat = obs_shape[0]
    default_linear("lin_hidden", d, in_feat)
    orthogonal_("lin_hidden.weight", math.sqrt(2))
    default_linear("transformer.linear_embedding", d, d)
    orthogonal_("transformer.linear_embedding.weight", math.sqrt(2))
    if tcfg["positional_encoding"] == "learned":
        sd["transformer.pos_embedding"] = torch.randn((max_episode_length, d), generator=g)
    elif tcfg["positional_encoding"] == "relative":
        sd["transformer.pos_embedding.inv_freqs"] = 1e4 ** (-torch.arange(0, d, 2.0) / d)
    for i in range(tcfg["num_blocks"]):
        p = f"transformer.transformer_blocks.{i}"
        for nm in ("values", "keys", "queries"):
            default_linear(f"{p}.attention.{nm}", d, d, bias=False)
        default_linear(f"{p}.attention.fc_out", d, d)
        if tcfg.get("gtrxl", False):
            for gate in ("gate1", "gate2"):
                for nm in ("Wr", "Ur", "Wz", "Uz", "Wg", "Ug"):
                    xavier(f"{p}.{gate}.{nm}.weight", d)
                sd[f"{p}.{gate}.bg"] = torch.full((d,), float(tcfg["gtrxl_bias"]))
        norms = ["norm1", "norm2"] + (["norm_kv"] if tcfg["layer_norm"] == "pre" else [])
        for nm in norms:
            sd[f"{p}.{nm}.weight"] = torch.ones(d)
            sd[f"{p}.{nm}.bias"] = torch.zeros(d)
        default_linear(f"{p}.fc.0", d, d)
    default_linear("lin_policy", hid, d)
    orthogonal_("lin_policy.weight", math.sqrt(2))
    default_linear("lin_value", hid, d)
    orthogonal_("lin_value.weight", math.sqrt(2))
    for j, n_act in enumerate(action_space_shape):
        default_linear(f"policy_branches.{j}", n_act, hid)
        orthogonal_(f"policy_branches.{j}.weight", math.sqrt(0.01))
    default_linear("value", 1, hid)
    orthogonal_("value.weight", 1.0)
    return {k: v.float().contiguous() for k, v in sd.items()}
