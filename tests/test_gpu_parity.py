"""-m gpu: the HIP path (through libetm_hip.so) against golden vectors from the reference and against the oracle.

Tolerances (fp32; differences come only from summation order and libm-vs-device exp): forward 2e-5 abs + 1e-4 rel
(values up to O(10)), attention weights 2e-6, gradients 2e-4 of the tensor norm, window indices / masks / GAE: exact.
"""
import json
import os

import numpy as np
import pytest
import torch

import detgen as dg

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch.device("cuda", 0)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def shapes_of(z, tag):
    keys = [str(k) for k in z[tag + "keys"]]
    shapes = [tuple(int(x) for x in str(s).split(",") if x) for s in z[tag + "shapes"]]
    return keys, shapes


def load_det(module, case, keys, shapes):
    gen = dg.det_state_dict(case, keys, shapes)
    sd = module.state_dict()
    assert list(sd.keys()) == keys, "state_dict keys differ from the reference"
    module.load_state_dict({k: (torch.from_numpy(gen[k]) if k in gen else sd[k]) for k in keys})


def close(a, b, atol=2e-5, rtol=1e-4, what=""):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    assert (err <= atol + rtol * np.abs(b)).all(), f"{what}: max err {err.max():.3e} (ref max {np.abs(b).max():.3e})"


def grad_close(g, z, tag, key, n_sample, rel=2e-4):
    g = g.detach().cpu().numpy()
    norm = float(z[tag + "grad_norm/" + key])
    tol = rel * max(norm, 1e-6)
    assert abs(float(np.linalg.norm(g.astype(np.float64))) - norm) <= 10 * tol, (key, np.linalg.norm(g), norm)
    close(dg.sample(g, n_sample), z[tag + "grad_sample/" + key], atol=tol, rtol=1e-3, what=key)


@pytest.fixture(params=["folded", "dense"])
def attn_impl(request):
    """Both kernel families behind ops.mha: the folded HBM-bound pass (default) and the dense fp32-MFMA contractions."""
    from etm import ops
    ops.set_attention_impl(request.param)
    yield request.param
    ops.set_attention_impl("folded")


def test_library_loaded_and_no_fallback():
    from etm import lib, ops
    h = lib.load()
    assert h.etm_abi_version() == lib.ABI_VERSION
    with pytest.raises(RuntimeError):
        ops.gae(torch.zeros(2, 2), torch.zeros(2, 2, dtype=torch.bool), torch.zeros(2, 2), torch.zeros(2), 0.9, 0.9)


# ------------------------------------------------------------------ kernel #1 vs reference golden
def test_mha_module_vs_reference(golden_dir, attn_impl):
    from transformer import MultiHeadAttention
    dev = _dev()
    z = load(golden_dir, "mha.npz")
    for case in sorted({k.split("/")[0] for k in z.files}):
        tag = case + "/"
        D, H, L, n = (int(x) for x in z[tag + "dims"])
        keys, shapes = shapes_of(z, tag)
        m = MultiHeadAttention(D, H)
        load_det(m, case, keys, shapes)
        m.to(dev)
        kv = torch.from_numpy(dg.det_normal(case, "kv", (n, L, D))).to(dev)
        q = torch.from_numpy(dg.det_normal(case, "q", (n, 1, D))).to(dev).requires_grad_(True)
        mask = torch.from_numpy(dg.leading_mask(case, n, L)).to(dev)
        out, att = m(kv, kv, q, mask)
        assert out.shape == (n, 1, D) and att.shape == (n, H, 1, L)
        close(out, z[tag + "out"], what=case + " out")
        close(att, z[tag + "att"], atol=2e-6, rtol=1e-4, what=case + " att")
        go = torch.from_numpy(dg.det_normal(case, "gout", (n, 1, D))).to(dev)
        (out * go).sum().backward()
        close(q.grad, z[tag + "gq"], atol=2e-5, rtol=1e-3, what=case + " gq")
        for k, p in m.named_parameters():
            grad_close(p.grad, z, tag, k, 384)


def test_transformer_variants_vs_reference(golden_dir, attn_impl):
    from transformer import Transformer
    dev = _dev()
    z = load(golden_dir, "transformer.npz")
    cases = sorted({k.split("/")[0] for k in z.files}, key=lambda s: int(s.split("_v")[1]))
    assert len(cases) == 27
    for case in cases:
        tag = case + "/"
        info = json.loads(str(z[tag + "cfg_json"]))
        cfg, T, n = info["cfg"], info["T"], info["n"]
        D, L, nb = cfg["embed_dim"], cfg["memory_length"], cfg["num_blocks"]
        keys, shapes = shapes_of(z, tag)
        tr = Transformer(cfg, D, T)
        load_det(tr, case, keys, shapes)
        tr.to(dev)
        h = torch.from_numpy(dg.det_normal(case, "h", (n, D))).to(dev)
        mem = torch.from_numpy(dg.det_normal(case, "mem", (n, L, nb, D), 0.5)).to(dev)
        mask = torch.from_numpy(dg.leading_mask(case, n, L)).to(dev)
        idx = torch.from_numpy(dg.window_indices(case, n, L, T)).to(dev)
        out, new_mem = tr(h, mem, mask, idx)
        assert out.shape == (n, D) and new_mem.shape == (n, nb, D)
        close(out, z[tag + "out"], atol=5e-5, what=case + " out")
        close(new_mem, z[tag + "new_mem"], atol=5e-5, what=case + " new_mem")
        go = torch.from_numpy(dg.det_normal(case, "gout", (n, D))).to(dev)
        (out * go).sum().backward()
        for k, p in tr.named_parameters():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            grad_close(g, z, tag, k, 96, rel=3e-4)


def test_actor_critic_vs_reference(golden_dir):
    from types import SimpleNamespace
    from model import ActorCriticModel
    dev = _dev()
    z = load(golden_dir, "model.npz")
    for case in ("model_vec", "model_img", "model_img_post"):
        tag = case + "/"
        info = json.loads(str(z[tag + "cfg_json"]))
        cfg, T, n, obs_shape = info["cfg"], info["T"], info["n"], tuple(info["obs_shape"])
        t = cfg["transformer"]
        keys, shapes = shapes_of(z, tag)
        m = ActorCriticModel(cfg, SimpleNamespace(shape=obs_shape), tuple(info["act"]), T)
        load_det(m, case, keys, shapes)
        m.to(dev)
        obs = torch.from_numpy(np.abs(dg.det_normal(case, "obs", (n,) + obs_shape, 0.4)).clip(0, 1)).to(dev)
        mem = torch.from_numpy(dg.det_normal(case, "mem", (n, t["memory_length"], t["num_blocks"], t["embed_dim"]), 0.3)).to(dev)
        mask = torch.from_numpy(dg.leading_mask(case, n, t["memory_length"])).to(dev)
        idx = torch.from_numpy(dg.window_indices(case, n, t["memory_length"], T)).to(dev)
        pi, value, new_mem = m(obs, mem, mask, idx)
        close(pi[0].logits, z[tag + "log_probs_all"], atol=1e-4, what=case + " logp")
        close(value, z[tag + "value"], atol=1e-4, what=case + " value")
        close(new_mem, z[tag + "new_mem"], atol=1e-4, what=case + " mem")
        loss = (pi[0].logits * torch.from_numpy(dg.det_normal(case, "gl", tuple(pi[0].logits.shape))).to(dev)).sum() + \
               (value * torch.from_numpy(dg.det_normal(case, "gv", tuple(value.shape))).to(dev)).sum()
        loss.backward()
        for k, p in m.named_parameters():
            if k.startswith("conv"):
                # THESE fixtures (batch of 4, ~85k convolution activations) hold units within fp32 rounding of the ReLU kink, whose gates
                # any two fp32 evaluation orders may decide differently -- one flipped gate moves the encoder gradients of a batch of
                # 4 by per cents.  The encoder gradients are therefore compared where no gate is ambiguous, at full tolerance and
                # element-wise: test_actor_critic_gradients_away_from_the_relu_kink (2e-6 of the norm),
                # test_kink_free_update_vs_reference (BASELINE sizes, against the float64 evaluation) and
                # test_train_encoder_fwd_bwd_vs_float64_convs.  (Round 3 kept a `cos > 0.98` check here; it asserted nothing the
                # three tests above do not assert better and is gone.)
                continue
            grad_close(p.grad, z, tag, k, 96, rel=1e-3)


# Measured on the MI355X (round 3): 3.4e-7 (profiles/r03/teacher_forced_measured.jsonl, case model_nokink).
NOKINK_GRAD_REL = 2e-6


def test_actor_critic_gradients_away_from_the_relu_kink(golden_dir):
    """model_nokink.npz: the procedural case was chosen so that every ReLU input of the reference's forward pass keeps
    |x| > 2e-5 (make_golden.py:golden_model_nokink) -- every gate is decided identically by any fp32 evaluation order, so ALL
    gradients, the encoder's included, are compared element-wise with the reference (model.py:71-112 backward)."""
    from types import SimpleNamespace
    from model import ActorCriticModel
    dev = _dev()
    z = load(golden_dir, "model_nokink.npz")
    case, tag = str(z["case_name"]), "case/"
    assert float(z["relu_margin"]) > 2e-5
    info = json.loads(str(z[tag + "cfg_json"]))
    cfg, T, n, obs_shape = info["cfg"], info["T"], info["n"], tuple(info["obs_shape"])
    t = cfg["transformer"]
    keys, shapes = shapes_of(z, tag)
    m = ActorCriticModel(cfg, SimpleNamespace(shape=obs_shape), tuple(info["act"]), T)
    load_det(m, case, keys, shapes)
    m.to(dev)
    obs = torch.from_numpy(np.abs(dg.det_normal(case, "obs", (n,) + obs_shape, 0.4)).clip(0, 1)).to(dev)
    mem = torch.from_numpy(dg.det_normal(case, "mem", (n, t["memory_length"], t["num_blocks"], t["embed_dim"]), 0.3)).to(dev)
    mask = torch.from_numpy(dg.leading_mask(case, n, t["memory_length"])).to(dev)
    idx = torch.from_numpy(dg.window_indices(case, n, t["memory_length"], T)).to(dev)
    pi, value, new_mem = m(obs, mem, mask, idx)
    close(pi[0].logits, z[tag + "log_probs_all"], atol=2e-5, what="logp")
    close(value, z[tag + "value"], atol=2e-5, what="value")
    close(new_mem, z[tag + "new_mem"], atol=2e-5, what="mem")
    loss = (pi[0].logits * torch.from_numpy(dg.det_normal(case, "gl", tuple(pi[0].logits.shape))).to(dev)).sum() + \
           (value * torch.from_numpy(dg.det_normal(case, "gv", tuple(value.shape))).to(dev)).sum()
    loss.backward()
    assert m._train_encoder_ok, "the hand-written encoder kernels ran"
    worst = 0.0
    for k, p in m.named_parameters():
        want = z[tag + "grad_sample/" + k].astype(np.float64)
        got = dg.sample(p.grad.detach().cpu().numpy(), 384).astype(np.float64)
        norm = float(z[tag + "grad_norm/" + k])
        rel = float(np.abs(got - want).max()) / max(norm, 1e-12)
        worst = max(worst, rel)
        assert rel <= NOKINK_GRAD_REL, (k, rel)
        assert abs(float(np.linalg.norm(p.grad.detach().cpu().numpy().astype(np.float64))) - norm) <= 10 * NOKINK_GRAD_REL * norm, k
    print(f"[nokink] worst element error of any gradient tensor / its norm: {worst:.2e} (bound {NOKINK_GRAD_REL:.0e})")
    if os.environ.get("ETM_TF_MEASURE_LOG"):
        with open(os.environ["ETM_TF_MEASURE_LOG"], "a") as f:
            f.write(json.dumps({"case": "model_nokink", "grad_worst_element_over_norm": worst}) + "\n")


# ------------------------------------------------------------------ kernel #1 vs the oracle: banked gather, LN, positions, Q5
@pytest.mark.parametrize("D,H,L,N,ln,pos", [(384, 4, 64, 37, False, True), (384, 4, 128, 9, True, True), (128, 1, 32, 50, True, False),
                                            (64, 1, 32, 7, True, True), (256, 4, 96, 6, False, False), (384, 4, 118, 5, True, True),
                                            (96, 3, 5, 11, False, True), (1024, 8, 16, 3, True, True), (32, 1, 1, 1, False, False),
                                            (128, 1, 128, 2, True, True), (64, 2, 33, 130, False, True), (512, 4, 64, 1, True, False)])
def test_mha_banked_vs_oracle(D, H, L, N, ln, pos, attn_impl):
    from etm import ops
    from oracle import ref_model as rm
    dev = _dev()
    g = torch.Generator().manual_seed(D + L + N)
    E, T, nb, blk = 6, L + 9, 3, 1
    bank = torch.randn((E, T, nb, D), generator=g)
    ep = torch.randint(0, E, (N,), generator=g)
    win = (torch.randint(0, T - L + 1, (N,), generator=g)[:, None] + torch.arange(L)[None, :]).long()
    pidx = (torch.randint(0, T - L + 1, (N,), generator=g)[:, None] + torch.arange(L)[None, :]).long()  # != win (Q5)
    cnt = torch.randint(0, L + 1, (N,), generator=g)
    cnt[0] = 0
    mask = torch.arange(L)[None, :] < cnt[:, None]
    table = torch.randn((T, D), generator=g) * 0.5
    wk = (torch.randn((D, D), generator=g) / D ** 0.5).requires_grad_(True)
    wv = (torch.randn((D, D), generator=g) / D ** 0.5).requires_grad_(True)
    lg = (1 + 0.1 * torch.randn((D,), generator=g)).requires_grad_(True)
    lb = (0.1 * torch.randn((D,), generator=g)).requires_grad_(True)
    pt = table.clone().requires_grad_(True)
    q = torch.randn((N, D), generator=g).requires_grad_(True)
    gout = torch.randn((N, D), generator=g)
    # oracle: explicit gather -> (+pos) -> (LN) -> projections -> attention (ref_model.mha without q-proj / fc_out)
    x = bank[ep][torch.arange(N)[:, None], win][:, :, blk]
    if pos:
        x = x + pt[pidx]
    if ln:
        x = torch.nn.functional.layer_norm(x, (D,), lg, lb, 1e-5)
    # the attention itself is the oracle's restatement of transformer.py:31-86 (oracle.ref_model.mha); the test hands over already
    # projected queries and takes the context before fc_out, so those two maps are the identity here
    eye = torch.eye(D)
    sd = {"a.values.weight": wv, "a.keys.weight": wk, "a.queries.weight": eye, "a.fc_out.weight": eye, "a.fc_out.bias": torch.zeros(D)}
    ctx3, a4 = rm.mha(sd, "a", H, x, x, q.unsqueeze(1), mask)
    ctx, a = ctx3[:, 0], a4[:, :, 0]
    (ctx * gout).sum().backward()
    ref = dict(ctx=ctx.detach(), att=a.detach(), q=q.grad, wk=wk.grad, wv=wv.grad, lg=lg.grad, lb=lb.grad, pt=pt.grad)

    d = lambda t: t.detach().clone().to(dev)
    qd, wkd, wvd = d(q).requires_grad_(True), d(wk).requires_grad_(True), d(wv).requires_grad_(True)
    lgd, lbd, ptd = d(lg).requires_grad_(True), d(lb).requires_grad_(True), d(pt).requires_grad_(True)
    spec = ops.WindowSpec.from_bank(bank.to(dev), ep.to(dev), win.to(dev), pidx.to(dev) if pos else None, mask.to(dev))
    out, att = ops.mha(qd, wkd, wvd, spec, blk, H, lgd if ln else None, lbd if ln else None, ptd if pos else None)
    close(out, ref["ctx"], what="ctx")
    close(att, ref["att"], atol=2e-6, what="att")
    assert torch.allclose(att[0].cpu(), torch.full((H, L), 1.0 / L), atol=1e-6)  # fully masked row -> uniform (Q2)
    (out * gout.to(dev)).sum().backward()

    def gclose(got, want, name, rel=2e-4):
        tol = rel * float(want.double().norm()) + 1e-7
        close(got, want.numpy(), atol=tol, rtol=1e-3, what=name)
    gclose(qd.grad, ref["q"], "dq")
    gclose(wkd.grad, ref["wk"], "dwk")
    gclose(wvd.grad, ref["wv"], "dwv")
    if ln:
        gclose(lgd.grad, ref["lg"], "dln_g", 5e-4)
        gclose(lbd.grad, ref["lb"], "dln_b", 5e-4)
    if pos:
        gclose(ptd.grad, ref["pt"], "dpos", 5e-4)


def test_mha_full_size_vs_torch_on_device(attn_impl):
    """BASELINE config (3) training shape (N=2048, L=64, D=384, H=4, 3 blocks in the bank) against plain torch ops
    on the same device, plus size-independent properties (rows of the attention sum to 1, masked slots are 0)."""
    from etm import ops
    dev = _dev()
    torch.manual_seed(0)
    N, L, D, H, T, nb, E = 2048, 64, 384, 4, 96, 3, 416
    bank = torch.randn((E, T, nb, D), device=dev)
    ep = torch.randint(0, E, (N,), device=dev)
    win = torch.randint(0, T - L + 1, (N, 1), device=dev) + torch.arange(L, device=dev)[None, :]
    cnt = torch.randint(0, L, (N,), device=dev)
    mask = torch.arange(L, device=dev)[None, :] < cnt[:, None]
    pos = torch.randn((T, D), device=dev) * 0.5
    wk = (torch.randn((D, D), device=dev) / D ** 0.5).requires_grad_(True)
    wv = (torch.randn((D, D), device=dev) / D ** 0.5).requires_grad_(True)
    q = torch.randn((N, D), device=dev).requires_grad_(True)
    gout = torch.randn((N, D), device=dev)
    spec = ops.WindowSpec.from_bank(bank, ep, win, win, mask)
    out, att = ops.mha(q, wk, wv, spec, 2, H, pos=pos)
    (out * gout).sum().backward()
    got = [t.detach().clone() for t in (out, att, q.grad, wk.grad, wv.grad)]
    q.grad = wk.grad = wv.grad = None
    torch.backends.cuda.matmul.allow_tf32 = False
    x = bank[ep[:, None], win][:, :, 2] + pos[win]
    hd = D // H
    k = (x @ wk.t()).reshape(N, L, H, hd)
    v = (x @ wv.t()).reshape(N, L, H, hd)
    e = torch.einsum("nhd,nkhd->nhk", q.reshape(N, H, hd), k).masked_fill(mask[:, None, :] == 0, float("-1e20"))
    a = torch.softmax(e / (D ** 0.5), dim=2)
    ctx = torch.einsum("nhl,nlhd->nhd", a, v).reshape(N, D)
    (ctx * gout).sum().backward()
    want = [ctx, a, q.grad, wk.grad, wv.grad]
    for name, g_, w_ in zip(("ctx", "att", "dq", "dwk", "dwv"), got, want):
        tol = 2e-4 * float(w_.double().norm()) / max(1.0, float(w_.numel()) ** 0.5) + 2e-5
        err = (g_ - w_).abs().max().item()
        assert err <= tol * 50, f"{name}: max err {err:.3e} vs tol {tol * 50:.3e}"
        assert (g_ - w_).double().norm() <= 2e-4 * w_.double().norm() + 1e-6, name
    assert torch.allclose(got[1].sum(-1), torch.ones((N, H), device=dev), atol=1e-5)
    nonfull = cnt > 0
    assert (got[1][nonfull] * (~mask[nonfull])[:, None, :]).abs().max() == 0


# ------------------------------------------------------------------ kernel #2
def test_gae_bit_exact(golden_dir):
    from etm import ops
    from oracle import ref_algo as ra
    dev = _dev()
    z = load(golden_dir, "gae.npz")
    for case in sorted({k.split("/")[0] for k in z.files}):
        t = lambda k: torch.from_numpy(z[f"{case}/{k}"]).to(dev)
        adv = ops.gae(t("rewards"), t("dones"), t("values"), t("last_value"), float(z[case + "/gamma"]), float(z[case + "/lamda"]))
        assert np.array_equal(adv.cpu().numpy(), z[case + "/adv"]), case
    rng = np.random.default_rng(3)
    for W, S in ((256, 512), (1000, 77), (1, 1), (65, 33)):
        r = rng.normal(size=(W, S)).astype(np.float32)
        d = rng.random((W, S)) < 0.03
        v = rng.normal(size=(W, S)).astype(np.float32)
        lv = rng.normal(size=(W,)).astype(np.float32)
        adv = ops.gae(torch.from_numpy(r).to(dev), torch.from_numpy(d).to(dev), torch.from_numpy(v).to(dev), torch.from_numpy(lv).to(dev), 0.995, 0.95)
        assert np.array_equal(adv.cpu().numpy(), ra.gae(r, d, v, lv, 0.995, 0.95).numpy()), (W, S)


# ------------------------------------------------------------------ kernel #3
def test_ppo_loss_vs_reference(golden_dir):
    from etm import ops
    dev = _dev()
    z = load(golden_dir, "loss.npz")
    for case in sorted({k.split("/")[0] for k in z.files}):
        t = lambda k: torch.from_numpy(z[f"{case}/{k}"]).to(dev)
        logits = t("logits").requires_grad_(True)
        value = t("value").requires_grad_(True)
        clip, cv, beta = (float(x) for x in z[f"{case}/hp"])
        loss, stats = ops.ppo_loss([logits], value, t("actions"), t("old_logp"), t("adv"), t("old_v"), clip, cv, beta)
        loss.backward()
        close(stats, z[f"{case}/stats"], atol=2e-6, rtol=1e-4, what=case + " stats")
        assert abs(loss.item() - float(z[f"{case}/stats"][2])) < 1e-5
        close(logits.grad, z[f"{case}/glogits"], atol=1e-8, rtol=2e-3, what=case + " glogits")
        close(value.grad, z[f"{case}/gvalue"], atol=1e-8, rtol=2e-3, what=case + " gvalue")


def test_ppo_loss_multi_branch_vs_oracle():
    from etm import ops
    from oracle import ref_algo as ra
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    N = 300
    lg = [torch.randn((N, a), generator=g).requires_grad_(True) for a in (3, 5)]
    value = torch.randn((N,), generator=g).requires_grad_(True)
    actions = torch.stack([torch.randint(0, 3, (N,), generator=g), torch.randint(0, 5, (N,), generator=g)], 1)
    old = torch.randn((N, 2), generator=g) * 0.1 - 1.2
    adv, oldv = torch.randn((N,), generator=g), torch.randn((N,), generator=g)
    loss, stats = ra.ppo_loss(lg, value, actions, old, adv, oldv, 0.2, 0.3, 0.01)
    loss.backward()
    lgd = [t.detach().to(dev).requires_grad_(True) for t in lg]
    vd = value.detach().to(dev).requires_grad_(True)
    l2, s2 = ops.ppo_loss(lgd, vd, actions.to(dev), old.to(dev), adv.to(dev), oldv.to(dev), 0.2, 0.3, 0.01)
    l2.backward()
    close(s2, stats.numpy(), atol=2e-6, rtol=1e-4, what="stats")
    for a, b in zip(lgd, lg):
        close(a.grad, b.grad.numpy(), atol=1e-8, rtol=2e-3, what="glogits")
    close(vd.grad, value.grad.numpy(), atol=1e-8, rtol=2e-3, what="gvalue")


def test_adv_stats_many_workgroups_vs_float64():
    """etm_adv_stats_ws (round 6): at N >= 65536 the (count, mean, M2) of the advantages come from per-chunk statistics on many
    workgroups + a fixed-order merge (the one-workgroup kernel read 2^24 samples at 6 GB/s); against float64, for ragged sizes, twice
    (deterministic); below the threshold `ops.adv_stats` is the old kernel, bit for bit."""
    from etm import lib as etm_lib
    from etm import ops
    dev = _dev()
    lib = etm_lib.load()
    g = torch.Generator().manual_seed(4)
    for N in (65536, 65537, 100003, 1 << 20, (1 << 24) + 12345):
        adv = (torch.randn(N, generator=g) * 3.0 + 0.7).to(dev)
        assert lib.etm_adv_stats_workspace_bytes(N) > 0
        a, b = ops.adv_stats(adv), ops.adv_stats(adv)
        assert torch.equal(a, b)
        ref = adv.double().cpu()
        mean, m2 = float(ref.mean()), float(((ref - ref.mean()) ** 2).sum())
        got = a.double().cpu().tolist()
        assert got[0] == float(np.float32(N)) and abs(got[1] - mean) <= 2e-6 * max(1.0, abs(mean)) and abs(got[2] - m2) <= 3e-6 * m2, (N, got, mean, m2)
    for N in (7, 2048, 65535):
        adv = torch.randn(N, generator=g).to(dev)
        assert lib.etm_adv_stats_workspace_bytes(N) == 0
        old = torch.empty(3, device=dev)
        etm_lib.check(lib.etm_adv_stats(adv.data_ptr(), N, old.data_ptr(), torch.cuda.current_stream().cuda_stream), "etm_adv_stats")
        assert torch.equal(ops.adv_stats(adv), old)


def test_ppo_loss_vectorised_path_vs_oracle():
    """Large-batch form of the loss kernel (four samples per thread, 16-byte operand moves; taken from 65,536 samples with three
    actions) against the oracle's restatement of trainer.py:276-304, including ties of the clipping rules (exact 1.0 ratios)."""
    from etm import ops
    from oracle import ref_algo as ra
    dev = _dev()
    g = torch.Generator().manual_seed(14)
    N = 1 << 16
    lg = [torch.randn((N, 3), generator=g).requires_grad_(True)]
    value = torch.randn((N,), generator=g).requires_grad_(True)
    actions = torch.randint(0, 3, (N, 1), generator=g)
    with torch.no_grad():
        lsm = torch.log_softmax(lg[0], dim=1).gather(1, actions)
    old = lsm + torch.randn((N, 1), generator=g) * 0.2
    old[::7] = lsm[::7]                                        # ratio exactly 1: the tie branches of min / clamp
    adv, oldv = torch.randn((N,), generator=g), torch.randn((N,), generator=g)
    loss, stats = ra.ppo_loss(lg, value, actions, old, adv, oldv, 0.2, 0.3, 0.01)
    loss.backward()
    lgd = [lg[0].detach().to(dev).requires_grad_(True)]
    vd = value.detach().to(dev).requires_grad_(True)
    l2, s2 = ops.ppo_loss(lgd, vd, actions.to(dev), old.to(dev), adv.to(dev), oldv.to(dev), 0.2, 0.3, 0.01)
    l2.backward()
    close(s2, stats.numpy(), atol=2e-6, rtol=2e-4, what="stats")
    close(lgd[0].grad, lg[0].grad.numpy(), atol=1e-10, rtol=2e-3, what="glogits")
    close(vd.grad, value.grad.numpy(), atol=1e-10, rtol=2e-3, what="gvalue")


# ------------------------------------------------------------------ whole path: teacher-forced rollout + updates vs the reference
# Every rollout path can be teacher-forced (the sampling kernels read recorded actions from a fixed-address table), so the
# reference fixtures pin the EXACT configuration bench.py times -- captured head/tail graphs, observation streaming, two
# pipelined worker groups of 16 (case img32, default keys) -- as well as the eager twin and the optional host paths.
_ROLLOUT_PATHS = {
    "default": {},                                              # graphs (+ streaming / worker groups where the shape allows)
    "eager": {"hip_graph_rollout": False},
    "graph_one_group": {"rollout_groups": 1},
    "graph_unstreamed": {"stream_observations": False},
    "groups4": {"rollout_groups": 4},
    "event_handover": {"host_flag_actions": False},          # head graph, event, tail graph instead of one graph + pinned flag
    "eager_train": {"hip_graph_train": False},
    "library_convs": {"fused_train_encoder": False},          # optimisation phase on the library convolutions
    "multi_launch_blocks": {"fused_rollout_block": False},    # rollout: one launch per GEMM / attention / LayerNorm instead of one per step
    "launched_tail": {"fused_rollout_tail": False},           # bank write + K/V projection of the new items as separate launches
    "uploaded_rows": {"direct_observation_rows": False},      # observation rows through pinned memory + a copy-engine transfer (rounds 1 - 5) instead of host writes into device memory
    "four_groups": {"rollout_groups": 4, "rollout_min_group_size": 2},          # four worker groups (correct with any number of hardware queues)
    "separate_heads": {"fused_heads_loss": False, "grouped_dw_train": False,    # heads / loss / weight gradients as separate ops (round-2 form),
                       "grouped_colsum_train": False},                          # every column-sum gradient reduced by its own launch
    # round 4: environments in worker PROCESSES over a shared, HIP-registered segment; the per-step host loop is the library's native
    # driver (etm_rollout_drive) where the step is a flag-hand-over graph with streamed observations, else the host-driven protocol
    "kslice_hidden": {"fused_conv3_hidden": False},              # lin_hidden of a rollout step as 16 K-slice sums behind the third convolution (round 3)
    # round 5 (pre-LN models): norm_kv's statistics gathered from per-bank-row statistics taken once per update (opt-in) instead of per
    # window row inside etm_window_fwd; norm_kv's gain / bias gradients through the generic dX kernel instead of csrc/window_ln_grad.hip
    "window_row_stats": {"bank_row_stats": False},      # norm_kv statistics per window row inside the passes (default: once per bank row)
    "generic_ln_grad": {"fused_ln_grad": False},
    "rows_ln_grad": {"fused_ln_grad": "rows"},           # norm_kv's gradients by round 5's pass over the window rows (default since round 6: from the passes' outputs)
    "fp32_encoder": {"encoder_products": "fp32"},        # round 6: the fp32-MFMA encoder kernels in the optimisation phase (default: the bf16 matrix pipe at fp32 accuracy, csrc/conv_b3*.hip)
    "worker_processes": {"worker_processes": True},
    "worker_processes_k4": {"worker_processes": True, "envs_per_process": 4, "rollout_groups": 4, "rollout_min_group_size": 2},
    "worker_processes_eager": {"worker_processes": True, "envs_per_process": 2, "hip_graph_rollout": False},
}
_TF_CASES = [("vec", "default"), ("vec", "eager"), ("gtrxl", "default"), ("gtrxl", "eager"), ("img", "default"), ("img", "eager"),
             ("img32", "default"), ("img32", "eager"), ("img32", "graph_one_group"), ("img32", "graph_unstreamed"),
             ("img32", "groups4"), ("img32", "event_handover"), ("img32", "eager_train"), ("img32", "library_convs"),
             ("img32", "multi_launch_blocks"), ("vec", "multi_launch_blocks"), ("img32", "launched_tail"), ("vec", "launched_tail"),
             # BASELINE model sizes (round 3): the kernel instantiations bench.py / tools/config_bench.py time, pinned to the reference
             ("cfg2", "default"), ("cfg2", "eager"), ("cfg3", "default"), ("cfg3", "eager"), ("cfg3", "multi_launch_blocks"),
             ("cfg5", "default"), ("cfg5", "eager"),
             ("img32", "separate_heads"), ("cfg3", "separate_heads"), ("img32", "four_groups"), ("cfg3", "four_groups"), ("img32", "uploaded_rows"), ("cfg3", "uploaded_rows"), ("cfg5", "uploaded_rows"),
             ("img32", "worker_processes"), ("img32", "worker_processes_k4"), ("cfg3", "worker_processes"), ("cfg3", "worker_processes_k4"),
             ("vec", "worker_processes"), ("img32", "worker_processes_eager"), ("cfg5", "worker_processes"),
             ("img32", "kslice_hidden"), ("cfg3", "kslice_hidden"), ("cfg5", "kslice_hidden"), ("cfg2", "window_row_stats"), ("cfg2", "generic_ln_grad"), ("cfg5", "rows_ln_grad"), ("gtrxl", "rows_ln_grad"), ("cfg5", "window_row_stats"),
             ("cfg3", "fp32_encoder"), ("cfg5", "fp32_encoder")]


def movement_error(sd, z, tag, keys, prev):
    """Relative error of the parameter MOVEMENT of one update on the fixture's sampled elements: ||got - ref|| / ||ref - before||
    per tensor and over all tensors.  (AdamW's first steps move every element by ~lr * sign(g): an absolute tolerance on the
    parameters hides errors as large as the movement itself; this one does not.)"""
    worst, worst_key, num, den = 0.0, "", 0.0, 0.0
    for k in keys:
        if k.endswith("inv_freqs"):
            continue
        got = dg.sample(sd[k].detach().cpu().numpy(), 64).astype(np.float64)
        ref = z[tag + "sd_after_sample/" + k].astype(np.float64)
        mv, err = ref - prev[k], got - ref
        n_mv = float(np.linalg.norm(mv))
        if n_mv > 0 and float(np.linalg.norm(err)) / n_mv > worst:
            worst, worst_key = float(np.linalg.norm(err)) / n_mv, f"{k} (movement norm {n_mv:.2e} over {mv.size} sampled elements)"
        num, den = num + float(np.sum(err ** 2)), den + float(np.sum(mv ** 2))
        prev[k] = ref
    return worst, worst_key, (num / max(den, 1e-300)) ** 0.5


def test_column_sums_of_the_captured_step_are_the_librarys_own():
    """Round 5: the one torch reduction of the captured optimisation step (the norm_kv gradient pass's fallback column sum) returned
    wrong sums in some replays -- torch's two-stage reduction (memset node + semaphore kernel) under this runtime's graph launch,
    tools/graph_reduce_hazard.py / profiles/r05/graph_reduce_hazard.txt.  The fallback is the library's fixed-order reduction now:
    bit-identical to the grouped launch of a collector, and exact in every replay of a captured graph."""
    from etm import ops
    dev = _dev()
    torch.manual_seed(3)
    rows, C = 1024, 256
    src = torch.randn(16, rows, C + 64, device=dev)
    partial = torch.empty(rows, C + 64, device=dev)
    idx = torch.zeros((), dtype=torch.long, device=dev)
    out = torch.empty(C, device=dev)

    def body():
        partial.copy_(src.index_select(0, idx.view(1))[0])
        out.copy_(ops.colsum_rows(partial, rows, C))          # (first C columns of rows with stride C + 64)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    eager = []
    for i in range(16):
        idx.fill_(i)
        body()
        eager.append(out.clone())
        ref = src[i, :, :C].double().sum(dim=0)
        assert float((out.double() - ref).abs().max()) < 1e-3
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for it in range(600):
        idx.fill_(it % 16)
        g.replay()
        assert torch.equal(out, eager[it % 16]), f"replay {it}: the captured reduction differs from the eager one"
    # and no torch reduction is left in the autograd functions of the step (a reminder for whoever adds one)
    import inspect
    assert ".sum(" not in inspect.getsource(ops._WindowFn.backward)


@pytest.mark.parametrize("name,path", _TF_CASES)
def test_trainer_teacher_forced_vs_reference(golden_dir, name, path):
    from trainer import PPOTrainer
    dev = _dev()
    z = load(golden_dir, f"rollout_{name}.npz")
    info = json.loads(str(z["cfg_json"]))
    cfg, envk = info["cfg"], info["env"]
    cfg = {**cfg, **_ROLLOUT_PATHS[path], "environment": {"type": "Synthetic", **envk}}     # the trainer builds (and groups) the env
    tr = PPOTrainer(cfg, run_id="parity", device=dev, tensorboard=False)
    keys, shapes = shapes_of(z, "")
    load_det(tr.model, "rollout_" + name, keys, shapes)
    prev = {k: dg.sample(v, 64).astype(np.float64) for k, v in dg.det_state_dict("rollout_" + name, keys, shapes).items()}
    W, S = cfg["n_workers"], cfg["worker_steps"]
    for upd in range(cfg["updates"]):
        tag = f"u{upd}/"
        tr._sample_training_data(forced_actions=z[tag + "actions"][:, :, 0])
        tr.buffer.prepare_batch_dict()
        b = tr.buffer
        assert np.array_equal(b.actions.cpu().numpy(), z[tag + "actions"])
        # memory-window bookkeeping: bit exact
        assert np.array_equal(b.memory_mask.cpu().numpy(), z[tag + "memory_mask"])
        assert np.array_equal(b.memory_indices.cpu().numpy(), z[tag + "memory_indices"])
        assert np.array_equal(b.memory_index.cpu().numpy(), z[tag + "memory_index"])
        assert np.array_equal(b.dones, z[tag + "dones"]) and np.array_equal(b.rewards, z[tag + "rewards"])
        assert np.array_equal(tr.worker_current_episode_step, z[tag + "ep_step_after"])
        if tag + "obs" in z:
            assert np.array_equal(b.obs.cpu().numpy(), z[tag + "obs"])
        else:      # image observations: the fixture holds a subsample and the sum
            ob = b.obs.cpu().numpy()
            assert np.array_equal(dg.sample(ob, 8192), z[tag + "obs_sample"]) and np.float64(ob.astype(np.float64).sum()) == z[tag + "obs_sum"]
        # error scaled by max(1, |reference|): values / advantages of the cfg2 fixture are O(100) (a reward every step)
        measured = {f: float((np.abs(getattr(b, f).cpu().numpy().astype(np.float64) - z[tag + f]) / np.maximum(1.0, np.abs(z[tag + f]))).max())
                    for f in ("values", "log_probs", "advantages")}
        if os.environ.get("ETM_TF_MEASURE_LOG"):
            with open(os.environ["ETM_TF_MEASURE_LOG"], "a") as f:
                f.write(json.dumps({"case": name, "path": path, "update": upd, "stage": "rollout", **measured}) + "\n")
        bound = tf_bounds(name, upd)
        for f in ("values", "log_probs", "advantages"):
            assert measured[f] <= bound["forward"], f"{name}/{path} update {upd}: {f} off by {measured[f]:.2e} (scaled by max(1, |ref|)); bound {bound['forward']:.0e}"
        if tag + "memories" in z:
            e_ref = z[tag + "memories"].shape[0]
            assert b.num_episodes >= e_ref
            close(b.memories[:e_ref], z[tag + "memories"], atol=1e-4, what="memories")
        else:      # many episodes: subsample + sum of the reference's stacked episode list
            e_ref = int(z[tag + "memories_shape"][0])
            assert b.num_episodes >= e_ref
            mem = b.memories[:e_ref].cpu().numpy()
            assert tuple(mem.shape) == tuple(z[tag + "memories_shape"])
            close(dg.sample(mem, 32768), z[tag + "memories_sample"], atol=1e-4, what="memories")
            assert abs(float(mem.astype(np.float64).sum()) - float(z[tag + "memories_sum"])) < 1e-4 * mem.size ** 0.5 + 1e-3
        lr, clip, beta = (float(x) for x in z[tag + "hp"])
        assert (lr, beta, clip) == tuple(float(x) for x in tr.schedules(upd))
        # the un-clipped gradient of the update's first minibatch (what loss.backward() leaves behind, trainer.py:310) against the
        # reference's: error of the sampled elements relative to the tensor's norm (worst tensor) and over all tensors
        mbs = (W * S) // cfg["n_mini_batch"]
        grads = tr.minibatch_gradients(z[tag + "perms"][0][:mbs], clip, beta)
        num = den = g_worst = 0.0
        g_worst_key = ""
        for k, g in grads.items():
            ref, got = z[tag + "grad0_sample/" + k].astype(np.float64), dg.sample(g.cpu().numpy(), 64).astype(np.float64)
            scale = float(z[tag + "grad0_norm/" + k]) * (ref.size / max(1, g.numel())) ** 0.5      # norm of a 64-element sample of this tensor
            e = float(np.linalg.norm(got - ref))
            if scale > 0 and e / scale > g_worst:
                g_worst, g_worst_key = e / scale, k
            num, den = num + e * e, den + float(np.sum(ref ** 2))
        g_all = (num / max(den, 1e-300)) ** 0.5
        print(f"[teacher-forced {name}/{path} update {upd}] gradient error: all tensors {g_all:.2e}, worst tensor {g_worst:.2e} = {g_worst_key}")
        stats, _ = tr._train_epochs(lr, clip, beta, perms=z[tag + "perms"])
        st_err = np.abs(np.asarray(stats, dtype=np.float64) - z[tag + "stats"]) / np.maximum(1.0, np.abs(z[tag + "stats"]))
        measured["stats"] = float(st_err.max())
        if os.environ.get("ETM_TF_DEBUG"):
            print("STATS-DEBUG", name, path, upd, [f"{e:.1e}" for e in st_err.max(axis=1)], "\n got", np.asarray(stats)[int(st_err.max(axis=1).argmax())],
                  "\n ref", z[tag + "stats"][int(st_err.max(axis=1).argmax())], flush=True)
        assert measured["stats"] <= bound["stats"], f"{name}/{path} update {upd}: loss statistics off by {measured['stats']:.2e}; bound {bound['stats']:.0e}"
        worst, worst_key, overall = movement_error(tr.model.state_dict(), z, tag, keys, prev)
        measured.update(move_all=overall, move_worst=worst, grad_all=g_all, grad_worst=g_worst)
        assert g_all <= bound["grad_all"] and g_worst <= bound["grad_tensor"], (g_all, g_worst, g_worst_key)
        ratchet_violations = tf_ratchet(name, path, upd, measured)
        print(f"[teacher-forced {name}/{path} update {upd}] parameter-movement error: all tensors {overall:.2e}, worst tensor {worst:.2e} = {worst_key}")
        print(f"[teacher-forced {name}/{path} update {upd}] measured: " + ", ".join(f"{k} {v:.2e}" for k, v in measured.items()))
        if os.environ.get("ETM_TF_MEASURE_LOG"):
            with open(os.environ["ETM_TF_MEASURE_LOG"], "a") as f:
                f.write(json.dumps({"case": name, "path": path, "update": upd, **measured}) + "\n")
        assert overall <= bound["move_all"] and worst <= bound["move_tensor"], (worst, worst_key, overall)
        assert not ratchet_violations, f"{name}/{path} update {upd}: " + "; ".join(ratchet_violations)
    if name == "img32" and path == "default":
        assert tr._step_graph is not None and tr._stream_obs and len(tr._groups) == 2 and tr._train_graph is not None, \
            "img32/default must run the benchmarked configuration: graphs, observation streaming, two worker groups"
        assert tr.model._train_encoder_ok, "img32/default: the optimisation phase runs the hand-written encoder kernels"
        assert tr.model._rf is not None, "img32/default: transformer + heads + sampling of a rollout step are one launch"
    if name in ("cfg2", "cfg3", "cfg5") and path == "default":
        # the benchmarked instantiations ran: captured step graphs with the one-launch step kernel (teams of etm_rollout_trxl_team(H)
        # workgroups per worker: 4 at H = 4), captured optimisation step; visual configs: observation streaming, two worker groups,
        # the hand-written encoder kernels
        from etm import lib as etm_lib
        H = cfg["transformer"]["num_heads"]
        assert tr._step_graph is not None and tr._train_graph is not None and tr.model._rf is not None, name
        assert all(g.rf_scratch is not None and g.tail_in_kernel for g in tr._groups), "every group's step ran etm_rollout_trxl incl. its tail"
        assert etm_lib.load().etm_rollout_trxl_team(H) == (4 if H == 4 else 1)
        # round 5: the gated layouts (cfg2, cfg5) run the GROUP form of the step kernel (csrc/rollout_group.hip: groups of <= 8 workers)
        assert all(bool(g.group_kernel) == (name in ("cfg2", "cfg5")) for g in tr._groups), [(g.W, g.group_kernel) for g in tr._groups]
        assert tr.model._rf["pre_ln"] == int(cfg["transformer"]["layer_norm"] == "pre") and tr.model._rf["gtrxl"] == int(cfg["transformer"]["gtrxl"])
        if name != "cfg2":
            assert tr._stream_obs and tr._host_flag and len(tr._groups) == 2 and tr.model._train_encoder_ok, name
    if path.startswith("worker_processes"):
        # (environments per process: the configured number, or more where the rank's CPU share does not cover that many spinning
        # processes -- etm/hostcpu.py; the 1-GPU boxes of this pool run under a 16-CPU quota -- bookkeeping is independent of it)
        assert tr._shm_env is not None and tr._shm_env.envs_per_proc == min(tr._host_plan["envs_per_process"], tr.num_workers // len(tr._groups))
        assert tr._host_plan["envs_per_process"] >= cfg.get("envs_per_process", 1)
        # visual observations + graphs: the native driver ran the per-step loop; vector observations / eager: the host-driven protocol
        assert bool(getattr(tr, "_native_rollout", False)) == (name != "vec" and path != "worker_processes_eager"), (name, path)
        if path == "worker_processes_k4":
            assert len(tr._groups) == 4
    assert tr.buffer.block_major and tr.buffer.bank.stride(2) > tr.buffer.bank.stride(0)   # [blocks][slots][T][D] in memory
    if path == "groups4":
        assert len(tr._groups) == 4
    if name in ("img32", "cfg3", "cfg5") and path in ("default", "four_groups", "uploaded_rows"):
        from etm import ops as etm_ops
        # round 6: the in-process front-ends write their rows straight into the staging array in device memory where the host can
        # (large BAR); the bit-exact observation comparison above is what checks every one of those writes
        assert tr._direct_rows == (path != "uploaded_rows" and etm_ops.host_direct_write_ok(dev)), (name, path, tr._direct_rows)
    if path == "event_handover":
        assert not tr._host_flag
    tr.close()


# ---- Tolerances of the teacher-forced whole-path comparison.  Measured maxima on the MI355X over all fixtures and paths (round 3,
# profiles/r03/teacher_forced_measured.jsonl); the bounds are 2 - 5 x those.
#
#   forward (values, log-probs, advantages; |got - ref| / max(1, |ref|)):  <= 1.0e-6 on the small fixtures in every update,
#       <= 2.6e-6 at the BASELINE model sizes while the parameters are still the fixture's (update 0); SURVEY 8c asks for 1e-5.
#       After an optimiser phase the two parameter sets differ by the optimiser's own amplification of rounding noise (below), so
#       later updates of the BASELINE-size fixtures are bounded at 1e-4 (measured <= 2.9e-5, cfg2 after 8 AdamW steps).
#   loss statistics (same scaling): <= 5e-7 small, <= 1.3e-5 BASELINE sizes.
#   gradient of the update's first minibatch against the reference's (un-clipped, trainer.py:310; error of the sampled elements
#       over the norm): all tensors <= 3.2e-7 small / <= 7.2e-6 BASELINE sizes; worst single tensor 6e-6 / 1.0e-3 (cfg2's
#       lin_hidden.weight: ONE flipped ReLU unit of the layer above it, see below; every other tensor <= 7.7e-5).  SURVEY 8c: 1e-4.
#   post-update parameters, stated on the MOVEMENT of an update (||got - ref|| / ||ref - before|| on the sampled elements; an
#       absolute tolerance on parameters would hide errors as large as the movement itself): all tensors <= 7.6e-5, worst tensor
#       5.2e-4 on the small fixtures (the optimiser kernel follows the reference's single-tensor AdamW operation by operation,
#       csrc/optim.hip).  At the BASELINE sizes the FULL-minibatch trajectory cannot be held to a floor: a minibatch there holds
#       1e7 - 1e8 ReLU inputs, a few of them within fp32 rounding of zero, and two correct fp32 evaluations may put such a unit on
#       different sides of the kink.  Round 4 traced the whole round-3 excess to exactly that (profiles/r04/parity_decomposition.md):
#       cfg2 -- unit 107 of linear_embedding for one observation, pre-activation 5.3e-8; cfg3 -- one unit of lin_value; the error of
#       every affected tensor is rank 1 with a one-hot left factor, and the REFERENCE shows the same against its own float64
#       evaluation (cfg3 u0 step 1: lin_policy 2.1e-4; cfg2 u1 step 2: gate1.Wr 1.7e-3).  One flipped dense unit moves the
#       tensors below it by 1e-5 .. 1e-3 of their norm, and AdamW (update lr * m_hat / (sqrt(v_hat) + eps): scale-free) turns that
#       into sign changes of every element whose gradient is smaller -- measured 8.2e-4 / 4.6e-3 (cfg3), 1.4e-4 / 1.3e-3 (cfg2) in
#       the first update where the flip-free paths of cfg5 and img32 sit at 4e-5 .. 8e-5.  cfg5 has such a unit too: lin_policy unit 162
#       of the sample at sorted position 1672, float64 pre-activation within 3.3e-7 of zero under EVERY rollout path's memory items, so
#       which paths evaluate it as active is an accident of rounding and moves whenever a summation order upstream changes.  END STATE
#       OF ROUND 5 (tests/golden/tf_measured_baseline.json, known_flips[2]; profiles/r05/parity_pair_cfg5_final.txt): after the group
#       step kernel folded fc_out into the first gate's maps the pre-activation under the DEFAULT rollout went from +9.7e-8 to -6.2e-9,
#       i.e. cfg5/default is QUIET (4.2e-5 / 1.6e-4, like eager and worker_processes) and the flip shows on kslice_hidden and
#       window_row_stats (1.4e-3 / 5.4e-3; pre-activations +2.4e-7 / -8.4e-9).  (Mid round 5 it was the other way round -- default
#       1.37e-3 / 5.4e-3 with the other paths quiet, profiles/r05/parity_pair_cfg5.txt: same unit, same signature.)  lin_policy.bias
#       differs in element 162 only, lin_policy.weight by a rank-1 term in that row; paths on the same side agree to 1.5e-7 .. 1.9e-7.
#       Bounds of this test at the BASELINE sizes therefore
#       stay at 2e-3 / 1e-2 (first update), 5e-3 / 2e-2 later; gradient of the first minibatch 2e-5 / 2e-3 of the norm (a flipped
#       unit: cfg2 lin_hidden.weight 1.0e-3).  The TIGHT statement -- per tensor and per optimiser step against the float64
#       evaluation, with bounds that are multiples of the floor of the same step -- is test_kink_free_update_vs_reference below, on
#       minibatches from which the near-kink samples are removed: gradients <= 3 x the reference's own error (measured 0.24 - 1.9 x:
#       1.3e-7 vs 5.4e-7 at cfg3), per tensor <= max(4 x, 2e-6), movement <= 3 x the twins' distance (measured 0.9 - 1.6 x).
#       And because these bounds cannot see a regression, every (case, path, update) is also held to 3 x its own recorded
#       measurement (tf_ratchet below).
# ---- The ratchet (round 5).  The bounds of tf_bounds at the BASELINE sizes are wide by necessity (a flipped ReLU unit is a
# legitimate outcome there) -- wide enough that an 18 x shift of cfg5/default went through green in round 4.  So next to them every
# (case, path, update) has its MEASURED values on record (tests/golden/tf_measured_baseline.json, written by tools/tf_ratchet.py from
# the measurement log of a GPU suite run); a value that exceeds ratio (3) x max(record, floor) fails.  Recording a louder value is
# only possible with a `known_flips` item that names the flipped unit, its pre-activation and the probe output
# (tools/parity_pair.py): cfg2 -- unit 107 of linear_embedding; cfg3 -- one unit of lin_value; cfg5 kslice_hidden /
# window_row_stats -- unit 162 of lin_policy (profiles/r05/parity_pair_cfg5_final.txt; cfg5/default, eager, worker_processes are quiet).
_RATCHET = None


def tf_ratchet(name, path, upd, measured):
    global _RATCHET
    if _RATCHET is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_measured_baseline.json")) as f:
            _RATCHET = json.load(f)
    if os.environ.get("ETM_TF_RATCHET", "1") == "0":      # re-measuring after a deliberate change (tools/tf_ratchet.py update <log>)
        return []
    entry = _RATCHET["entries"].get(f"{name}/{path}/{upd}")
    if entry is None:
        return [f"no ratchet entry for {name}/{path}/{upd}: run the GPU suite with ETM_TF_MEASURE_LOG=<log> ETM_TF_RATCHET=0 and "
                f"`python tools/tf_ratchet.py update <log>`"]
    out = []
    for f in ("move_all", "move_worst", "grad_all", "grad_worst"):
        limit = float(_RATCHET["ratio"]) * max(float(entry[f]), float(_RATCHET["floors"][f]))
        if entry.get("flip"):
            # round 6: an entry that carries a known flip is dominated by ONE fixed rank-1 term (the flipped unit) that every path
            # reproduces to four digits (profiles/r06: 8.221e-4 .. 8.225e-4 over nine cfg3 paths, the same on the boxes of two rounds),
            # so ratio x the record would admit a regression as large as the flip itself.  What is held instead is the flip-free
            # REMAINDER: an independent error e on top of the flip f shows as sqrt(f^2 + e^2), and e is bounded by ratio x the
            # flip-free level of the BASELINE-size fixtures (the quiet cfg5 paths; x later_update_factor once the trajectories
            # have parted).  cfg3 update 0: 8.22e-4 -> limit 8.32e-4 instead of 2.5e-3.
            ff = float(_RATCHET["flip_free"][f]) * (float(_RATCHET["flip_free"]["later_update_factor"]) if upd > 0 else 1.0)
            limit = 1.005 * (float(entry[f]) ** 2 + (float(_RATCHET["ratio"]) * max(ff, float(_RATCHET["floors"][f]))) ** 2) ** 0.5   # (0.5 %: the records' own scatter over boxes is 0.02 %)
        if measured[f] > limit:
            out.append(f"{f} {measured[f]:.3e} exceeds the limit {limit:.3e} from the recorded {entry[f]:.3e} (tests/golden/tf_measured_baseline.json"
                       f"{', known flip ' + entry['flip'] if entry.get('flip') else ''}): find the cause with tools/parity_pair.py {name} {path},eager "
                       f"before recording a new value")
    return out


def tf_bounds(name, upd):
    big = name.startswith("cfg")
    return {"forward": 1e-4 if (big and upd > 0) else 5e-6,
            "grad_all": 2e-4 if (big and upd > 0) else 2e-5,
            "grad_tensor": 1e-2 if (big and upd > 0) else 2e-3,
            "stats": 5e-5 if big else 5e-6,
            "move_all": (5e-3 if upd > 0 else 2e-3) if big else 1.5e-4,
            "move_tensor": (2e-2 if upd > 0 else 1e-2) if big else 1e-3}


# ---- Kink-free optimisation run (round 4).  Why it exists: at the BASELINE model sizes a minibatch holds 10^7 - 10^8 ReLU inputs, and
# a handful of them lie within fp32 rounding distance of zero.  Two CORRECT fp32 evaluations can put such a unit on different sides of
# the kink; one flipped dense-layer unit changes every gradient tensor below it by a rank-1 term of 1e-5 .. 1e-3 of the tensor's norm
# (profiles/r04/parity_decomposition.md: cfg2 -- unit 107 of linear_embedding, pre-activation 5.3e-8; cfg3 -- one unit of lin_value; the
# reference's own fp32 result shows the same against its float64 evaluation, lin_policy at cfg3 u0/s1: 2.1e-4), and AdamW turns that
# into parameter-movement differences that no accumulation order can remove.  The fixture generator therefore also records a short
# run on minibatches from which every sample with a ReLU input |x| < 1e-5 (float64 forward at the step's parameters) is removed, with
# two twins: the reference's fp32 step and the float64-gradient step (same fp32 clipping + AdamW).  On those minibatches no unit can
# flip, so the HIP path must agree with BOTH to accumulation noise, per tensor and per step -- and the bounds below are multiples of
# the floor measured in the same step (the twins' distance), not free constants.
_KF_GRAD_RATIO_ALL = 3.0       # HIP-vs-float64 gradient error over all tensors <= this x reference-vs-float64 (same step, same samples)
_KF_GRAD_RATIO_TENSOR = (4.0, 8.0)   # ... per tensor (64-element samples scatter more), or the absolute floor below: first step / later steps.  In
#                                      the later steps the two evaluations no longer share their parameters and -- at cfg2, whose values are O(100), so
#                                      that the value-loss gradient cancels to ~1e-5 of its terms -- BOTH get noisier by the step (reference 3.2e-7 ->
#                                      1.2e-6, HIP 1.6e-7 -> 3.0e-6 over three steps; cfg3 / cfg5 stay flat); per tensor the HIP / reference ratio
#                                      measured up to 4.6 there (lin_policy.bias, step 2: 2.7e-6 against 5.8e-7), first steps stay under 4
_KF_GRAD_ABS_TENSOR = 2e-6           # per-tensor error / tensor norm that is accepted whatever the reference's own error is.  In the later
#                                      steps the parameters differ from the fp32 twin's by the movement error of the earlier steps;
#                                      what that does to a tensor's gradient is MEASURED by the generator (round 5, `xgrad_twin_shift`:
#                                      the float64 gradient at the fp32 twin's parameters minus the float64 gradient at the
#                                      float64-gradient twin's, whole tensors) and enters the bound as _KF_MOVE_RATIO x that shift --
#                                      the HIP trainer's parameters may be _KF_MOVE_RATIO x the twins' distance away (asserted below).
#                                      It replaces round 4's free constant 1e-5 for the later steps.
#                                      Round 5: the first-step floor is back at 2e-6 (round 4 had widened it to 5e-6 for one gate matrix of
#                                      cfg5 at 2.7e-6 on its 64 samples).  The reference's own error of a tensor is now taken as the LARGER
#                                      of its 64-sample estimate and its WHOLE-tensor value (`xgrad_err / xgrad_norm`, recorded by the
#                                      generator): a 64-element sample of a 147,456-element matrix scatters by tens of per cent and the
#                                      "4 x the reference" arm of the bound inherited that scatter.
_KF_MOVE_RATIO = 3.0           # parameter movement error (vs either twin) <= this x the twins' own distance in the same step


@pytest.mark.parametrize("mode", ["api", "graph"])
@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg5"])
def test_kink_free_update_vs_reference(golden_dir, name, mode):
    """mode "api": the steps go through the upstream-API path (gathered minibatch -> `_train_mini_batch`, eager).  mode "graph"
    (round 5): through `_train_step_graph` with SORTED indices -- the captured HIP graph of the optimisation step that bench.py and
    `run_training` replay (two eager warm-up steps first, taken on a snapshot of the optimiser state that is restored before the
    compared steps; the graph is re-captured when the kink-free minibatch changes its size)."""
    from trainer import PPOTrainer
    dev = _dev()
    z = load(golden_dir, f"rollout_{name}.npz")
    info = json.loads(str(z["cfg_json"]))
    cfg, envk = info["cfg"], info["env"]
    cfg = {**cfg, "environment": {"type": "Synthetic", **envk}}
    tr = PPOTrainer(cfg, run_id="kinkfree", device=dev, tensorboard=False)
    keys, shapes = shapes_of(z, "")
    load_det(tr.model, "rollout_" + name, keys, shapes)
    tr._sample_training_data(forced_actions=z["u0/actions"][:, :, 0])
    tr.buffer.prepare_batch_dict()
    lr, clip, beta = (float(x) for x in z["u0/hp"])
    pnames = [str(k) for k in z["param_keys"]]
    params = dict(tr.model.named_parameters())
    assert list(params) == pnames
    init = {k: dg.sample(params[k].detach().cpu().numpy(), 64).astype(np.float64) for k in pnames}
    if mode == "graph":
        assert tr._use_train_graph and tr.config.get("sort_minibatch", True)
        opt = tr.optimizer
        arenas = (opt.flat_params, opt.exp_avg, opt.exp_avg_sq, opt.step_dev)
        snap = [t.clone() for t in arenas]
        idx0 = torch.as_tensor(z["kf/s0/idx"], device=dev, dtype=torch.long).sort().values
        with torch.no_grad():
            tr._bank_pos, tr._obs_train = tr._bank_with_positions(), tr._observations_channels_last()
        for _ in range(2):                                   # the two eager warm-up steps every capture is preceded by
            tr._train_step_graph(idx0, lr, clip, beta, False)
        assert tr._train_graph is None and tr._train_warm == 2
        with torch.no_grad():
            for t, s0 in zip(arenas, snap):
                t.copy_(s0)
        for k in pnames:
            assert np.array_equal(dg.sample(params[k].detach().cpu().numpy(), 64).astype(np.float64), init[k])

    def rows(key):
        a = z[key]
        return [a[i][~np.isnan(a[i])].astype(np.float64) for i in range(a.shape[0])]

    for s in range(int(z["kf/steps"])):
        st = f"kf/s{s}/"
        idx = z[st + "idx"]
        mbs_kf = (cfg["n_workers"] * cfg["worker_steps"]) // cfg["n_mini_batch"]
        # candidates: the update's first minibatch; round 6: one step on ANOTHER sample set (the second minibatch, or the second half of
        # the first where an epoch has one minibatch) and a revisit of the first (tests/golden/make_golden.py:_kink_free_run)
        assert int(z[st + "dropped"]) + idx.size in (mbs_kf, mbs_kf - mbs_kf // 2), (s, int(z[st + "dropped"]), idx.size)
        # ---- gradient: HIP vs the float64 evaluation, beside the reference's fp32 gradient vs the same
        grads = tr.minibatch_gradients(idx, clip, beta)
        xs, rs, xnorm, xerr = rows(st + "xgrad_samples"), rows(st + "grad_samples"), z[st + "xgrad_norm"], z[st + "xgrad_err"]
        xshift = z[st + "xgrad_twin_shift"]          # zero in step 0 (the twins start from the same parameters)
        num_h = num_r = den = 0.0
        worst = (0.0, "", 0.0)
        violations = []
        per_tensor = []
        for i, k in enumerate(pnames):
            got = dg.sample(grads[k].cpu().numpy(), 64).astype(np.float64)
            scale = float(xnorm[i]) * (xs[i].size / grads[k].numel()) ** 0.5          # norm of a sample of this size
            eh, er = float(np.linalg.norm(got - xs[i])), float(np.linalg.norm(rs[i] - xs[i]))
            num_h, num_r, den = num_h + eh * eh, num_r + er * er, den + float(np.sum(xs[i] ** 2))
            if scale > 0:
                er_rel = max(er / scale, float(xerr[i]) / max(float(xnorm[i]), 1e-300))     # 64-sample estimate / whole tensor
                # (tensors of a handful of elements -- the 2 .. 4 logit biases, the value bias: every element is a sum of ~10^3 cancelling per-sample
                # terms -- get twice the absolute floor in the first step too: policy_branches.0.bias at cfg3 measured 1.66e-6 with the fp32-MFMA
                # encoder kernels and 2.10e-6 with the bf16-pipe ones of round 6, whose own results are 3 x CLOSER to float64 -- the element's
                # error is a re-roll of the heads' summation noise, not a property of either encoder)
                abs_floor = _KF_GRAD_ABS_TENSOR * (2.0 if grads[k].numel() < 64 else 1.0)
                allowed = max(_KF_GRAD_RATIO_TENSOR[min(s, 1)] * er_rel, abs_floor) + _KF_MOVE_RATIO * float(xshift[i]) / max(float(xnorm[i]), 1e-300)
                if eh / scale / allowed > worst[0]:
                    worst = (eh / scale / allowed, k, eh / scale)
                per_tensor.append((k, eh / scale, er / scale, float(xerr[i]) / max(float(xnorm[i]), 1e-300), float(xshift[i]) / max(float(xnorm[i]), 1e-300)))
                if s > 0 and grads[k].numel() < 64:
                    # later steps, tensors of a handful of elements (the 2 .. 4 logit biases, the value bias): every element is a sum of
                    # ~10^3 cancelling per-sample terms, evaluated at parameters that are no longer the reference's -- their relative
                    # error has no floor the fixture records (measured: policy_branches.0.bias, 2 elements, 1.0e-5 at cfg2 step 2 with
                    # the reference itself at 1.1e-6).  They stay under the all-tensors bound and the movement bound below.
                    continue
                if eh / scale > allowed:
                    violations.append(f"{k}: {eh / scale:.2e} of its norm from the float64 evaluation; the reference's fp32 gradient "
                                      f"is {er / scale:.2e} from it (bound {allowed:.2e})")
        hip_all, ref_all = (num_h / den) ** 0.5, (num_r / den) ** 0.5
        print(f"[kink-free {name}/{mode} step {s}] {idx.size} samples; gradient vs float64, all tensors: HIP {hip_all:.2e}, reference {ref_all:.2e} "
              f"(ratio {hip_all / ref_all:.2f}); tensor closest to its bound: {worst[1]} at {worst[2]:.2e} ({worst[0]:.2f} of the bound)")
        if os.environ.get("ETM_KF_TENSOR_LOG"):        # every tensor's (HIP error, reference error on the samples / whole tensor, twin shift): diagnostics
            with open(os.environ["ETM_KF_TENSOR_LOG"], "a") as f:
                f.write(json.dumps({"case": name, "mode": mode, "step": s, "tensors": per_tensor}) + "\n")
        assert not violations, f"{name} kink-free step {s}: " + "; ".join(violations)
        assert hip_all <= _KF_GRAD_RATIO_ALL * ref_all, (hip_all, ref_all)
        # ---- the step itself, then the parameter movement
        if mode == "api":       # the upstream-API path: gathered minibatch -> _train_mini_batch
            tr._train_mini_batch(tr.buffer.gather(torch.as_tensor(idx, device=dev)), lr, clip, beta)
        else:                   # the captured optimisation step on sorted indices (what run_training / bench.py replay)
            idx_t = torch.as_tensor(idx, device=dev, dtype=torch.long).sort().values
            if getattr(tr, "_tg_idx", None) is not None and tr._tg_idx.numel() != idx_t.numel():
                tr._train_graph, tr._tg_idx = None, None         # another minibatch size: capture again (workspaces are kept alive)
            with torch.no_grad():
                tr._bank_pos, tr._obs_train = tr._bank_with_positions(), tr._observations_channels_last()
            tr._train_step_graph(idx_t, lr, clip, beta, False)
            assert tr._train_graph is not None and tr._tg_idx.numel() == idx_t.numel(), "the step must have been a graph replay"
        a_rows, x_rows = rows(st + "sd_samples"), rows(st + "sd_exact_samples")
        num_a = num_x = num_f = den = 0.0
        for i, k in enumerate(pnames):
            got = dg.sample(params[k].detach().cpu().numpy(), 64).astype(np.float64)
            num_a += float(np.sum((got - a_rows[i]) ** 2))
            num_x += float(np.sum((got - x_rows[i]) ** 2))
            num_f += float(np.sum((a_rows[i] - x_rows[i]) ** 2))
            den += float(np.sum((x_rows[i] - init[k]) ** 2))
        mv_a, mv_x, floor_s = (num_a / den) ** 0.5, (num_x / den) ** 0.5, (num_f / den) ** 0.5
        # the twins' distance: over whole tensors (recorded by the generator) or over the 64-element samples, whichever is larger --
        # the movement error sits in few elements (those whose gradient is of the size of its noise), so a 64-element sample of it
        # scatters by a factor of a few either way (cfg3 step 0: 2.4e-6 on the samples, 9.3e-6 over whole tensors)
        floor = max(floor_s, float(z[st + "floor_move_all"]))
        print(f"[kink-free {name} step {s}] parameter movement error (sampled elements, from the initial parameters): HIP vs the reference's "
              f"fp32 twin {mv_a:.2e}, vs the float64-gradient twin {mv_x:.2e}; the twins differ by {floor_s:.2e} "
              f"(whole tensors: {float(z[st + 'floor_move_all']):.2e})")
        if os.environ.get("ETM_TF_MEASURE_LOG"):
            with open(os.environ["ETM_TF_MEASURE_LOG"], "a") as f:
                f.write(json.dumps({"case": name, "path": "kink_free" if mode == "api" else "kink_free_graph", "step": s, "samples": int(idx.size), "grad_hip_vs_exact": hip_all,
                                    "grad_ref_vs_exact": ref_all, "grad_tensor_nearest_bound": worst[1], "grad_tensor_err": worst[2],
                                    "grad_tensor_frac_of_bound": worst[0], "move_vs_ref": mv_a, "move_vs_exact": mv_x, "move_floor": floor}) + "\n")
        assert mv_a <= _KF_MOVE_RATIO * floor and mv_x <= _KF_MOVE_RATIO * floor, (mv_a, mv_x, floor)
    tr.close()


def test_rollout_conv3_hidden_vs_float64():
    """csrc/conv3_hidden.hip (round 4): the last encoder layer + lin_hidden's partial sums of a rollout step in one launch, one
    workgroup per output pixel -- the summed rows + bias + ReLU against relu(linear(flatten(relu(conv2d)))) in float64
    (model.py:92-97), for ragged image counts and both hidden sizes of the BASELINE configs."""
    from etm import ops
    dev = _dev()
    torch.manual_seed(11)
    for (W, hi, wi, D) in ((8, 9, 9, 384), (5, 9, 9, 384), (16, 9, 9, 512), (3, 5, 6, 128)):
        conv = torch.nn.Conv2d(64, 64, 3, 1).to(dev)
        lin = torch.nn.Linear(64 * (hi - 2) * (wi - 2), D).to(dev)
        x2 = torch.rand((W, hi, wi, 64), device=dev)                     # NHWC, as the second convolution leaves it
        assert ops.rollout_conv3_hidden_supported(conv, hi, wi, D)
        w3k = conv.weight.detach().permute(2, 3, 1, 0).reshape(-1, 64).contiguous()
        part = ops.rollout_conv3_hidden(x2, w3k, conv.bias.detach(), lin.weight.detach().t().contiguous())
        assert part.shape == ((hi - 2) * (wi - 2), W, D)
        got = torch.relu(part.sum(dim=0) + lin.bias.detach())
        with torch.no_grad():
            f = torch.relu(torch.nn.functional.conv2d(x2.permute(0, 3, 1, 2).double().cpu(), conv.weight.double().cpu(), conv.bias.double().cpu()))
            ref = torch.relu(f.reshape(W, -1) @ lin.weight.double().cpu().t() + lin.bias.double().cpu())
        err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, (W, hi, wi, D, err)


def test_trainer_self_consistency_and_free_run():
    """Recomputed values from the buffer equal the rollout values for every sample with a non-empty mask row
    (SURVEY quirk Q10), and a free-running update produces finite statistics."""
    from trainer import PPOTrainer
    from etm.ops import WindowSpec
    dev = _dev()
    cfg = dict(environment=dict(type="Synthetic", obs_shape=[3, 84, 84], num_actions=3, max_episode_steps=40, seed=1, p_done=0.03, pool=8),
               gamma=0.995, lamda=0.95, updates=1, epochs=2, n_workers=8, worker_steps=96, n_mini_batch=4, value_loss_coefficient=0.5,
               hidden_layer_size=128, max_grad_norm=0.5,
               transformer=dict(num_blocks=3, embed_dim=128, num_heads=4, memory_length=32, positional_encoding="relative",
                                layer_norm="post", gtrxl=False, gtrxl_bias=0.0),
               learning_rate_schedule=dict(initial=3e-4, final=3e-4, power=1.0, max_decay_steps=10),
               beta_schedule=dict(initial=1e-3, final=1e-3, power=1.0, max_decay_steps=10),
               clip_range_schedule=dict(initial=0.1, final=0.1, power=1.0, max_decay_steps=10))
    torch.manual_seed(0)
    tr = PPOTrainer(cfg, run_id="selfcheck", device=dev, tensorboard=False)
    for _ in range(2):
        infos = tr._sample_training_data()
        tr.buffer.prepare_batch_dict()
        flat = tr.buffer.samples_flat
        with torch.no_grad():
            spec = WindowSpec.from_bank(tr.buffer.memories, flat["memory_index"], flat["memory_indices"], flat["memory_indices"], flat["memory_mask"])
            _, value, _ = tr.model.forward_logits(flat["obs"], spec)
        keep = flat["memory_mask"].any(dim=1)
        assert keep.sum() > 0.8 * keep.numel()
        assert torch.allclose(value[keep], flat["values"][keep], atol=2e-4), (value[keep] - flat["values"][keep]).abs().max()
        stats, grads = tr._train_epochs(3e-4, 0.1, 1e-3)
        assert np.isfinite(np.asarray(stats)).all() and len(stats) == 8
        assert set(grads) == {"encoder", "linear_layer", "transformer_block_0", "transformer_block_1", "transformer_block_2",
                              "policy_head_0", "lin_policy", "value", "model"}
    assert len(infos) > 0 and all({"reward", "length"} <= set(i) for i in infos)
    tr.close()


# ------------------------------------------------------------------ rollout K/V cache path
def test_attn_cached_matches_fused_kernel():
    """Cached-projection attention (rollout) == the fused projection+attention kernel on the same window."""
    from etm import ops
    dev = _dev()
    torch.manual_seed(3)
    for (W, T, nb, D, H, L, ln) in ((9, 40, 3, 384, 4, 32, False), (5, 130, 2, 128, 1, 128, True), (7, 20, 2, 64, 2, 7, True)):
        bank = torch.randn((W, T, nb, D), device=dev)
        pos = torch.randn((T, D), device=dev) * 0.5
        wk = torch.randn((D, D), device=dev) / D ** 0.5
        wv = torch.randn((D, D), device=dev) / D ** 0.5
        g = 1 + 0.1 * torch.randn((D,), device=dev)
        b = 0.1 * torch.randn((D,), device=dev)
        q = torch.randn((W, D), device=dev)
        win = torch.randint(0, T - L + 1, (W, 1), device=dev) + torch.arange(L, device=dev)[None, :]
        cnt = torch.randint(0, L + 1, (W,), device=dev)
        cnt[0] = 0
        mask = torch.arange(L, device=dev)[None, :] < cnt[:, None]
        blk = nb - 1
        spec = ops.WindowSpec.from_bank(bank, None, win, win, mask)
        with torch.no_grad():
            ref, ref_att = ops.mha(q, wk, wv, spec, blk, H, g if ln else None, b if ln else None, pos)
            x = bank[:, :, blk] + pos[None]
            if ln:
                x = torch.nn.functional.layer_norm(x, (D,), g, b, 1e-5)
            cache = torch.zeros((W, T, nb, 2 * D), device=dev)
            cache[:, :, blk, :D] = x @ wk.t()
            cache[:, :, blk, D:] = x @ wv.t()
            kv_spec = ops.WindowSpec.from_bank(cache, None, win, None, mask)
            ctx, att = ops.attn_cached(q, kv_spec, blk, H, want_att=True)
        close(ctx, ref.cpu().numpy(), atol=5e-5, rtol=1e-4, what="ctx")
        close(att, ref_att.cpu().numpy(), atol=2e-6, rtol=1e-4, what="att")
    # reset_rows touches only workers at episode step 0
    dst = torch.ones((4, 6, 8), device=dev)
    init = torch.arange(48, dtype=torch.float32, device=dev).reshape(6, 8)
    ops.reset_rows(dst, init, torch.tensor([0, 3, 0, 1], device=dev))
    assert torch.equal(dst[0], init) and torch.equal(dst[2], init) and bool((dst[1] == 1).all()) and bool((dst[3] == 1).all())


def test_rollout_cache_and_graph_paths_agree():
    """Same seeds, same recorded actions: rollout with K/V cache (eager), without it, must fill the buffer identically
    (up to fp32 summation order), including episode bookkeeping."""
    from trainer import PPOTrainer
    dev = _dev()
    base = dict(environment=dict(type="Synthetic", obs_shape=[6], num_actions=3, max_episode_steps=24, seed=2, p_done=0.06, pool=8),
                gamma=0.99, lamda=0.95, updates=1, epochs=1, n_workers=6, worker_steps=48, n_mini_batch=2, value_loss_coefficient=0.5,
                hidden_layer_size=64, max_grad_norm=0.5,
                transformer=dict(num_blocks=2, embed_dim=64, num_heads=2, memory_length=16, positional_encoding="relative",
                                 layer_norm="pre", gtrxl=True, gtrxl_bias=0.0),
                learning_rate_schedule=dict(initial=3e-4, final=3e-4, power=1.0, max_decay_steps=10),
                beta_schedule=dict(initial=1e-3, final=1e-3, power=1.0, max_decay_steps=10),
                clip_range_schedule=dict(initial=0.1, final=0.1, power=1.0, max_decay_steps=10))
    rng = np.random.default_rng(0)
    acts = rng.integers(0, 3, size=(2, 6, 48))
    results = []
    for use_cache in (True, False):
        cfg = json.loads(json.dumps(base))
        cfg["kv_cache_rollout"] = use_cache
        torch.manual_seed(11)
        tr = PPOTrainer(cfg, run_id="cache", device=dev, tensorboard=False)
        snap = []
        for u in range(2):   # second rollout starts with live episodes whose cache must be re-projected
            tr._sample_training_data(forced_actions=acts[u])
            tr.buffer.prepare_batch_dict()
            b = tr.buffer
            snap.append({k: getattr(b, k).clone() for k in ("values", "log_probs", "advantages", "memory_mask", "memory_indices", "memory_index")})
            snap[-1]["memories"] = b.memories.clone()
            tr._train_epochs(3e-4, 0.1, 1e-3, perms=[np.arange(6 * 48)])
        results.append(snap)
        tr.close()
    for a, b in zip(*results):
        for k in ("memory_mask", "memory_indices", "memory_index"):
            assert torch.equal(a[k], b[k]), k
        for k in ("values", "log_probs", "advantages", "memories"):
            assert torch.allclose(a[k], b[k], atol=2e-4, rtol=1e-3), (k, (a[k] - b[k]).abs().max())


def _rollout_variants_agree(variants, n_workers=16, checks=True):
    """Two updates per config variant (same torch seed => same device uniforms): every variant must sample the same actions and
    fill the buffer like the first one."""
    from trainer import PPOTrainer
    dev = _dev()
    base = dict(environment=dict(type="Synthetic", obs_shape=[3, 36, 36], num_actions=4, max_episode_steps=20, seed=5, p_done=0.08, pool=8),
                gamma=0.99, lamda=0.95, updates=1, epochs=1, n_workers=n_workers, worker_steps=40, n_mini_batch=2, value_loss_coefficient=0.5,
                hidden_layer_size=64, max_grad_norm=0.5,
                transformer=dict(num_blocks=2, embed_dim=64, num_heads=2, memory_length=8, positional_encoding="relative",
                                 layer_norm="post", gtrxl=False, gtrxl_bias=0.0),
                learning_rate_schedule=dict(initial=3e-4, final=3e-4, power=1.0, max_decay_steps=10),
                beta_schedule=dict(initial=1e-3, final=1e-3, power=1.0, max_decay_steps=10),
                clip_range_schedule=dict(initial=0.1, final=0.1, power=1.0, max_decay_steps=10))
    results = []
    for over in variants:
        cfg = json.loads(json.dumps(base))
        cfg.update(over)
        torch.manual_seed(3)
        tr = PPOTrainer(cfg, run_id="paths", device=dev, tensorboard=False)
        snap = []
        for u in range(2):
            tr._sample_training_data()
            tr.buffer.prepare_batch_dict()
            b = tr.buffer
            snap.append({k: getattr(b, k).clone() for k in ("obs", "actions", "values", "log_probs", "advantages", "memory_mask",
                                                            "memory_indices", "memory_index")})
            snap[-1]["rewards"] = torch.from_numpy(np.asarray(b.rewards).copy())
            tr._train_epochs(3e-4, 0.1, 1e-3, perms=[np.arange(n_workers * 40)])
        if not over and checks:
            assert tr._stream_obs and tr._host_flag and len(tr._groups) == 2, "default config: streamed, flag hand-over, two worker groups"
        if "rollout_groups" in over:
            assert len(tr._groups) == over["rollout_groups"]
        if over.get("host_flag_actions") is False:
            assert not tr._host_flag
        results.append(snap)
        tr.close()
    for other in results[1:]:
        for a, b in zip(results[0], other):
            for k in ("obs", "actions", "memory_mask", "memory_indices", "memory_index", "rewards"):
                assert torch.equal(a[k], b[k]), k
            for k in ("values", "log_probs", "advantages"):
                assert torch.allclose(a[k], b[k], atol=1e-5, rtol=1e-5), (k, (a[k] - b[k]).abs().max())


def test_rollout_fast_paths_agree():
    """Graph rollout with observation streaming + host-flag action hand-over (defaults), the graph without them, and the eager
    step must sample the same actions and fill the buffer identically."""
    _rollout_variants_agree([dict(), dict(host_flag_actions=False), dict(rollout_groups=1), dict(stream_observations=False),
                             dict(stream_observations=False, rollout_groups=1), dict(hip_graph_rollout=False)])


def test_fresh_observation_draws_reach_the_device_bit_for_bit():
    """Round 6, the default environment path of the benchmark: every observation is a fresh draw of its worker's numpy stream
    (`pool: 0`), made by libetm_envgen.so (or numpy without it) STRAIGHT into the staging array in device memory where the host can
    write there (large BAR; else through pinned memory + uploads), over several rollouts of a pipelined trainer: `buffer.obs[w, t]` of
    rollout r must be draw number r * S + t of `default_rng(seed + w)` -- regenerated here by numpy alone."""
    from etm import ops
    from trainer import PPOTrainer
    dev = _dev()
    W, S, R, shape, seed = 8, 24, 4, (3, 36, 36), 11
    base = dict(environment=dict(type="Synthetic", obs_shape=list(shape), num_actions=3, max_episode_steps=9, seed=seed, p_done=0.1, pool=0, gen_threads=3),
                gamma=0.99, lamda=0.95, updates=1, epochs=1, n_workers=W, worker_steps=S, n_mini_batch=2, value_loss_coefficient=0.5,
                hidden_layer_size=32, max_grad_norm=0.5, rollout_groups=2, rollout_min_group_size=2,
                transformer=dict(num_blocks=2, embed_dim=64, num_heads=2, memory_length=6, positional_encoding="relative",
                                 layer_norm="post", gtrxl=False, gtrxl_bias=0.0),
                learning_rate_schedule=dict(initial=3e-4, final=3e-4, power=1.0, max_decay_steps=10),
                beta_schedule=dict(initial=1e-3, final=1e-3, power=1.0, max_decay_steps=10),
                clip_range_schedule=dict(initial=0.1, final=0.1, power=1.0, max_decay_steps=10))
    n = int(np.prod(shape))
    want = np.stack([np.random.default_rng(seed + w).random((R * S + 1) * n, dtype=np.float32).reshape(R * S + 1, *shape) for w in range(W)])
    for direct in (True, False):
        cfg = json.loads(json.dumps(base))
        cfg["direct_observation_rows"] = direct
        torch.manual_seed(3)
        tr = PPOTrainer(cfg, run_id="fresh", device=dev, tensorboard=False)
        for r in range(R):
            tr._sample_training_data()
            tr.buffer.prepare_batch_dict()
            got = tr.buffer.obs.cpu().numpy()                      # [W, S, ...]
            assert np.array_equal(got, want[:, r * S:(r + 1) * S]), (direct, r)
        assert len(tr._groups) == 2 and tr._stream_obs
        assert tr._direct_rows == (direct and ops.host_direct_write_ok(dev))
        tr.close()


def test_small_worker_groups_stress():
    """Pipelined worker groups of TWO workers with a near-free environment (the host is back with the next step's (episode
    step, slot) block long before the tail of the current step has run): 12 teacher-forced rollouts of a gated (GTrXL) model
    must fill the buffer and the episode bank exactly like the eager single-group path.  Guards the (step, slot) latch of the
    step head -- without it the upload of step t + 1 raced with the bank / cache writes of step t."""
    from trainer import PPOTrainer
    dev = _dev()
    W, S, R = 4, 48, 12
    base = dict(environment=dict(type="Synthetic", obs_shape=[3, 36, 36], num_actions=3, max_episode_steps=9, seed=4, p_done=0.15, pool=4),
                gamma=0.99, lamda=0.95, updates=1, epochs=1, n_workers=W, worker_steps=S, n_mini_batch=2, value_loss_coefficient=0.5,
                hidden_layer_size=32, max_grad_norm=0.5,
                transformer=dict(num_blocks=2, embed_dim=64, num_heads=2, memory_length=6, positional_encoding="relative",
                                 layer_norm="pre", gtrxl=True, gtrxl_bias=0.0),
                learning_rate_schedule=dict(initial=3e-4, final=3e-4, power=1.0, max_decay_steps=10),
                beta_schedule=dict(initial=1e-3, final=1e-3, power=1.0, max_decay_steps=10),
                clip_range_schedule=dict(initial=0.1, final=0.1, power=1.0, max_decay_steps=10))
    acts = np.random.default_rng(7).integers(0, 3, size=(R, W, S))
    results = []
    # ... and (round 4) the same through worker PROCESSES of one environment each and the native per-step driver: the workers answer
    # within microseconds of the device's go word, the driver's (step, slot) block and step graph follow at once
    for over in (dict(hip_graph_rollout=False), dict(rollout_groups=2, rollout_min_group_size=2),
                 dict(rollout_groups=2, rollout_min_group_size=2, worker_processes=True, envs_per_process=1)):
        cfg = json.loads(json.dumps(base))
        cfg.update(over)
        torch.manual_seed(5)
        tr = PPOTrainer(cfg, run_id="stress", device=dev, tensorboard=False)
        snaps = []
        for r in range(R):
            tr._sample_training_data(forced_actions=acts[r])
            tr.buffer.prepare_batch_dict()
            b = tr.buffer
            snap = {k: getattr(b, k).clone() for k in ("values", "log_probs", "advantages", "memory_mask", "memory_indices", "memory_index")}
            snap["memories"] = b.memories.clone()
            snaps.append(snap)
        # round 6: the bootstrap value of the last observation (get_last_value) is a graph replay from the third rollout on; the
        # advantages compared below are its consumers, against the eager path's
        assert (tr._lv.graph is not None) == ("rollout_groups" in over), over
        # ... likewise the per-update weight repacking and the K | V cache refresh (ops.ReplayAfterWarmup)
        assert (tr.model._refresh_replay.graph is not None) == ("rollout_groups" in over) and (tr._kv_refresh_replay.graph is not None) == ("rollout_groups" in over), over
        if "rollout_groups" in over:
            assert len(tr._groups) == 2 and tr._groups[0].W == 2 and tr._stream_obs
            assert bool(getattr(tr, "_native_rollout", False)) == bool(over.get("worker_processes")), over
        results.append(snaps)
        tr.close()
    for other in results[1:]:
        for r, (a, b) in enumerate(zip(results[0], other)):
            for k in ("memory_mask", "memory_indices", "memory_index"):
                assert torch.equal(a[k], b[k]), (r, k)
            for k in ("values", "log_probs", "advantages", "memories"):
                assert a[k].shape == b[k].shape and torch.allclose(a[k], b[k], atol=2e-5, rtol=1e-4), (r, k, (a[k] - b[k]).abs().max())


def test_rollout_group_counts_agree():
    """One, two and four pipelined worker groups (and the host-flag hand-over) against the shipped defaults at 32 workers."""
    _rollout_variants_agree([dict(), dict(rollout_groups=4), dict(rollout_groups=1), dict(rollout_groups=4, host_flag_actions=False)],
                            n_workers=32)


def test_fused_rollout_step_kernel_vs_multi_launch_path():
    """etm_rollout_trxl (transformer + heads + sampling of a rollout step in one launch) against the multi-launch path
    (library GEMMs, cached attention, residual + LayerNorm and policy kernels) on the same weights, cache and observations:
    memory items, values, log-probs and -- for every worker whose uniform is not within 1e-5 of a CDF boundary -- actions."""
    from etm import lib as etm_lib
    from trainer import PPOTrainer
    dev = _dev()
    # teams of 4 / 2 / 1 workgroups; two heads per member with a window beyond 64 rows; D = 512 (the 32-row register slices)
    # ... and the general block layouts of transformer.py:117-172: pre-LN and / or GRU gates (BASELINE configs 2 and 5 are pre-LN GTrXL)
    for (D, H, L, nb, hid, A, W, ln, gtrxl) in ((384, 4, 64, 3, 384, 3, 16, "post", False), (64, 2, 8, 2, 64, 4, 5, "post", False),
                                                (128, 1, 32, 4, 128, 2, 32, "post", False), (256, 8, 96, 2, 256, 5, 9, "post", False),
                                                (512, 4, 33, 2, 512, 3, 8, "post", False),
                                                (384, 4, 128, 2, 384, 3, 16, "pre", True), (128, 1, 32, 4, 128, 2, 8, "pre", True),
                                                (64, 2, 8, 2, 64, 4, 5, "post", True), (256, 8, 96, 2, 256, 5, 9, "pre", False)):
        cfg = dict(environment=dict(type="Synthetic", obs_shape=[7], num_actions=A, max_episode_steps=L + 5, seed=3, p_done=0.1, pool=4),
                   gamma=0.99, lamda=0.95, updates=1, epochs=1, n_workers=W, worker_steps=L + 12, n_mini_batch=1, value_loss_coefficient=0.5,
                   hidden_layer_size=hid, max_grad_norm=0.5, rollout_groups=1,
                   transformer=dict(num_blocks=nb, embed_dim=D, num_heads=H, memory_length=L, positional_encoding="relative",
                                    layer_norm=ln, gtrxl=gtrxl, gtrxl_bias=1.0 if gtrxl else 0.0),
                   learning_rate_schedule=dict(initial=3e-4, final=3e-4, power=1.0, max_decay_steps=10),
                   beta_schedule=dict(initial=1e-3, final=1e-3, power=1.0, max_decay_steps=10),
                   clip_range_schedule=dict(initial=0.1, final=0.1, power=1.0, max_decay_steps=10))
        snaps = []
        # the one-launch step under both workgroup placements (a team on one XCD / one member index per XCD: its members exchange
        # through system-scope packets only, so results must not depend on where they run), then the multi-launch path
        for fused, placement in ((True, "team_xcd"), (True, "member_xcd"), (False, "team_xcd")):
            c = json.loads(json.dumps(cfg))
            c["fused_rollout_block"] = fused
            torch.manual_seed(17)
            tr = PPOTrainer(c, run_id="fusedstep", device=dev, tensorboard=False)
            # (process-wide library switch, read at launch and by etm_rollout_trxl_grid; the trainer sets the default, team_xcd)
            etm_lib.check(etm_lib.load().etm_rollout_trxl_set_placement(1 if placement == "member_xcd" else 0), "set_placement")
            with torch.no_grad():
                for prm in tr.model.parameters():          # non-trivial LayerNorm gains / biases
                    if prm.dim() == 1:
                        prm.add_(0.1 * torch.randn_like(prm))
            tr._sample_training_data()
            assert (tr.model._rf is not None) == fused
            tr.buffer.prepare_batch_dict()
            b = tr.buffer
            snaps.append({k: getattr(b, k).clone() for k in ("actions", "values", "log_probs", "memory_index")} | {"mem": b.memories.clone()})
            tr.close()
            etm_lib.load().etm_rollout_trxl_set_placement(0)
        a, a2, m = snaps
        for k in ("actions", "values", "log_probs", "mem"):
            assert torch.equal(a[k], a2[k]), (D, ln, gtrxl, k, "the two placements of the step kernel must agree bit for bit")
        assert torch.equal(a["memory_index"], m["memory_index"])
        same = (a["actions"] == m["actions"]).float().mean().item()
        assert same > 0.999, same                            # a different action only where a uniform sits on a CDF boundary
        if same == 1.0:
            for k in ("values", "log_probs", "mem"):
                assert torch.allclose(a[k], m[k], atol=2e-5, rtol=1e-4), (D, ln, gtrxl, k, (a[k] - m[k]).abs().max())


def test_group_rollout_step_kernel_vs_the_other_rollout_paths():
    """etm_rollout_trxl_group (round 5, csrc/rollout_group.hip: GRU-gated blocks, the <= 8 workers of a group as the rows of every
    product, columns over 32 workgroups, weights read once per group and step) against the per-worker step kernel
    (etm_rollout_trxl, GEN instantiation) and the multi-launch path on the same weights, cache and observations: pre- and post-LN
    gated layouts, D = 384 / H = 4 (config 5: L = 128, four blocks) and D = 128 / H = 1 (config 2), full and ragged groups
    (W = 8, 6, 3), several groups side by side, 2 .. 5 actions."""
    from trainer import PPOTrainer
    dev = _dev()
    for (D, H, L, nb, hid, A, W, groups, ln) in ((384, 4, 128, 4, 384, 4, 8, 1, "pre"), (384, 4, 64, 2, 384, 3, 6, 1, "pre"),
                                                 (128, 1, 32, 4, 128, 2, 8, 1, "pre"), (384, 4, 40, 2, 384, 5, 8, 1, "post"),
                                                 (128, 1, 32, 2, 128, 2, 3, 1, "pre"), (384, 4, 64, 2, 384, 3, 16, 2, "pre"),
                                                 (128, 1, 16, 2, 128, 3, 16, 4, "post")):
        cfg = dict(environment=dict(type="Synthetic", obs_shape=[7], num_actions=A, max_episode_steps=L + 5, seed=3, p_done=0.1, pool=4),
                   gamma=0.99, lamda=0.95, updates=1, epochs=1, n_workers=W, worker_steps=L + 12, n_mini_batch=1, value_loss_coefficient=0.5,
                   hidden_layer_size=hid, max_grad_norm=0.5, rollout_groups=groups, rollout_min_group_size=2,
                   transformer=dict(num_blocks=nb, embed_dim=D, num_heads=H, memory_length=L, positional_encoding="relative",
                                    layer_norm=ln, gtrxl=True, gtrxl_bias=1.0),
                   learning_rate_schedule=dict(initial=3e-4, final=3e-4, power=1.0, max_decay_steps=10),
                   beta_schedule=dict(initial=1e-3, final=1e-3, power=1.0, max_decay_steps=10),
                   clip_range_schedule=dict(initial=0.1, final=0.1, power=1.0, max_decay_steps=10))
        snaps = []
        for variant in (dict(), dict(rollout_group_kernel=False), dict(fused_rollout_block=False)):
            c = {**json.loads(json.dumps(cfg)), **variant}
            torch.manual_seed(23)
            tr = PPOTrainer(c, run_id="groupstep", device=dev, tensorboard=False)
            with torch.no_grad():
                for prm in tr.model.parameters():          # non-trivial LayerNorm gains / biases / gate biases
                    if prm.dim() == 1:
                        prm.add_(0.1 * torch.randn_like(prm))
            tr._sample_training_data()
            used = [bool(getattr(g, "group_kernel", False)) for g in tr._groups]
            assert all(used) == (not variant), (D, W, groups, variant, used)
            assert len(tr._groups) == groups
            tr.buffer.prepare_batch_dict()
            b = tr.buffer
            snaps.append({k: getattr(b, k).clone() for k in ("actions", "values", "log_probs", "memory_index")} | {"mem": b.memories.clone()})
            tr.close()
        grp, per_worker, multi = snaps
        for other, what in ((per_worker, "per-worker step kernel"), (multi, "multi-launch path")):
            assert torch.equal(grp["memory_index"], other["memory_index"])
            same = (grp["actions"] == other["actions"]).float().mean().item()
            assert same > 0.999, (D, W, ln, what, same)          # a different action only where a uniform sits on a CDF boundary
            if same == 1.0:
                for k in ("values", "log_probs", "mem"):
                    assert torch.allclose(grp[k], other[k], atol=2e-5, rtol=1e-4), (D, W, ln, what, k, (grp[k] - other[k]).abs().max())


def test_window_ln_grad_and_bank_row_stats_vs_float64():
    """Round 5 (csrc/window_ln_grad.hip): norm_kv's gain / bias gradients of the folded pre-LN attention by the dedicated window pass
    == the float64 contraction sum_{n,l} dY xhat / sum dY with dY = sum_h dE u + att gz, and == the old generic dX kernel
    (etm_window_dx); etm_ln_row_stats == float64 row statistics; the statistics a WindowSpec gathers from per-bank-row statistics
    equal the ones etm_window_fwd computes per window row (bit for bit: same kernel arithmetic per row)."""
    import ctypes
    from etm import lib as etm_lib
    from etm import ops
    from etm.ops import WindowSpec
    dev = _dev()
    lib = etm_lib.load()
    torch.manual_seed(5)
    for (N, L, D, H, E, T, nb) in ((37, 128, 384, 4, 9, 140, 2), (130, 32, 128, 1, 20, 40, 3), (64, 64, 256, 8, 7, 70, 1)):
        bank_mem = torch.randn((nb, E, T, D), device=dev) * 0.7 + 0.1
        bank = bank_mem.permute(1, 2, 0, 3)                                   # [E, T, nb, D] view of block-major memory
        ep = torch.randint(0, E, (N,), device=dev)
        win = torch.randint(0, T, (N, L), device=dev)
        mask = (torch.rand((N, L), device=dev) < 0.7)
        mask[:, 0] = True
        block = nb - 1
        u, gz = torch.randn((H, N, D), device=dev), torch.randn((H, N, D), device=dev)
        att = torch.rand((N, H, L), device=dev) * mask[:, None, :]
        d_e = torch.randn((N, H, L), device=dev) * mask[:, None, :]
        spec = WindowSpec.from_bank(bank, ep, win, None, mask)
        # row statistics
        stats_bank = ops.bank_row_stats(bank, 1e-5)                             # [nb, E, T, 2]
        x64 = bank_mem.double()
        mean64, var64 = x64.mean(-1), x64.var(-1, unbiased=False)
        assert float((stats_bank[..., 0].double() - mean64).abs().max()) < 1e-6
        assert float((stats_bank[..., 1].double() * torch.sqrt(var64 + 1e-5) - 1).abs().max()) < 1e-5
        spec.row_stats = stats_bank
        gathered = spec.window_stats(block)                                     # [N, L, 2]
        ln_g, ln_b = torch.randn(D, device=dev), torch.randn(D, device=dev)
        z = torch.empty((H, N, D), device=dev); a_out = torch.empty((N, H, L), device=dev)
        per_window = torch.empty((N, L, 2), device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        m8 = mask.contiguous().view(torch.uint8)
        etm_lib.check(lib.etm_window_fwd(spec.block_ptr(block), spec.ep_stride, spec.row_stride, ep.data_ptr(), win.data_ptr(), None, m8.data_ptr(),
                                         None, ln_g.data_ptr(), ln_b.data_ptr(), 1e-5, u.data_ptr(), N * D, D, a_out.data_ptr(), z.data_ptr(), N * D, D,
                                         per_window.data_ptr(), 0, N, L, D, H, st), "etm_window_fwd")
        assert float((per_window - gathered).abs().max()) < 2e-6 * float(per_window.abs().max())
        z2 = torch.empty_like(z); a2 = torch.empty_like(a_out)
        etm_lib.check(lib.etm_window_fwd(spec.block_ptr(block), spec.ep_stride, spec.row_stride, ep.data_ptr(), win.data_ptr(), None, m8.data_ptr(),
                                         None, ln_g.data_ptr(), ln_b.data_ptr(), 1e-5, u.data_ptr(), N * D, D, a2.data_ptr(), z2.data_ptr(), N * D, D,
                                         gathered.data_ptr(), 1, N, L, D, H, st), "etm_window_fwd")
        assert torch.allclose(z, z2, atol=1e-5, rtol=1e-5) and torch.allclose(a_out, a2, atol=1e-6, rtol=1e-5)
        # gain / bias gradients
        rows = lib.etm_window_ln_grad_rows(N)
        partial = torch.full((rows, 2 * D), float("nan"), device=dev)
        etm_lib.check(lib.etm_window_ln_grad(spec.block_ptr(block), spec.ep_stride, spec.row_stride, ep.data_ptr(), win.data_ptr(), None, None,
                                             gathered.data_ptr(), att.data_ptr(), d_e.data_ptr(), u.data_ptr(), gz.data_ptr(), N * D, D,
                                             partial.data_ptr(), N, L, D, H, st), "etm_window_ln_grad")
        got = partial.double().sum(0)
        xw = bank_mem[block].double()[ep[:, None], win]                           # [N, L, D]
        xhat = (xw - xw.mean(-1, keepdim=True)) / torch.sqrt(xw.var(-1, unbiased=False, keepdim=True) + 1e-5)
        dY = torch.einsum("nhl,hnd->nld", d_e.double(), u.double()) + torch.einsum("nhl,hnd->nld", att.double(), gz.double())
        want = torch.cat(((dY * xhat).sum((0, 1)), dY.sum((0, 1))))
        err = float((got - want).norm() / want.norm())
        assert err < 2e-6, (N, L, D, H, err)
        # the old generic kernel on the same inputs
        d_g, d_b = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        uw = torch.stack((u.transpose(0, 1), gz.transpose(0, 1))).contiguous()
        etm_lib.check(lib.etm_window_dx(spec.block_ptr(block), spec.ep_stride, spec.row_stride, ep.data_ptr(), win.data_ptr(), None, None,
                                        ln_g.data_ptr(), ln_b.data_ptr(), gathered.data_ptr(), att.data_ptr(), d_e.data_ptr(), uw.data_ptr(),
                                        d_g.data_ptr(), d_b.data_ptr(), None, N, L, D, H, st), "etm_window_dx")
        old = torch.cat((d_g, d_b)).double()
        assert float((old - want).norm() / want.norm()) < 2e-5 and float((got - old).norm() / want.norm()) < 2e-5
        # round 6: the same gradients from the window passes' OUTPUTS (no window row is read): the real forward / backward pair --
        # (att, z) from etm_window_fwd, (d_e, du) from etm_window_bwd on a gradient gz -- against float64 and against the rows kernel
        d_e2, du2 = torch.empty((N, H, L), device=dev), torch.empty((H, N, D), device=dev)
        etm_lib.check(lib.etm_window_bwd(spec.block_ptr(block), spec.ep_stride, spec.row_stride, ep.data_ptr(), win.data_ptr(), None, m8.data_ptr(),
                                         None, ln_g.data_ptr(), ln_b.data_ptr(), gathered.data_ptr(), a2.data_ptr(), gz.data_ptr(), N * D, D,
                                         d_e2.data_ptr(), du2.data_ptr(), N * D, D, N, L, D, H, st), "etm_window_bwd")
        dY2 = torch.einsum("nhl,hnd->nld", d_e2.double(), u.double()) + torch.einsum("nhl,hnd->nld", a2.double(), gz.double())
        want2 = torch.cat(((dY2 * xhat).sum((0, 1)), dY2.sum((0, 1))))
        p_rows, p_out = torch.full((rows, 2 * D), float("nan"), device=dev), torch.full((rows, 2 * D), float("nan"), device=dev)
        etm_lib.check(lib.etm_window_ln_grad(spec.block_ptr(block), spec.ep_stride, spec.row_stride, ep.data_ptr(), win.data_ptr(), None, None,
                                             gathered.data_ptr(), a2.data_ptr(), d_e2.data_ptr(), u.data_ptr(), gz.data_ptr(), N * D, D,
                                             p_rows.data_ptr(), N, L, D, H, st), "etm_window_ln_grad")
        etm_lib.check(lib.etm_window_ln_grad_from_outputs(u.data_ptr(), gz.data_ptr(), du2.data_ptr(), z2.data_ptr(), a2.data_ptr(), d_e2.data_ptr(),
                                                          ln_g.data_ptr(), ln_b.data_ptr(), N * D, D, p_out.data_ptr(), N, L, D, H, st),
                      "etm_window_ln_grad_from_outputs")
        e_rows = float((p_rows.double().sum(0) - want2).norm() / want2.norm())
        e_out_g = float((p_out.double().sum(0)[:D] - want2[:D]).norm() / want2[:D].norm())
        e_out_b = float((p_out.double().sum(0)[D:] - want2[D:]).norm() / want2[D:].norm())
        print(f"[norm_kv gradients N={N} L={L} D={D} H={H}] vs float64: rows kernel {e_rows:.1e}; from outputs: gain {e_out_g:.1e}, bias {e_out_b:.1e} "
              f"(smallest |gain| {float(ln_g.abs().min()):.1e})")
        # (random N(0,1) gains: the smallest of 128 - 384 is ~1e-2 .. 1e-3, and the identity divides by it -- the error of that ONE column
        # is what the norm-wise bound of the gain sees; LayerNorm gains in training sit near 1)
        assert e_rows < 2e-6 and e_out_b < 2e-6 and e_out_g < 2e-4, (N, L, D, H, e_rows, e_out_g, e_out_b)
        good = ln_g.abs() > 0.2
        e_good = float(((p_out.double().sum(0)[:D] - want2[:D])[good]).norm() / want2[:D][good].norm())
        assert e_good < 5e-6, e_good


def test_rollout_glue_riders_and_fused_policy():
    """rollout_window's riders (t_row, cache reset) == the stand-alone ops; rollout_policy == rollout_heads + rollout_sample
    (bit-exact), including the pinned-memory hand-over; conv_relu with a device-side row index == conv_relu on that row."""
    from etm import ops
    from trainer import build_window_tables
    dev = _dev()
    torch.manual_seed(0)
    W, L, T, S, A, hid = 7, 8, 12, 5, 3, 64
    mask_tab, idx_tab = build_window_tables(L, T)
    mask_tab, idx_tab = mask_tab.to(dev).bool(), idx_tab.to(dev)
    step = torch.tensor([0, 3, 11, 0, 7, 1, 0], device=dev)
    t_dev = torch.tensor(2, dtype=torch.int64, device=dev)
    outs = []
    for riders in (False, True):
        mask_t = torch.zeros((W, L), dtype=torch.bool, device=dev); win_t = torch.zeros((W, L), dtype=torch.int64, device=dev)
        st_m = torch.zeros((S, W, L), dtype=torch.bool, device=dev); st_i = torch.zeros((S, W, L), dtype=torch.int64, device=dev)
        cache = torch.randn((W, 4, 2, 16), device=dev); init = torch.randn((4, 2, 16), device=dev)
        torch.manual_seed(1); cache.copy_(torch.randn_like(cache)); init.copy_(torch.randn_like(init))
        t_row = torch.full((), -1, dtype=torch.int64, device=dev)
        if riders:
            ops.rollout_window(step, mask_tab, idx_tab, t_dev, mask_t, win_t, st_m, st_i, t_row=t_row, reset=(cache, init))
        else:
            ops.rollout_window(step, mask_tab, idx_tab, t_dev, mask_t, win_t, st_m, st_i)
            ops.reset_rows(cache, init, step)
            t_row.copy_(t_dev)
        outs.append((mask_t, win_t, st_m, st_i, cache, t_row))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert int(outs[1][5]) == 2 and torch.equal(outs[1][4][0], outs[1][4][3])

    lin_p, lin_v = torch.nn.Linear(hid, A).to(dev), torch.nn.Linear(hid, 1).to(dev)
    h2 = torch.randn((W, 2 * hid), device=dev)
    uni = torch.rand((S, W), device=dev)
    res = []
    for fused in (False, True):
        t = torch.tensor(1, dtype=torch.int64, device=dev)
        act = torch.zeros((W, 1), dtype=torch.int64, device=dev)
        sa = torch.zeros((S, W, 1), dtype=torch.int64, device=dev); sl = torch.zeros((S, W, 1), device=dev); sv = torch.zeros((S, W), device=dev)
        hp = torch.full((W, 1), -1, dtype=torch.int64).pin_memory(); hf = torch.zeros((1,), dtype=torch.int64).pin_memory()
        if fused:
            ops.rollout_policy(h2, lin_p, lin_v, uni, None, t, act, sa, sl, sv, host_actions=hp, host_flag=hf)
        else:
            logits, value = ops.rollout_heads(h2, lin_p, lin_v)
            ops.rollout_sample(logits, value, uni, None, t, act, sa, sl, sv)
        torch.cuda.synchronize()
        res.append((act, sa, sl, sv, t))
        if fused:
            assert int(hf[0]) == 2 and torch.equal(hp, act.cpu())
    for a, b in zip(*res):
        assert torch.equal(a, b)

    # folded epilogues: add_layernorm(act(a + bias) + b) and rollout_policy on pre-activations
    with torch.no_grad():
        a_, b_ = torch.randn((W, 96), device=dev), torch.randn((W, 96), device=dev)
        norm, bias = torch.nn.LayerNorm(96).to(dev), torch.randn(96, device=dev)
        norm.weight.add_(0.1 * torch.randn_like(norm.weight))
        got = ops.add_layernorm(a_, b_, norm, bias=bias, relu=True)
        close(got, norm(torch.relu(a_ + bias) + b_).cpu().numpy(), atol=2e-5, rtol=1e-4, what="add_layernorm with folded bias + relu")
        hb = torch.randn(2 * hid, device=dev)
        raw = torch.randn((W, 2 * hid), device=dev)
        outs2 = []
        for folded in (False, True):
            t = torch.tensor(1, dtype=torch.int64, device=dev)
            act = torch.zeros((W, 1), dtype=torch.int64, device=dev)
            sa = torch.zeros((S, W, 1), dtype=torch.int64, device=dev); sl = torch.zeros((S, W, 1), device=dev); sv = torch.zeros((S, W), device=dev)
            if folded:
                ops.rollout_policy(raw, lin_p, lin_v, uni, None, t, act, sa, sl, sv, h_bias=hb)
            else:
                ops.rollout_policy(torch.relu(raw + hb), lin_p, lin_v, uni, None, t, act, sa, sl, sv)
            outs2.append((act, sa, sl, sv))
        for x_, y_ in zip(*outs2):
            assert torch.equal(x_, y_)

    stack = torch.rand((3, 5, 3, 36, 36), device=dev)
    conv = torch.nn.Conv2d(3, 32, 8, 4).to(dev)
    wp = ops.conv_pack_weights(conv.weight.detach().reshape(32, -1))
    idx = torch.tensor(2, dtype=torch.int64, device=dev)
    a = ops.conv_relu(stack, wp, conv.bias.detach(), 3, 36, 36, 8, 8, 4, False, False, index=idx)
    b = ops.conv_relu(stack[2].contiguous(), wp, conv.bias.detach(), 3, 36, 36, 8, 8, 4, False, False)
    assert torch.equal(a, b)
    want = torch.relu(conv(stack[2])).permute(0, 2, 3, 1)
    close(a, want.detach().cpu().numpy(), atol=2e-5, rtol=1e-4, what="conv_relu vs library")


# ------------------------------------------------------------------ other BASELINE config shapes + RCCL plumbing on one device
@pytest.mark.parametrize("cfg_name,over", [
    ("synthetic_cartpole", dict(n_workers=8, worker_steps=64, n_mini_batch=2, epochs=1)),              # config (2): GTrXL, pre-LN, D=128 H=1
    ("synthetic_mortar_gtrxl", dict(n_workers=4, worker_steps=160, n_mini_batch=2, epochs=1)),         # config (5): GTrXL 4 blocks, L=128, pre-LN
    ("synthetic_minigrid", dict(n_workers=4, worker_steps=128, n_mini_batch=2, epochs=1)),             # config (3)
])
def test_baseline_config_shapes_train(cfg_name, over):
    """One rollout + optimisation pass at the model shapes of BASELINE configs (2), (3), (5) (fewer workers/steps):
    the buffer self-consistency check (quirk Q10) must hold and every parameter must receive a finite gradient."""
    from yaml_parser import YamlParser
    from trainer import PPOTrainer
    from etm.ops import WindowSpec
    dev = _dev()
    here = os.path.dirname(os.path.abspath(__file__))
    cfg = YamlParser(os.path.join(here, "..", "episodic-transformer-memory-ppo_amd", "configs", cfg_name + ".yaml")).get_config()
    cfg.update(over)
    cfg["environment"] = dict(cfg["environment"], pool=4)
    torch.manual_seed(0)
    tr = PPOTrainer(cfg, run_id="shapes", device=dev, tensorboard=False)
    tr._sample_training_data()
    tr.buffer.prepare_batch_dict()
    flat = tr.buffer.samples_flat
    with torch.no_grad():
        spec = WindowSpec.from_bank(tr.buffer.memories, flat["memory_index"], flat["memory_indices"], flat["memory_indices"], flat["memory_mask"])
        _, value, _ = tr.model.forward_logits(flat["obs"], spec)
    keep = flat["memory_mask"].any(dim=1)
    assert torch.allclose(value[keep], flat["values"][keep], atol=5e-4), (value[keep] - flat["values"][keep]).abs().max()
    stats, _ = tr._train_epochs(1e-4, 0.1, 1e-3)
    assert np.isfinite(np.asarray(stats)).all()
    for name, p_ in tr.model.named_parameters():
        assert torch.isfinite(p_.grad).all(), name
        assert torch.isfinite(p_).all(), name
    tr.close()


def test_rccl_single_rank_plumbing(tmp_path):
    """The data-parallel path with the real RCCL backend on one device (world size 1 forced): communicator creation,
    flat-bucket all-reduce, merged advantage statistics, rank-max -- what every rank does in the multi-GPU bench."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(here, "..", "episodic-transformer-memory-ppo_amd")
    code = f"""
import os, sys
sys.path.insert(0, {pkg!r})
import torch, torch.distributed as dist
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
from etm.dist import DataParallel
dp = DataParallel(dev)
dp.world = 1
flat = torch.arange(1000, dtype=torch.float32, device=dev)
dist.all_reduce(flat)                       # RCCL all-reduce on the device
assert float(flat[999]) == 999.0
g = [torch.empty(3, device=dev)]
dist.all_gather(g, torch.tensor([4.0, 1.0, 2.0], device=dev))
assert g[0].tolist() == [4.0, 1.0, 2.0]
dist.barrier()
dist.destroy_process_group()
print("rccl-ok")
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert "rccl-ok" in out.stdout, out.stderr[-2000:]


def test_candidate_library_collective_single_rank():
    """etm_comm_* / etm_allreduce_f32 on one device (world size 1): the sum over one rank is the identity, enqueued on the
    caller's stream, in place and out of place."""
    import ctypes
    from etm import lib as etm_lib
    dev = _dev()
    lib = etm_lib.load()
    buf = ctypes.create_string_buffer(128)
    etm_lib.check(lib.etm_comm_unique_id(buf), "etm_comm_unique_id")
    comm = ctypes.c_void_p()
    with torch.cuda.device(dev):
        etm_lib.check(lib.etm_comm_init(bytes(buf.raw), 0, 1, ctypes.byref(comm)), "etm_comm_init")
    x = torch.randn(1 << 20, device=dev)
    ref = x.clone()
    y = torch.empty_like(x)
    st = torch.cuda.current_stream(dev).cuda_stream
    etm_lib.check(lib.etm_allreduce_f32(comm, x.data_ptr(), y.data_ptr(), x.numel(), st), "etm_allreduce_f32")
    etm_lib.check(lib.etm_allreduce_f32(comm, x.data_ptr(), x.data_ptr(), x.numel(), st), "etm_allreduce_f32")
    torch.cuda.synchronize(dev)
    assert torch.equal(y, ref) and torch.equal(x, ref)
    etm_lib.check(lib.etm_comm_destroy(comm), "etm_comm_destroy")


def test_graph_and_eager_optimisation_steps_agree():
    """The captured minibatch step (device-resident lr / clip / beta under changing schedules) trains exactly like the eager
    `_train_mini_batch` path that mirrors upstream's method (hip_graph_train: false), and the upstream-style generator API
    still yields usable minibatches."""
    from trainer import PPOTrainer
    dev = _dev()
    cfg = dict(environment=dict(type="Synthetic", obs_shape=[3, 36, 36], num_actions=3, max_episode_steps=20, seed=7, p_done=0.07, pool=8),
               gamma=0.99, lamda=0.95, updates=1, epochs=2, n_workers=6, worker_steps=32, n_mini_batch=4, value_loss_coefficient=0.5,
               hidden_layer_size=64, max_grad_norm=0.5,
               transformer=dict(num_blocks=2, embed_dim=64, num_heads=2, memory_length=8, positional_encoding="relative",
                                layer_norm="pre", gtrxl=True, gtrxl_bias=0.0),
               learning_rate_schedule=dict(initial=3e-4, final=1e-4, power=1.0, max_decay_steps=3),
               beta_schedule=dict(initial=1e-3, final=1e-4, power=1.0, max_decay_steps=3),
               clip_range_schedule=dict(initial=0.2, final=0.1, power=1.0, max_decay_steps=3))
    rng = np.random.default_rng(1)
    acts = rng.integers(0, 3, size=(3, 6, 32))
    perms = [[rng.permutation(6 * 32) for _ in range(2)] for _ in range(3)]
    params, stats = [], []
    for graph in (True, False):
        c = json.loads(json.dumps(cfg))
        c["hip_graph_train"] = graph
        torch.manual_seed(9)
        tr = PPOTrainer(c, run_id="ge", device=dev, tensorboard=False)
        rows = []
        for u in range(3):
            lr, beta, clip = tr.schedules(u)
            tr._sample_training_data(forced_actions=acts[u])
            tr.buffer.prepare_batch_dict()
            info, _ = tr._train_epochs(lr, clip, beta, perms=perms[u])
            rows.append(np.asarray(info))
        assert (tr._train_graph is not None) == graph
        params.append([p.detach().clone() for p in tr.model.parameters()])
        stats.append(np.concatenate(rows))
        if not graph:     # upstream-style API: generator + one more eager step
            mb = next(iter(tr.buffer.mini_batch_generator(perms[0][0])))
            assert set(mb) >= {"actions", "values", "log_probs", "advantages", "obs", "memory_mask", "memory_indices", "memories"}
            st = tr._train_mini_batch(mb, 1e-4, 0.1, 1e-4)
            assert st.shape == (6,) and bool(torch.isfinite(st).all())
        tr.close()
    assert np.allclose(stats[0], stats[1], atol=1e-5, rtol=1e-4), np.abs(stats[0] - stats[1]).max()
    for a, b in zip(*params):
        assert torch.allclose(a, b, atol=1e-6, rtol=1e-5), float((a - b).abs().max())


def test_dp_graph_step_single_rank(tmp_path):
    """Data-parallel optimisation step on one device with the collectives really issued (world size 1, `active` forced), in both
    forms: ONE graph holding the library's RCCL all-reduce between backward and clip + AdamW (round 6, the default with the library
    collective) and graph A -> all-reduce as a host call -> graph B (torch's collective, or `dp_graph_collective: false`); advantage
    statistics merged with an all-gather.  Both must train exactly like the single-GPU captured step on the same seeds, and the
    two data-parallel forms bit-identically."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(here, "..", "episodic-transformer-memory-ppo_amd")
    code = f"""
import os, sys, json
sys.path.insert(0, {pkg!r})
import numpy as np, torch, torch.distributed as dist
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29633")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
from etm.dist import DataParallel
from trainer import PPOTrainer

class ForcedDP(DataParallel):
    @property
    def active(self):
        return True

cfg = dict(environment=dict(type="Synthetic", obs_shape=[6], num_actions=3, max_episode_steps=24, seed=2, p_done=0.06, pool=8),
           gamma=0.99, lamda=0.95, updates=1, epochs=2, n_workers=6, worker_steps=48, n_mini_batch=4, value_loss_coefficient=0.5,
           hidden_layer_size=64, max_grad_norm=0.5,
           transformer=dict(num_blocks=2, embed_dim=64, num_heads=2, memory_length=16, positional_encoding="relative",
                            layer_norm="post", gtrxl=False, gtrxl_bias=0.0),
           learning_rate_schedule=dict(initial=3e-4, final=1e-4, power=1.0, max_decay_steps=4),
           beta_schedule=dict(initial=1e-3, final=1e-4, power=1.0, max_decay_steps=4),
           clip_range_schedule=dict(initial=0.2, final=0.1, power=1.0, max_decay_steps=4))
rng = np.random.default_rng(0)
acts = rng.integers(0, 3, size=(3, 6, 48))
perms = [[rng.permutation(6 * 48) for _ in range(2)] for _ in range(3)]
res = []
for mode in ("single", "one_graph", "three_call", "torch_collective"):
    dp = ForcedDP(dev, collective="torch" if mode == "torch_collective" else "etm") if mode != "single" else None
    torch.manual_seed(5)
    c = json.loads(json.dumps(cfg))
    c["dp_graph_collective"] = mode != "three_call"
    tr = PPOTrainer(c, run_id="dpg", device=dev, dp=dp, tensorboard=False)
    for u in range(3):
        lr, beta, clip = tr.schedules(u)
        tr._sample_training_data(forced_actions=acts[u])
        tr.buffer.prepare_batch_dict()
        tr._train_epochs(lr, clip, beta, perms=perms[u])
    assert tr._train_graph is not None and (tr._train_graph[1] is not None) == (mode in ("three_call", "torch_collective")), mode
    assert bool(getattr(tr, "_dp_one_graph", False)) == (mode == "one_graph"), mode
    res.append([p.detach().clone() for p in tr.model.parameters()])
    tr.close()
for other in res[1:]:
    for a, b in zip(res[0], other):
        assert torch.allclose(a, b, atol=1e-6, rtol=1e-5), float((a - b).abs().max())
for other in res[2:]:
    for a, b in zip(res[1], other):
        assert torch.equal(a, b), float((a - b).abs().max())
dist.destroy_process_group()
print("dp-graph-ok")
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "dp-graph-ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_dp_overlapped_step_single_rank(tmp_path):
    """dp_overlap (round 4): the backward pass cut at the encoder output, the all-reduce of the head / transformer / lin_hidden slice
    on a side stream under the encoder's backward pass, the convolution slice after it -- on one device with the collectives really
    issued (world size 1, library RCCL communicator), as host calls between graph replays and (round 6) captured with the whole step
    in ONE graph (with and without the overlap): parameters BIT-IDENTICAL to the non-overlapped three-call data-parallel step (the
    same kernels in the same order per stream; only the all-reduce is split / captured) and equal to the single-GPU step within rounding."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(here, "..", "episodic-transformer-memory-ppo_amd")
    code = f"""
import os, sys, json
sys.path.insert(0, {pkg!r})
import numpy as np, torch, torch.distributed as dist
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
from etm.dist import DataParallel
from trainer import PPOTrainer

class ForcedDP(DataParallel):
    @property
    def active(self):
        return True

cfg = dict(environment=dict(type="Synthetic", obs_shape=[3, 36, 36], num_actions=3, max_episode_steps=24, seed=2, p_done=0.06, pool=8),
           gamma=0.99, lamda=0.95, updates=1, epochs=2, n_workers=8, worker_steps=48, n_mini_batch=4, value_loss_coefficient=0.5,
           hidden_layer_size=64, max_grad_norm=0.5,
           transformer=dict(num_blocks=2, embed_dim=64, num_heads=2, memory_length=16, positional_encoding="relative",
                            layer_norm="post", gtrxl=False, gtrxl_bias=0.0),
           learning_rate_schedule=dict(initial=3e-4, final=1e-4, power=1.0, max_decay_steps=4),
           beta_schedule=dict(initial=1e-3, final=1e-4, power=1.0, max_decay_steps=4),
           clip_range_schedule=dict(initial=0.2, final=0.1, power=1.0, max_decay_steps=4))
rng = np.random.default_rng(0)
acts = rng.integers(0, 3, size=(3, 8, 48))
perms = [[rng.permutation(8 * 48) for _ in range(2)] for _ in range(3)]
res = {{}}
for mode in ("single", "dp", "dp_overlap", "dp_one_graph", "dp_overlap_one_graph"):
    dp = ForcedDP(dev, collective="etm") if mode != "single" else None
    torch.manual_seed(5)
    c = json.loads(json.dumps(cfg))
    c["dp_overlap"] = "overlap" in mode
    c["dp_graph_collective"] = "one_graph" in mode
    tr = PPOTrainer(c, run_id="dpo", device=dev, dp=dp, tensorboard=False)
    for u in range(3):
        lr, beta, clip = tr.schedules(u)
        tr._sample_training_data(forced_actions=acts[u])
        tr.buffer.prepare_batch_dict()
        tr._train_epochs(lr, clip, beta, perms=perms[u])
    assert tr._train_graph is not None and tr.model._train_encoder_ok
    assert (getattr(tr, "_train_graph_a2", None) is not None) == (mode == "dp_overlap"), mode
    assert bool(getattr(tr, "_dp_one_graph", False)) == ("one_graph" in mode), mode
    res[mode] = [p.detach().clone() for p in tr.model.parameters()]
    tr.close()
for other in ("dp_overlap", "dp_one_graph", "dp_overlap_one_graph"):      # same kernels in the same order per stream: bit-identical parameters
    for a, b in zip(res["dp"], res[other]):
        assert torch.equal(a, b), (other, float((a - b).abs().max()))
for a, b in zip(res["single"], res["dp_overlap"]):
    assert torch.allclose(a, b, atol=1e-6, rtol=1e-5), float((a - b).abs().max())
dist.destroy_process_group()
print("dp-overlap-ok")
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "dp-overlap-ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_train_cli_and_checkpoint_format(tmp_path):
    """`train.py --config ... --run-id ...` runs end to end and writes upstream's checkpoint format:
    pickle((state_dict, config)) at ./models/<run_id>.nn with the reference's key names (trainer.py:356-362, enjoy.py:48-55)."""
    import pickle
    import subprocess
    import sys
    import yaml
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(here, "..", "episodic-transformer-memory-ppo_amd")
    cfg = yaml.safe_load(open(os.path.join(pkg, "configs", "synthetic_cartpole.yaml")))
    cfg.update(updates=2, n_workers=4, worker_steps=32, epochs=1, n_mini_batch=2)
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    out = subprocess.run([sys.executable, os.path.join(pkg, "train.py"), "--config", str(cfg_path), "--run-id", "clitest"],
                         cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "steps/s" in out.stdout and "Model saved" in out.stdout
    state, saved_cfg = pickle.load(open(tmp_path / "models" / "clitest.nn", "rb"))
    assert saved_cfg["transformer"] == cfg["transformer"]
    assert "transformer.transformer_blocks.0.attention.keys.weight" in state and "policy_branches.0.weight" in state
    assert all(v.device.type == "cpu" for v in state.values())
    bad = subprocess.run([sys.executable, os.path.join(pkg, "train.py"), "--config", str(cfg_path), "--cpu"], cwd=tmp_path,
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "no CPU trainer" in (bad.stderr + bad.stdout)
    # enjoy.py (upstream enjoy.py:48-96) loads that checkpoint and plays one episode with the model on the device
    play = subprocess.run([sys.executable, os.path.join(pkg, "enjoy.py"), "--model", str(tmp_path / "models" / "clitest.nn")], cwd=tmp_path,
                          capture_output=True, text=True, timeout=300)
    assert play.returncode == 0, play.stderr[-2000:]
    assert "Episode length:" in play.stdout and "Episode reward:" in play.stdout


@pytest.mark.parametrize("N", [2048, 2560, 130, 37, 3])
def test_grouped_weight_gradients_vs_float64(N):
    """etm_grouped_dw: dW = dy^T x of several layers in one launch -- plain [384, 384] layers, the per-head folds (the head's columns
    of A, its plane of B, its rows of C) and a [96, 128] corner case -- against the float64 products; odd / tiny N exercise the
    zero-filled tails of the four k-ranges.  Also through the autograd functions with a DeferredDw collector (what the trainer does)."""
    import ctypes
    from etm import lib as etm_lib
    from etm import ops
    dev = _dev()
    lib = etm_lib.load()
    torch.manual_seed(N)
    D, H = 384, 4
    hd = D // H
    dy1, x1 = torch.randn((N, D), device=dev), torch.randn((N, D), device=dev)
    q, du = torch.randn((N, D), device=dev), torch.randn((H, N, D), device=dev)
    dy3, x3 = torch.randn((N, 96), device=dev), torch.randn((N, 128), device=dev)
    c1, c2, c3 = (torch.full(shape, float("nan"), device=dev) for shape in ((D, D), (D, D), (96, 128)))
    probs = [(dy1, 0, x1, c1, D, D)] + [(q, h * hd, du[h], c2[h * hd:(h + 1) * hd], hd, D) for h in range(H)] + [(dy3, 0, x3, c3, 96, 128)]
    k = len(probs)
    pa = (ctypes.c_void_p * k)(*[a.data_ptr() + 4 * off for a, off, *_ in probs])
    pb = (ctypes.c_void_p * k)(*[b.data_ptr() for _, _, b, *_ in probs])
    pc = (ctypes.c_void_p * k)(*[c.data_ptr() for _, _, _, c, *_ in probs])
    dims = (ctypes.c_int32 * (5 * k))(*[v for a, _, b, c, ma, nb in probs for v in (ma, nb, a.stride(0), b.stride(0), c.stride(0))])
    etm_lib.check(lib.etm_grouped_dw(pa, pb, pc, dims, k, N, torch.cuda.current_stream(dev).cuda_stream), "etm_grouped_dw")
    want1 = dy1.double().t() @ x1.double()
    want2 = torch.cat([q[:, h * hd:(h + 1) * hd].double().t() @ du[h].double() for h in range(H)])
    want3 = dy3.double().t() @ x3.double()
    worst = 0.0
    for got, want in ((c1, want1), (c2, want2), (c3, want3)):
        rel = float((got.double() - want).abs().max() / want.norm() * want.numel() ** 0.5)     # element error / rms element
        worst = max(worst, rel)
        assert rel < 2e-5, rel
    # the same problems through autograd: linear (no bias), head fold / unfold, linear + ReLU, collected and flushed by DeferredDw
    w = [torch.randn((D, D), device=dev).mul_(0.05).requires_grad_(True) for _ in range(4)]
    b = torch.zeros(D, device=dev, requires_grad=True)
    xin = torch.randn((N, D), device=dev)

    def net():
        h1 = ops.linear_nobias(xin, w[0])
        u = ops._HeadFoldFn.apply(h1, w[1], H)                       # [H, N, D]
        c = ops._HeadUnfoldFn.apply(torch.tanh(u), w[2], H)
        return ops.linear_relu_train(c, w[3], b)

    gout = torch.randn((N, D), device=dev)
    (net() * gout).sum().backward()
    ref = [t.grad.clone() for t in w]
    for t in w:
        t.grad = None
    views = [torch.full((D, D), float("nan"), device=dev) for _ in w]
    with ops.DeferredDw({t.data_ptr(): v for t, v in zip(w, views)}) as col:
        (net() * gout).sum().backward()
    assert col.written == {t.data_ptr() for t in w} and all(t.grad is None for t in w)
    for v, r in zip(views, ref):
        assert float((v - r).abs().max()) <= 2e-5 * float(r.abs().max()) + 1e-6, float((v - r).abs().max())
    print(f"[grouped dW N={N}] worst element error / rms element vs float64: {worst:.2e}")


@pytest.mark.parametrize("N,D", [(2048, 384), (601, 128), (9, 96), (1, 64)])
def test_grouped_column_sums_bit_identical_to_per_call_reductions(N, D):
    """etm_colsum_reduce_grouped: the LayerNorm weight / bias and linear bias gradients of a backward pass (two fused LayerNorms --
    with bias + ReLU + residual, and plain -- and a linear + ReLU layer) reduced by the DeferredDw collector's ONE launch into 1-D
    destination views, against the per-call reductions (same summation tree: bit-identical) and float64 column sums."""
    from etm import ops
    dev = _dev()
    torch.manual_seed(N * 7 + D)
    n1, n2 = torch.nn.LayerNorm(D).to(dev), torch.nn.LayerNorm(D).to(dev)
    lin = torch.nn.Linear(D, D).to(dev)
    fcb = torch.randn(D, device=dev).mul_(0.1).requires_grad_(True)
    with torch.no_grad():
        for m in (n1, n2):
            m.weight.add_(torch.randn(D, device=dev) * 0.1)
            m.bias.add_(torch.randn(D, device=dev) * 0.1)
    a, res = torch.randn((N, D), device=dev, requires_grad=True), torch.randn((N, D), device=dev, requires_grad=True)
    gout = torch.randn((N, D), device=dev)
    params = [n1.weight, n1.bias, n2.weight, n2.bias, lin.bias, fcb]

    def net():
        x = ops.fused_layernorm(a, n1, bias=fcb, res=res, relu=True)
        y = ops.linear_relu_train(x, lin.weight, lin.bias)
        return ops.fused_layernorm(y, n2)

    (net() * gout).sum().backward()
    ref = [t.grad.clone() for t in params]
    ref_in = (a.grad.clone(), res.grad.clone(), lin.weight.grad.clone())
    for t in params + [a, res, lin.weight]:
        t.grad = None
    views = [torch.full((D,), float("nan"), device=dev) for _ in params]
    with ops.DeferredDw({t.data_ptr(): v for t, v in zip(params, views)}) as col:
        (net() * gout).sum().backward()
    assert col.written == {t.data_ptr() for t in params} and all(t.grad is None for t in params)
    for v, r in zip(views, ref):
        assert torch.equal(v, r), float((v - r).abs().max())
    for got, want in zip((a.grad, res.grad, lin.weight.grad), ref_in):
        assert torch.equal(got, want)
    # float64: d n2.bias = column sums of the upstream gradient
    want = gout.double().sum(0)
    assert float((views[3].double() - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max())) * max(1.0, N ** 0.5)


def test_rollout_driver_service_order():
    """csrc/rollout_driver.hip without a trainer: two fake worker groups (their `ready` words published by a thread of this test,
    group 1 BEFORE group 0 in every step).  Group 1 is served while group 0 is still stepping -- unless it has an episode end in
    that step, then it waits for group 0, so that the slot numbers (upstream trainer.py:211) come out in (step, group) order.
    Step counters, slots and the event list equal the Python loop the driver replaces."""
    import ctypes
    import threading
    import time
    from etm import lib as etm_lib
    if not hasattr(torch.cuda.CUDAGraph, "raw_cuda_graph_exec"):
        pytest.skip("needs CUDAGraph.raw_cuda_graph_exec")
    dev = _dev()
    lib = etm_lib.load()
    G, Wg, S, row = 2, 2, 8, 16
    W = G * Wg
    rng = np.random.default_rng(11)
    dones = (rng.random((S, W)) < 0.3).astype(np.uint8)
    dones[1, 2] = 1; dones[1, :2] = 0            # step 1: group 1 has an episode end, group 0 has none
    dones[2, :] = 0                              # step 2: nobody
    dones[3, 0] = dones[3, 3] = 1                # step 3: both
    ready = np.zeros((G, 8), dtype=np.int64)     # one cache line per process
    ss = [np.zeros((2, Wg), dtype=np.int64) for _ in range(G)]
    obs = np.zeros((W, row // 4), dtype=np.float32)
    stage = torch.zeros((S + 1, W, row // 4), device=dev)
    xs = [torch.zeros(8, device=dev) for _ in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    graphs = []
    for st, x in zip(streams, xs):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            x.add_(1.0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            x.add_(1.0)
        graphs.append(g)
    torch.cuda.synchronize()
    arr = (etm_lib.RolloutGroup * G)()
    for gi in range(G):
        a = arr[gi]
        a.graph_exec, a.stream = graphs[gi].raw_cuda_graph_exec(), streams[gi].cuda_stream
        a.ready, a.n_procs, a.ready_stride = ready[gi:].ctypes.data, 1, ready.shape[1]
        a.lo, a.hi = gi * Wg, (gi + 1) * Wg
        a.obs_src, a.stage_dst, a.ss_dst = obs[gi * Wg:].ctypes.data, stage.data_ptr() + gi * Wg * row, ss[gi].ctypes.data
    ep_step, slot = np.zeros(W, dtype=np.int64), np.arange(W, dtype=np.int64)
    ctr = np.array([W, 0], dtype=np.int64)
    events = np.zeros((W * S, 3), dtype=np.int64)
    abort = np.zeros((1, 8), dtype=np.int64)
    served_first = []                            # per step: was group 1 served while group 0 had not published?

    def workers():
        for t in range(S):
            ss[0][0, 0] = ss[1][0, 0] = -1                      # (episode steps are >= 0: the driver's write of step t + 1's words shows)
            ready[1, 0] = t + 1
            seen, t0 = False, time.time()
            while t + 1 < S and time.time() - t0 < 0.25 and not seen:
                seen = int(ss[1][0, 0]) != -1                   # the driver wrote group 1's (step, slot) words of step t + 1
                time.sleep(0.002)
            served_first.append(seen)
            ready[0, 0] = t + 1
            t0 = time.time()       # (real workers cannot publish step t + 1 before the driver has launched it: wait for both groups' words)
            while t + 1 < S and time.time() - t0 < 10.0 and (int(ss[0][0, 0]) == -1 or int(ss[1][0, 0]) == -1):
                time.sleep(0.001)

    th = threading.Thread(target=workers)
    th.start()
    try:
        rc = lib.etm_rollout_drive(ctypes.cast(arr, ctypes.c_void_p), G, 0, S, W, row, W * row, dones.ctypes.data, ep_step.ctypes.data,
                                   slot.ctypes.data, ctr.ctypes.data, 1000, events.ctypes.data, events.shape[0], ctr[1:].ctypes.data,
                                   abort.ctypes.data, 1, abort.shape[1], 20.0, None, None)
    finally:
        th.join()
    torch.cuda.synchronize()
    assert rc == 0
    # the loop it replaces (upstream trainer.py:195-213 in (step, group) order)
    e_ref, s_ref, nxt, ev = np.zeros(W, dtype=np.int64), np.arange(W, dtype=np.int64), W, []
    for t in range(S):
        for w in range(W):
            if dones[t, w]:
                e_ref[w], s_ref[w] = 0, nxt
                ev.append((t, w, nxt))
                nxt += 1
            else:
                e_ref[w] += 1
    assert np.array_equal(ep_step, e_ref) and np.array_equal(slot, s_ref) and ctr[0] == nxt and ctr[1] == len(ev)
    assert np.array_equal(events[: len(ev)], np.asarray(ev, dtype=np.int64).reshape(-1, 3))
    assert all(float(x[0]) == S for x in xs)                    # every group's step graph was launched S - 1 times (+ 1 warm-up)
    for t in range(S - 1):
        assert served_first[t] == (not dones[t, Wg:].any()), (t, served_first, dones[t])


@pytest.mark.parametrize("layout", ["trxl_post", "gtrxl_pre"])
def test_every_replay_of_the_captured_step_equals_its_eager_evaluation(layout):
    """tools/graph_replay_soak.py, short: 60 updates = 238 replays of the captured optimisation step, each compared tensor by tensor
    with an eager evaluation of the same minibatch at the same parameters.  A node that is not replay-safe (round 5: torch's column
    sum) shows as a tensor that is wrong in some replays; rounding-level otherwise (measured <= 2.1e-7)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import graph_replay_soak as soak
    r = soak.run(layout, 60, quiet=True)
    assert r["replays"] >= 230
    assert r["loud"] == 0 and r["worst"] < 1e-5, r


@pytest.mark.parametrize("N,D", [(2048, 384), (601, 128)])
def test_linear_bias_and_forked_gate_vs_plain_autograd(N, D):
    """Round 5: ops.linear_bias (fc_out of the pre-LN / gated blocks: bias gradient by the library's column sums, weight gradient
    deferrable) against nn.Linear under plain autograd in float64, with and without a DeferredDw collector (bit-identical to each
    other); ops.gru_gate_train(fork=True) (the output twice, gradients added by gate_bwd1 on load) against the un-forked gate fed
    the sum of the two gradients (bit-identical)."""
    from etm import ops
    import transformer as tfm
    dev = _dev()
    torch.manual_seed(N + D)
    lin = torch.nn.Linear(D, D).to(dev)
    x = torch.randn((N, D), device=dev, requires_grad=True)
    g = torch.randn((N, D), device=dev)
    (ops.linear_bias(lin, x) * g).sum().backward()
    got = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    lin64 = torch.nn.Linear(D, D).to(dev).double()
    lin64.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
    x64 = x.detach().double().requires_grad_(True)
    (lin64(x64) * g.double()).sum().backward()
    for a, b, what in zip(got, (x64.grad, lin64.weight.grad, lin64.bias.grad), ("dx", "dW", "db")):
        err = float((a.double() - b).norm() / b.norm())
        assert err < 2e-6, (what, err)
    # through a collector: the same bits (the grouped launches use the per-call summation trees)
    x.grad = lin.weight.grad = lin.bias.grad = None
    views = {lin.weight.data_ptr(): torch.full_like(lin.weight, float("nan")), lin.bias.data_ptr(): torch.full_like(lin.bias, float("nan"))}
    with ops.DeferredDw(views) as col:
        (ops.linear_bias(lin, x) * g).sum().backward()
    assert lin.bias.data_ptr() in col.written and lin.bias.grad is None
    assert torch.equal(views[lin.bias.data_ptr()], got[2]) and torch.equal(x.grad, got[0])
    dw = views[lin.weight.data_ptr()] if lin.weight.data_ptr() in col.written else lin.weight.grad
    assert float((dw.double() - lin64.weight.grad).norm() / lin64.weight.grad.norm()) < 2e-6

    gate = tfm.GRUGate(D, 0.1).to(dev)
    xg = torch.randn((N, D), device=dev, requires_grad=True)
    yg = torch.randn((N, D), device=dev, requires_grad=True)
    g1, g2 = torch.randn((N, D), device=dev), torch.randn((N, D), device=dev)
    params = [xg, yg] + list(gate.parameters())
    out = ops.gru_gate_train(gate, xg, yg)
    (out * (g1 + g2)).sum().backward()
    ref = [t.grad.clone() for t in params]
    for t in params:
        t.grad = None
    a, b = ops.gru_gate_train(gate, xg, yg, fork=True)
    assert a.data_ptr() == b.data_ptr() and torch.equal(a, out)
    ((a * g1).sum() + (b * g2).sum()).backward()
    for t, r in zip(params, ref):
        assert torch.equal(t.grad, r), float((t.grad - r).abs().max())
    for t in params:
        t.grad = None
    a, b = ops.gru_gate_train(gate, xg, yg, fork=True)          # only one consumer has a gradient
    (b * g2).sum().backward()
    out = ops.gru_gate_train(gate, xg.detach().requires_grad_(True), yg)
    assert all(t.grad is not None and bool(torch.isfinite(t.grad).all()) for t in params)


def test_captured_graphs_hold_no_memset_nodes(tmp_path):
    """No framework reduction inside a captured graph (tools/graph_reduce_hazard.py): torch's tall column sums
    show up as a MEMSET node (their semaphore) in front of a reduce kernel, and that pair is not replay-safe on this runtime.  A
    subprocess with DEBUG_HIP_GRAPH_DOT_PRINT=1 runs one update of a small post-LN TrXL and of a small pre-LN GTrXL trainer (the two
    block layouts; visual and vector observations) and every graph the runtime dumps is scanned."""
    import subprocess
    import sys
    script = r"""
import sys
sys.path[:0] = [%r, %r]
import __graft_entry__ as ge
ge.smoke()
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "episodic-transformer-memory-ppo_amd"))
    env = dict(os.environ, DEBUG_HIP_GRAPH_DOT_PRINT="1")
    res = subprocess.run([sys.executable, "-c", script], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    dumps = sorted(p for p in os.listdir(tmp_path) if p.startswith("graph_"))
    assert len(dumps) >= 4, dumps                  # two rollout step graphs + two optimisation step graphs at least
    sizes = []
    for name in dumps:
        text = open(os.path.join(tmp_path, name)).read()
        sizes.append(text.count("StreamId"))
        assert "MEMSET" not in text, f"{name}: a memset node (a framework reduction's semaphore?) inside a captured graph"
    assert max(sizes) > 40, sizes                  # the optimisation steps were among them


def test_grouped_column_sums_full_problem_table():
    """etm_colsum_reduce_grouped with etm_colsum_reduce_max_problems() (64 since round 5: every model here fits one launch) ragged
    problems -- different row counts, column counts (not multiples of 64), row strides, first columns -- against one launch per
    problem (bit-identical) and float64; one problem more is refused (ETM_EINVAL), not truncated."""
    import ctypes
    from etm import lib as etm_lib, ops
    dev = _dev()
    lib = etm_lib.load()
    n = lib.etm_colsum_reduce_max_problems()
    assert n == 64
    rng = np.random.default_rng(5)
    parts, outs, P, C, LD, C0 = [], [], [], [], [], []
    for i in range(n + 1):
        p_, c_ = int(rng.integers(1, 300)), int(rng.integers(1, 500))
        c0, pad = int(rng.integers(0, 9)), int(rng.integers(0, 70))
        parts.append(torch.randn((p_, c0 + c_ + pad), device=dev))
        outs.append(torch.full((c_,), float("nan"), device=dev))
        P.append(p_); C.append(c_); LD.append(c0 + c_ + pad); C0.append(c0)
    arr = lambda t, v: (t * len(v))(*v)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda k: lib.etm_colsum_reduce_grouped(arr(ctypes.c_void_p, [parts[i].data_ptr() + 4 * C0[i] for i in range(k)]), arr(ctypes.c_int32, P[:k]),
                                                   arr(ctypes.c_int32, C[:k]), arr(ctypes.c_int32, LD[:k]),
                                                   arr(ctypes.c_void_p, [outs[i].data_ptr() for i in range(k)]), k, st)
    assert call(n + 1) == -1          # ETM_EINVAL (include/etm_hip.h)
    assert all(bool(torch.isnan(o).all()) for o in outs)
    assert call(n) == 0
    for i in range(n):
        want = parts[i][:, C0[i]: C0[i] + C[i]].double().sum(0)
        assert float((outs[i].double() - want).abs().max()) <= 1e-5 * max(1.0, P[i] ** 0.5), i
        single = ops.colsum_rows(parts[i][:, C0[i]:], P[i], C[i])
        assert torch.equal(single, outs[i]), i


@pytest.mark.parametrize("N", [512, 601, 2048])
def test_encoder_backward_data_with_lds_resident_gradient_images(N):
    """csrc/conv_dgrad_lds.hip (4 x 4 / stride 2 layer: zero-bordered gradient images resident in LDS, the four stride classes as four
    channel tiles of a dense 2 x 2 convolution, ReLU mask of the layer below in the epilogue) against conv_gemm_kernel of the same
    build through etm_conv_train_dgrad, with and without the mask, ragged last group (601 = 4 * 150 + 1) -- and, N = 512, against
    the float64 transposed convolution."""
    from etm import lib as etm_lib
    from etm import ops
    dev = _dev()
    lib = etm_lib.load()
    torch.manual_seed(N + 2)
    st = torch.cuda.current_stream(dev).cuda_stream
    c, hw, cout, k, s = 32, 20, 64, 4, 2
    ho = (hw - k) // s + 1
    wt = torch.randn((cout, c, k, k), device=dev) * 0.05
    pd = ops.conv_pack_dgrad_weights(wt, s)
    dy = torch.randn((N, ho, ho, cout), device=dev)
    y_below = torch.randn((N, hw, hw, c), device=dev)
    try:
        for mask in (y_below, None):
            res = {}
            for on in (1, 0):
                etm_lib.check(lib.etm_conv_train_set_dgrad_lds(on), "set_dgrad_lds")
                dx = torch.full((N, hw, hw, c), float("nan"), device=dev)
                etm_lib.check(lib.etm_conv_train_dgrad(dy.data_ptr(), pd.data_ptr(), None if mask is None else mask.data_ptr(), dx.data_ptr(),
                                                       N, c, hw, hw, cout, k, k, s, st), "etm_conv_train_dgrad")
                res[on] = dx
            a, d = res[1], res[0]
            assert bool(torch.isfinite(a).all())
            assert bool(((a == 0) == (d == 0)).all()) or mask is None        # the same elements are masked
            rel = float((a - d).double().norm() / d.double().norm())
            assert rel < 2e-6, (mask is not None, rel)                         # measured ~2e-7: fp32 summation order only
            if N == 512 and mask is None:
                want = torch.nn.functional.conv_transpose2d(dy.permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), stride=s).permute(0, 2, 3, 1)
                assert float((a.double().cpu() - want).norm() / want.norm()) < 2e-6
    finally:
        etm_lib.check(lib.etm_conv_train_set_dgrad_lds(-1), "set_dgrad_lds")


@pytest.mark.parametrize("N", [512, 601, 2048])
def test_encoder_weight_gradients_with_lds_resident_images(N):
    """csrc/conv_wgrad_lds.hip (layer input and gradient image resident in LDS, pixel pairs, one slice per workgroup; N >= 512)
    against conv_wgrad_kernel of the same build layer by layer (both through etm_conv_train_wgrad; different summation order) and
    -- N = 512 -- against float64 weight / bias gradients; with the fused minibatch gather on the first layer, ragged last group
    (601 = 2 * 300 + 1), odd images (81 and 49 pixels: one zero pixel in the last pair)."""
    from etm import lib as etm_lib
    dev = _dev()
    lib = etm_lib.load()
    torch.manual_seed(N + 1)
    st = torch.cuda.current_stream(dev).cuda_stream
    layers = [(3, 84, 32, 8, 4), (32, 20, 64, 4, 2), (64, 9, 64, 3, 1)]
    worst = 0.0
    try:
        for li, (c, hw, cout, k, s) in enumerate(layers):
            ho = (hw - k) // s + 1
            K = k * k * c
            bank = torch.rand((N + 19, hw, hw, c), device=dev)
            index = torch.randperm(N + 19, device=dev)[:N].contiguous()
            dy = torch.randn((N, ho, ho, cout), device=dev)
            for use_index in ((False, True) if li == 0 else (False,)):
                res = {}
                for mask in (7, 0):
                    etm_lib.check(lib.etm_conv_train_set_wgrad_lds(mask), "set_wgrad_lds")
                    nbytes = lib.etm_conv_train_wgrad_workspace_bytes(N, c, hw, hw, cout, k, k, s)
                    ws = torch.full((nbytes // 4,), float("nan"), device=dev)
                    buf = torch.full((K * cout + cout,), float("nan"), device=dev)
                    etm_lib.check(lib.etm_conv_train_wgrad(bank.data_ptr(), index.data_ptr() if use_index else None, dy.data_ptr(), buf.data_ptr(),
                                                           ws.data_ptr(), nbytes, N, c, hw, hw, cout, k, k, s, st), "etm_conv_train_wgrad")
                    res[mask] = buf
                a, d = res[7], res[0]
                assert bool(torch.isfinite(a).all())
                for lo, hi in ((0, K * cout), (K * cout, K * cout + cout)):
                    rel = float((a[lo:hi] - d[lo:hi]).double().norm() / d[lo:hi].double().norm())
                    worst = max(worst, rel)
                    assert rel < 2e-6, (li, use_index, lo, rel)          # measured ~2e-7: fp32 summation order only
                if N == 512:
                    x64 = (bank[index] if use_index else bank[:N]).permute(0, 3, 1, 2).double().cpu()
                    g64 = dy.permute(0, 3, 1, 2).double().cpu()
                    w64 = torch.nn.grad.conv2d_weight(x64, (cout, c, k, k), g64, stride=s)
                    got = a[: K * cout].view(cout, c, k, k).double().cpu()
                    assert float((got - w64).norm() / w64.norm()) < 2e-6
                    assert float((a[K * cout:].double().cpu() - g64.sum((0, 2, 3))).norm() / g64.sum((0, 2, 3)).norm()) < 2e-6
    finally:
        etm_lib.check(lib.etm_conv_train_set_wgrad_lds(-1), "set_wgrad_lds")
    print(f"[wgrad lds N={N}] worst relative difference to conv_wgrad_kernel: {worst:.2e}")


@pytest.mark.parametrize("N", [96, 7])
def test_encoder_weight_gradients_through_the_grouped_slice_reduction(N):
    """etm_conv_wgrad_reduce_grouped (+ etm_conv_pack_weights_grouped in the forward): the six encoder gradients written by the
    DeferredDw collector's one reduction launch into destination views, against the per-layer reductions of the same build (same
    summation order: bit-identical; those are pinned to float64 convolutions by test_train_encoder_fwd_bwd_vs_float64_convs)."""
    from etm import ops
    dev = _dev()
    torch.manual_seed(N)
    convs = [torch.nn.Conv2d(3, 32, 8, 4).to(dev), torch.nn.Conv2d(32, 64, 4, 2).to(dev), torch.nn.Conv2d(64, 64, 3, 1).to(dev)]
    obs = torch.rand((N, 84, 84, 3), device=dev)
    params = [t for c in convs for t in (c.weight, c.bias)]
    gout = torch.randn((N, 7 * 7 * 64), device=dev)
    (ops.encoder_train(obs, *convs) * gout).sum().backward()
    ref = [t.grad.clone() for t in params]
    for t in params:
        t.grad = None
    views = [torch.full_like(t, float("nan")) for t in params]
    with ops.DeferredDw({t.data_ptr(): v for t, v in zip(params, views)}) as col:
        (ops.encoder_train(obs, *convs) * gout).sum().backward()
    assert col.written == {t.data_ptr() for t in params} and all(t.grad is None for t in params)
    for v, r in zip(views, ref):
        assert torch.equal(v, r), float((v - r).abs().max())


@pytest.mark.parametrize("N,hid,A,D", [(2048, 384, 3, 384), (37, 128, 2, 128), (130, 512, 8, 64), (5, 64, 4, 96)])
def test_fused_heads_and_loss_vs_composed_ops(N, hid, A, D):
    """etm_heads_loss (hidden heads' bias + ReLU, policy branch, value head, PPO loss, backward to the hidden heads) against the
    same computation composed of torch ops + the separate loss kernel (itself pinned to the reference's _train_mini_batch by
    loss.npz): loss, the six statistics and every gradient."""
    from etm import ops
    dev = _dev()
    torch.manual_seed(N + hid)
    lin_p, lin_v = torch.nn.Linear(D, hid).to(dev), torch.nn.Linear(D, hid).to(dev)
    branch, vhead = torch.nn.Linear(hid, A).to(dev), torch.nn.Linear(hid, 1).to(dev)
    h0 = torch.randn((N, D), device=dev)
    actions = torch.randint(0, A, (N, 1), device=dev)
    old_logp = -torch.rand((N, 1), device=dev) - 0.3
    adv = torch.randn(N, device=dev) * 2 + 0.3
    old_value = torch.randn(N, device=dev)
    clip, vf, beta = 0.15, 0.4, 0.01
    mods = (lin_p, lin_v, branch, vhead)
    res = []
    for fused in (False, True):
        for m in mods:
            m.zero_grad(set_to_none=True)
        h = h0.clone().requires_grad_(True)
        if fused:
            loss, st = ops.heads_ppo_loss(h, lin_p, lin_v, branch, vhead, actions, old_logp, adv, old_value, clip, vf, beta, unit_grad=True)
        else:
            hp, hv = torch.relu(lin_p(h)), torch.relu(lin_v(h))
            loss, st = ops.ppo_loss([branch(hp)], vhead(hv).reshape(-1), actions, old_logp, adv, old_value, clip, vf, beta)
        loss.backward()
        res.append((loss.detach().clone(), st.clone(), h.grad.clone(), [p.grad.clone() for m in mods for p in m.parameters()]))
    (l0, s0, gh0, gp0), (l1, s1, gh1, gp1) = res
    assert torch.allclose(l0, l1, atol=1e-6, rtol=1e-5) and torch.allclose(s0, s1, atol=1e-6, rtol=1e-5), (l0, l1, s0, s1)
    scale = float(gh0.abs().max())
    assert float((gh0 - gh1).abs().max()) <= 2e-5 * scale + 1e-9, float((gh0 - gh1).abs().max())
    for a, b in zip(gp0, gp1):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-8, (a.shape, float((a - b).abs().max()))


def test_linear_relu_backward_kernels_vs_autograd():
    """_LinearReluFn (mask + bias gradient in one pass, fixed-order column sums) against torch's relu(linear) in float64."""
    from etm import ops
    dev = _dev()
    torch.manual_seed(41)
    for (n, k, c) in ((2048, 384, 384), (100, 3136, 384), (7, 20, 70), (65, 33, 3)):
        x = torch.randn((n, k), device=dev, requires_grad=True)
        w = (torch.randn((c, k), device=dev) / k ** 0.5).requires_grad_()
        b = torch.randn(c, device=dev, requires_grad=True)
        go = torch.randn((n, c), device=dev)
        y = ops.linear_relu_train(x, w, b)
        gx, gw, gb = torch.autograd.grad(y, (x, w, b), go)
        xd, wd, bd = (t.detach().double().requires_grad_() for t in (x, w, b))
        yd = torch.relu(torch.nn.functional.linear(xd, wd, bd))
        rx, rw, rb = torch.autograd.grad(yd, (xd, wd, bd), go.double())
        close(y, yd.detach().cpu().numpy(), atol=2e-5, rtol=1e-5, what="linear_relu forward")
        for got, ref, what in ((gx, rx, "dx"), (gw, rw, "dw"), (gb, rb, "db")):
            close(got, ref.cpu().numpy(), atol=2e-4 if what != "dx" else 2e-5, rtol=1e-4, what=f"linear_relu {what} {(n, k, c)}")


def test_gather_rows_and_indexed_encoder_input():
    """etm_gather_rows == index_select per field (bit-exact, mixed dtypes / row sizes); the encoder kernels reading images through
    an index == the same kernels on the gathered batch (bit-exact forward and gradients)."""
    from etm import ops
    dev = _dev()
    torch.manual_seed(33)
    n_src, n = 300, 77
    fields = [torch.randint(0, 9, (n_src, 1), device=dev), torch.randn(n_src, device=dev), torch.randn((n_src, 1), device=dev),
              torch.rand((n_src, 64), device=dev) > 0.5, torch.randint(0, 50, (n_src, 64), device=dev), torch.randint(0, 5, (n_src,), device=dev),
              torch.randn((n_src, 7), device=dev), torch.rand((n_src, 3), device=dev) > 0.5]       # the last one: 3-byte rows -> index_select
    idx = torch.randint(0, n_src, (n,), device=dev)
    for got, t in zip(ops.gather_rows(fields, idx), fields):
        assert got.dtype == t.dtype and torch.equal(got, t.index_select(0, idx))
    convs = [torch.nn.Conv2d(3, 32, 8, 4).to(dev), torch.nn.Conv2d(32, 64, 4, 2).to(dev), torch.nn.Conv2d(64, 64, 3, 1).to(dev)]
    params = [p for c in convs for p in (c.weight, c.bias)]
    bank = torch.rand((40, 84, 84, 3), device=dev)
    pick = torch.randint(0, 40, (9,), device=dev)
    go = torch.randn((9, 64 * 7 * 7), device=dev)
    f_idx = ops.encoder_train(bank, *convs, index=pick)
    g_idx = torch.autograd.grad(f_idx, params, go)
    f_ref = ops.encoder_train(bank.index_select(0, pick), *convs)
    g_ref = torch.autograd.grad(f_ref, params, go)
    assert torch.equal(f_idx, f_ref) and all(torch.equal(a, b) for a, b in zip(g_idx, g_ref))


def test_rollout_hidden_partial_sums_vs_matmul():
    """etm_rollout_hidden_partial: the K-slice sums add up to x @ W^T (float64 reference) for group sizes around the 16-row
    chunk and feature sizes with a ragged last slice."""
    from etm import ops
    dev = _dev()
    torch.manual_seed(21)
    for (W, F, D) in ((16, 3136, 384), (5, 3136, 384), (32, 1024, 64), (17, 777, 96), (1, 40, 32)):
        x = torch.randn((W, F), device=dev)
        wt = (torch.randn((D, F), device=dev) / F ** 0.5).t().contiguous()
        part = ops.rollout_hidden_partial(x, wt)
        want = (x.double() @ wt.double()).cpu().numpy()
        close(part.sum(dim=0), want, atol=2e-5, rtol=1e-5, what=f"hidden partial sums {(W, F, D)}")


def test_fused_rollout_encoder_vs_library_convs():
    """conv_relu_kernel (MFMA implicit GEMM + bias + ReLU, no-grad path) == relu(conv2d) of the library for the three
    encoder layers, through ActorCriticModel._encode, including a weight update between calls (version tracking)."""
    from types import SimpleNamespace
    from model import ActorCriticModel
    dev = _dev()
    cfg = dict(hidden_layer_size=64, transformer=dict(num_blocks=1, embed_dim=64, num_heads=2, memory_length=8,
                                                      positional_encoding="", layer_norm="post", gtrxl=False, gtrxl_bias=0.0))
    torch.manual_seed(2)
    for shape, n in (((3, 84, 84), 32), ((3, 84, 84), 5), ((4, 64, 64), 3), ((3, 84, 84), 200)):   # 200: both channel tiles in one workgroup
        m = ActorCriticModel(cfg, SimpleNamespace(shape=shape), (3,), 8).to(dev)
        obs = torch.rand((n,) + shape, device=dev)
        with torch.no_grad():
            got = m._encode(obs)
            m.fused_encoder = False
            want = m._encode(obs)
            m.fused_encoder = True
            close(got, want.cpu().numpy(), atol=2e-5, rtol=1e-4, what=f"encoder {shape}")
            for conv in (m.conv1, m.conv2, m.conv3):
                conv.weight.mul_(1.1)
            got2 = m._encode(obs)
            m.fused_encoder = False
            want2 = m._encode(obs)
            close(got2, want2.cpu().numpy(), atol=2e-5, rtol=1e-4, what=f"encoder after update {shape}")
        assert not torch.allclose(got, got2)


def test_fused_gru_gate_vs_module_ops():
    """Rollout GRU gate (concatenated GEMMs + two elementwise kernels) == the six-map formulation, also after a weight change."""
    from transformer import GRUGate
    dev = _dev()
    torch.manual_seed(9)
    for D, N, bias in ((384, 32, 0.0), (128, 5, 2.0), (64, 1, 0.0)):
        gate = GRUGate(D, bias).to(dev)
        x = torch.randn((N, D), device=dev)
        y = torch.randn((N, D), device=dev)
        with torch.no_grad():
            want = gate(x, y)                      # copies not built yet -> reference formulation
            gate.refresh_rollout_weights()
            got = gate(x, y)
            close(got, want.cpu().numpy(), atol=2e-6, rtol=1e-5, what=f"gate D={D}")
            gate.Wz.weight.mul_(0.5)
            gate.bg.add_(0.3)
            got2 = gate(x, y)                      # version tracking refreshes the concatenated copies
            r = torch.sigmoid(gate.Wr(y) + gate.Ur(x))
            z = torch.sigmoid(gate.Wz(y) + gate.Uz(x) - gate.bg)
            h = torch.tanh(gate.Wg(y) + gate.Ug(r * x))
            close(got2, ((1 - z) * x + z * h).cpu().numpy(), atol=2e-6, rtol=1e-5, what=f"gate updated D={D}")


def _grads(out, wrt, go):
    gs = torch.autograd.grad(out, wrt, go, allow_unused=True)
    return [g for g in gs]


def _rel(a, b):
    return float((a - b).norm() / max(float(b.norm()), 1e-12))


@pytest.mark.parametrize("N,D", [(2048, 384), (37, 128), (5, 64), (3, 1024), (130, 96)])
@pytest.mark.parametrize("has_bias,relu,has_res", [(True, False, True), (True, True, True), (False, False, False), (False, False, True),
                                                   (True, False, False)])
def test_fused_layernorm_fwd_bwd_vs_torch(N, D, has_bias, relu, has_res):
    """Training-side fused LayerNorm(act(a + bias) + res): forward and every gradient against the framework ops it replaces
    (transformer.py:131-149, :160-170), on the device in fp32 (tolerance: summation order only)."""
    from etm import ops
    dev = _dev()
    torch.manual_seed(N + D)
    norm = torch.nn.LayerNorm(D).to(dev)
    with torch.no_grad():
        norm.weight.add_(0.2 * torch.randn(D, device=dev))
        norm.bias.add_(0.2 * torch.randn(D, device=dev))
    a = torch.randn((N, D), device=dev, requires_grad=True)
    bias = (0.3 * torch.randn(D, device=dev)).requires_grad_(True) if has_bias else None
    res = torch.randn((N, D), device=dev, requires_grad=True) if has_res else None
    go = torch.randn((N, D), device=dev)
    got = ops.fused_layernorm(a, norm, bias=bias, res=res, relu=relu)
    t = a if bias is None else a + bias
    if relu:
        t = torch.relu(t)
    if res is not None:
        t = t + res
    want = norm(t)
    close(got, want.detach().cpu().numpy(), atol=3e-6, rtol=1e-5, what="forward")
    wrt = [x for x in (a, bias, res, norm.weight, norm.bias) if x is not None]
    names = [n for n, x in zip(["a", "bias", "res", "gamma", "beta"], (a, bias, res, norm.weight, norm.bias)) if x is not None]
    for name, g, w in zip(names, _grads(got, wrt, go), _grads(want, wrt, go)):
        assert _rel(g, w) < 2e-5, (name, _rel(g, w))


@pytest.mark.parametrize("N,D,bg", [(2048, 384, 0.0), (33, 128, 2.0), (4, 64, 0.0)])
def test_gru_gate_train_fwd_bwd_vs_torch(N, D, bg):
    """Training-side GTrXL gate (concatenated GEMMs + fused kernels, hand-written backward) against the six-map formulation of
    transformer.py:287-298 with framework autograd."""
    from etm import ops
    from transformer import GRUGate
    dev = _dev()
    torch.manual_seed(D)
    gate = GRUGate(D, bg).to(dev)
    with torch.no_grad():
        gate.bg.add_(0.1 * torch.randn(D, device=dev))
    x = torch.randn((N, D), device=dev, requires_grad=True)
    y = torch.randn((N, D), device=dev, requires_grad=True)
    go = torch.randn((N, D), device=dev)
    got = ops.gru_gate_train(gate, x, y)
    r = torch.sigmoid(gate.Wr(y) + gate.Ur(x))
    z = torch.sigmoid(gate.Wz(y) + gate.Uz(x) - gate.bg)
    want = (1 - z) * x + z * torch.tanh(gate.Wg(y) + gate.Ug(r * x))
    close(got, want.detach().cpu().numpy(), atol=3e-6, rtol=1e-5, what="forward")
    wrt = [x, y] + [p for p in gate.parameters()]
    names = ["x", "y"] + [n for n, _ in gate.named_parameters()]
    for name, g, w in zip(names, _grads(got, wrt, go), _grads(want, wrt, go)):
        assert _rel(g, w) < 3e-5, (name, _rel(g, w))
    assert gate(x, y).grad_fn is not None and type(gate(x, y).grad_fn).__name__.startswith("_GruGateFn"), "the module must route training through the fused gate"


def test_flat_adamw_with_clipping_vs_torch():
    """clip_grad_norm_ + torch.optim.AdamW (upstream trainer.py:311-312; single-tensor CPU path = the reference's arithmetic) against
    the two-launch step on flat arenas, five steps with a changing learning rate; parameter views must stay live nn.Parameters."""
    from etm.optim import FlatAdamW
    dev = _dev()
    torch.manual_seed(21)
    shapes = [(64, 33), (7,), (3, 5, 2, 2), (1,), (130, 64), (64,)]
    ref = [torch.nn.Parameter(torch.randn(s) * 0.3) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref]
    opt_ref = torch.optim.AdamW(ref, lr=3e-4, foreach=False, fused=False)
    opt = FlatAdamW(mine, lr=3e-4)
    assert all(p.data_ptr() >= opt.flat_params.data_ptr() for p in mine)
    for it in range(5):
        lr = 3e-4 * (1 - 0.1 * it)
        scale = 10.0 if it % 2 == 0 else 0.01            # clipped / not clipped
        grads = [torch.randn(s) * scale for s in shapes]
        for p, g in zip(ref, grads):
            p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(ref, max_norm=0.5)
        for pg in opt_ref.param_groups:
            pg["lr"] = lr
        opt_ref.step()
        for v, g in zip(opt.grad_views, grads):
            v.copy_(g.to(dev))
        opt.set_lr(lr)
        opt.step(0.5)
        want_norm = float(torch.cat([g.reshape(-1) for g in grads]).norm())
        assert abs(float(opt.total_norm) - want_norm) <= 1e-5 * want_norm
        for p, q, g in zip(ref, mine, opt.grad_views):
            close(q, p.detach().numpy(), atol=1e-7, rtol=2e-6, what=f"param step {it}")
            close(g, p.grad.numpy(), atol=1e-8, rtol=2e-6, what=f"clipped grad step {it}")
    assert int(opt.step_dev) == 5


@pytest.mark.gpu
def test_conv_pack_kernel_matches_the_tensor_formulation():
    """etm_conv_pack_weights (both packings of a layer in one launch) == conv_pack_weights / conv_pack_dgrad_weights, bit-exact."""
    from etm import ops, lib as etm_lib
    lib = etm_lib.load()
    dev = _dev()
    torch.manual_seed(5)
    for (cout, c, k, s) in ((32, 3, 8, 4), (64, 32, 4, 2), (64, 64, 3, 1), (32, 64, 6, 3)):
        w = torch.randn((cout, c, k, k), device=dev)
        fwd = torch.empty(w.numel(), device=dev)
        st = torch.cuda.current_stream().cuda_stream
        dg = torch.empty(w.numel(), device=dev) if c % 32 == 0 else None
        etm_lib.check(lib.etm_conv_pack_weights(w.data_ptr(), fwd.data_ptr(), 0 if dg is None else dg.data_ptr(), cout, c, k, k, s, st), "pack")
        assert torch.equal(fwd, ops.conv_pack_weights(w.permute(0, 2, 3, 1).reshape(cout, -1)).reshape(-1))
        if dg is not None:
            assert torch.equal(dg, ops.conv_pack_dgrad_weights(w, s).reshape(-1))



@pytest.mark.parametrize("N,hw", [(5, 84), (64, 84), (7, 36), (33, 36)])
def test_train_encoder_fwd_bwd_vs_float64_convs(N, hw):
    """Training-side encoder kernels (fp32-MFMA implicit GEMMs with fused bias / ReLU / mask / bias gradient) against the
    convolutions of model.py:90-94 evaluated in float64 on the host: features and every weight / bias gradient.  (The library's
    fp32 weight-gradient kernels are themselves ~1e-3 off the float64 result, so they are not the yardstick.)"""
    from etm import ops
    dev = _dev()
    torch.manual_seed(N + hw)
    convs = [torch.nn.Conv2d(3, 32, 8, 4), torch.nn.Conv2d(32, 64, 4, 2), torch.nn.Conv2d(64, 64, 3, 1)]
    for c in convs:
        with torch.no_grad():
            c.bias.add_(0.05 * torch.randn_like(c.bias))
    ref = [torch.nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride).double() for c in convs]
    for r, c in zip(ref, convs):
        r.load_state_dict({k: v.double() for k, v in c.state_dict().items()})
        c.to(dev)
    assert ops.encoder_train_supported((3, hw, hw), convs)
    x = torch.rand((N, 3, hw, hw))
    got = ops.encoder_train(x.to(dev).permute(0, 2, 3, 1).contiguous(), *convs)            # (h, w, c) flatten order
    h = x.double()
    for r in ref:
        h = torch.relu(r(h))
    want = h.permute(0, 2, 3, 1).reshape(N, -1)
    assert got.shape == want.shape
    close(got, want.detach().numpy(), atol=2e-5, rtol=1e-4, what="features")
    go = torch.randn(want.shape)
    g_got = torch.autograd.grad(got, [p for c in convs for p in (c.weight, c.bias)], go.to(dev))
    g_want = torch.autograd.grad(want, [p for r in ref for p in (r.weight, r.bias)], go.double())
    for i, (a, b) in enumerate(zip(g_got, g_want)):
        # gradients within 2e-4 of the tensor norm; measured ~1e-6 (fp32 summation order only)
        assert a.shape == b.shape and _rel(a.cpu().double(), b) < 2e-5, (i, _rel(a.cpu().double(), b))


@pytest.mark.parametrize("N", [512, 601, 2048])
def test_encoder_forward_with_lds_resident_images(N):
    """csrc/conv_fwd_lds.hip (groups of input images resident in LDS, N >= 512) against the direct-from-L2 forward kernels of the
    same build layer by layer (both through etm_conv_train_fwd; each is pinned to float64 convolutions at small N by
    test_train_encoder_fwd_bwd_vs_float64_convs) and -- first 48 images -- against float64 convolutions; with and without the
    fused minibatch gather (layer 1; layers 2 / 3 with an index keep the direct kernel), ragged last image group
    (601 = 2 * 300 + 1 = 4 * 150 + 1)."""
    from etm import lib as etm_lib
    from etm import ops
    dev = _dev()
    lib = etm_lib.load()
    torch.manual_seed(N)
    st = torch.cuda.current_stream(dev).cuda_stream
    layers = [(3, 84, 32, 8, 4), (32, 20, 64, 4, 2), (64, 9, 64, 3, 1)]
    for (c, hw, cout, k, s) in layers:
        ho = (hw - k) // s + 1
        bank = torch.rand((N + 37, hw, hw, c), device=dev)
        index = torch.randperm(N + 37, device=dev)[:N].contiguous()
        wt = torch.randn((cout, c, k, k), device=dev) * 0.05
        b = torch.randn(cout, device=dev) * 0.1
        packed = ops.conv_pack_weights(wt.permute(0, 2, 3, 1).reshape(cout, -1))
        outs = {}
        for use_index in (False, True):
            for lds_on in (1, 0):
                etm_lib.check(lib.etm_conv_train_set_fwd_lds(7 if lds_on else 0), "set_fwd_lds")
                y = torch.full((N, ho, ho, cout), float("nan"), device=dev)
                etm_lib.check(lib.etm_conv_train_fwd(bank.data_ptr(), index.data_ptr() if use_index else None, bank.shape[0], packed.data_ptr(),
                                                     b.data_ptr(), y.data_ptr(), N, c, hw, hw, cout, k, k, s, 0, st), "etm_conv_train_fwd")
                outs[(use_index, lds_on)] = y
        etm_lib.check(lib.etm_conv_train_set_fwd_lds(-1), "set_fwd_lds")      # back to the default (layer 2 only)
        for use_index in (False, True):
            a, d = outs[(use_index, 1)], outs[(use_index, 0)]
            assert bool(torch.isfinite(a).all())
            assert float((a - d).abs().max()) <= 1e-5 * max(1.0, float(d.abs().max())), (c, use_index, float((a - d).abs().max()))
        x64 = bank[index[:48]].permute(0, 3, 1, 2).double().cpu()
        want = torch.relu(torch.nn.functional.conv2d(x64, wt.double().cpu(), b.double().cpu(), stride=s)).permute(0, 2, 3, 1)
        close(outs[(True, 1)][:48], want.numpy(), atol=2e-5, rtol=1e-5, what=f"layer C={c} forward vs float64")


@pytest.mark.parametrize("N", [7, 601, 2048])
def test_encoder_passes_on_the_bf16_matrix_pipe_vs_float64(N):
    """Round 6, csrc/conv_b3.hip + conv_b3_wgrad.hip (model.py:40-56, :90-92 and their backward): the eight encoder passes with every
    fp32 product taken as six bf16 MFMA products of exactly split operands, through the C ABI, against float64 -- convolutions on the
    host for the first / last images (forward, backward-data), the unfolded contraction in float64 on the device over ALL images (weight
    and bias gradients) -- next to the fp32-MFMA kernels of rounds 2 - 3 on the same inputs.  The claim under test is "fp32 accuracy":
    the error of every pass is at most the fp32-MFMA kernel's (measured 0.3 - 0.6 x: 8e-8 .. 1.3e-7 against 2.5e-7 .. 4.6e-7) and below
    3e-7 of the result's norm.  Also: ragged last groups (7, 601), the fused minibatch gather (layer 1, forward and weight gradient),
    the ReLU pattern words the forward pass writes and backward-data reads (identical to the mask taken from y_below's values)."""
    import ctypes
    import torch.nn.functional as F
    from etm import lib as etm_lib
    from etm import ops
    dev = _dev()
    lib = etm_lib.load()
    torch.manual_seed(1000 + N)
    st = torch.cuda.current_stream(dev).cuda_stream
    P = lambda t: None if t is None else t.data_ptr()
    rel = lambda a, ref: float((a.double().cpu() - ref).norm() / ref.norm())
    one = lambda ct, v: (ct * 1)(v)
    nref = min(N, 24)
    sel = torch.cat([torch.arange(nref // 2), torch.arange(N - (nref - nref // 2), N)]) if N > nref else torch.arange(N)

    def bits_of(t):      # bit c % 32 of word [..., c / 32] = (t > 0)
        w = ((t > 0).view(*t.shape[:-1], t.shape[-1] // 32, 32).to(torch.int64) << torch.arange(32, device=t.device)).sum(-1)
        return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()

    for li, (c, hw, cout, k, s) in enumerate([(3, 84, 32, 8, 4), (32, 20, 64, 4, 2), (64, 9, 64, 3, 1)]):
        ho = (hw - k) // s + 1
        bank = torch.rand((N + 5, hw, hw, c), device=dev) if li == 0 else torch.relu(torch.randn((N + 5, hw, hw, c), device=dev))
        index = torch.randperm(N + 5, device=dev)[:N].contiguous()
        x = bank[index].contiguous()
        wt = torch.randn((cout, c, k, k), device=dev) * 0.05
        b = torch.randn(cout, device=dev) * 0.1
        dy = torch.randn((N, ho, ho, cout), device=dev) * (torch.rand((N, ho, ho, cout), device=dev) > 0.5)
        fwd_p, dg_p = ops.conv_b3_pack([wt, wt], [0, 1], [s, s]) if li else (ops.conv_b3_pack([wt], [0], [s])[0], None)
        # ---- forward (+ pattern words), plain and through the index
        y3 = torch.full((N, ho, ho, cout), float("nan"), device=dev)
        ybits = torch.zeros((N, ho, ho, cout // 32), dtype=torch.int32, device=dev)
        etm_lib.check(lib.etm_conv_b3_fwd(P(x), None, P(fwd_p), P(b), P(y3), P(ybits), N, c, hw, hw, cout, k, k, s, st), "etm_conv_b3_fwd")
        y32 = torch.empty_like(y3)
        packed = ops.conv_pack_weights(wt.permute(0, 2, 3, 1).reshape(cout, -1))
        etm_lib.check(lib.etm_conv_train_fwd(P(x), None, N, P(packed), P(b), P(y32), N, c, hw, hw, cout, k, k, s, 0, st), "etm_conv_train_fwd")
        want = torch.relu(F.conv2d(x[sel].double().cpu().permute(0, 3, 1, 2), wt.double().cpu(), b.double().cpu(), stride=s)).permute(0, 2, 3, 1)
        e3, e32 = rel(y3[sel], want), rel(y32[sel], want)
        assert bool(torch.isfinite(y3).all()) and e3 <= max(e32, 1e-7) and e3 < 3e-7, (li, "forward", e3, e32)
        assert bool((bits_of(y3) == ybits).all()), (li, "ReLU pattern words")
        if li == 0:
            yi = torch.full_like(y3, float("nan"))
            etm_lib.check(lib.etm_conv_b3_fwd(P(bank), P(index), P(fwd_p), P(b), P(yi), None, N, c, hw, hw, cout, k, k, s, st), "etm_conv_b3_fwd (index)")
            assert bool((yi == y3).all()), "forward through the minibatch index"
        # ---- backward-data: pattern from the words, from the values, none
        if li:
            dx3, dxv, dxn, dx32 = (torch.full((N, hw, hw, c), float("nan"), device=dev) for _ in range(4))
            etm_lib.check(lib.etm_conv_b3_dgrad(P(dy), None, P(dg_p), None, P(bits_of(x)), P(dx3), N, c, hw, hw, cout, k, k, s, st), "etm_conv_b3_dgrad")
            etm_lib.check(lib.etm_conv_b3_dgrad(P(dy), None, P(dg_p), P(x), None, P(dxv), N, c, hw, hw, cout, k, k, s, st), "etm_conv_b3_dgrad (values)")
            etm_lib.check(lib.etm_conv_b3_dgrad(P(dy), None, P(dg_p), None, None, P(dxn), N, c, hw, hw, cout, k, k, s, st), "etm_conv_b3_dgrad (no mask)")
            etm_lib.check(lib.etm_conv_train_dgrad(P(dy), P(ops.conv_pack_dgrad_weights(wt, s)), P(x), P(dx32), N, c, hw, hw, cout, k, k, s, st), "etm_conv_train_dgrad")
            full = F.conv_transpose2d(dy[sel].double().cpu().permute(0, 3, 1, 2), wt.double().cpu(), stride=s).permute(0, 2, 3, 1)
            want = full * (x[sel].double().cpu() > 0)
            e3, e32 = rel(dx3[sel], want), rel(dx32[sel], want)
            assert bool(torch.isfinite(dx3).all()) and e3 <= max(e32, 1e-7) and e3 < 3e-7, (li, "backward-data", e3, e32)
            assert bool((dxv == dx3).all()) and rel(dxn[sel], full) < 3e-7, (li, "backward-data mask forms")
            # the layer's OWN ReLU backward at the fill: (gradient of the activation, pattern words of y) == the pre-masked gradient, bit for bit
            gact = torch.randn((N, ho, ho, cout), device=dev)
            dym = gact * (y3 > 0)
            dxa, dxb = torch.full_like(dx3, float("nan")), torch.full_like(dx3, float("nan"))
            etm_lib.check(lib.etm_conv_b3_dgrad(P(dym), None, P(dg_p), None, P(bits_of(x)), P(dxa), N, c, hw, hw, cout, k, k, s, st), "etm_conv_b3_dgrad (pre-masked)")
            etm_lib.check(lib.etm_conv_b3_dgrad(P(gact), P(ybits), P(dg_p), None, P(bits_of(x)), P(dxb), N, c, hw, hw, cout, k, k, s, st), "etm_conv_b3_dgrad (dy_relu_bits)")
            assert bool((dxa == dxb).all()), (li, "backward-data with the layer's own ReLU pattern at the fill")
            wsa, wsb = torch.full_like(ws_probe := torch.empty(lib.etm_conv_b3_wgrad_slices(N, c, hw, hw, cout, k, k, s) * (k * k * c * cout + cout), device=dev), float("nan")), None
            wsb = torch.full_like(wsa, float("nan"))
            etm_lib.check(lib.etm_conv_b3_wgrad(P(x), None, P(dym), None, P(wsa), wsa.numel() * 4, N, c, hw, hw, cout, k, k, s, st), "etm_conv_b3_wgrad (pre-masked)")
            etm_lib.check(lib.etm_conv_b3_wgrad(P(x), None, P(gact), P(ybits), P(wsb), wsb.numel() * 4, N, c, hw, hw, cout, k, k, s, st), "etm_conv_b3_wgrad (dy_relu_bits)")
            assert bool((wsa == wsb).all()), (li, "weight / bias gradient slices with the layer's own ReLU pattern at the fill")
        # ---- weight / bias gradients: slices + the grouped reduction, against the float64 contraction over all images
        K = k * k * c
        slices = lib.etm_conv_b3_wgrad_slices(N, c, hw, hw, cout, k, k, s)
        assert 0 < slices <= 512
        ws = torch.full((slices * (K * cout + cout),), float("nan"), device=dev)
        dw3, db3 = torch.full((cout, c, k, k), float("nan"), device=dev), torch.full((cout,), float("nan"), device=dev)

        def wgrad(src, idx):
            assert lib.etm_conv_b3_wgrad(P(src), P(idx), P(dy), None, P(ws), ws.numel() * 4 - 4, N, c, hw, hw, cout, k, k, s, st) != 0, "a workspace that is too small is refused"
            etm_lib.check(lib.etm_conv_b3_wgrad(P(src), P(idx), P(dy), None, P(ws), ws.numel() * 4, N, c, hw, hw, cout, k, k, s, st), "etm_conv_b3_wgrad")
            etm_lib.check(lib.etm_conv_wgrad_reduce_grouped(one(ctypes.c_void_p, P(ws)), one(ctypes.c_int32, slices), one(ctypes.c_void_p, P(dw3)),
                                                            one(ctypes.c_void_p, P(db3)), one(ctypes.c_int32, cout), one(ctypes.c_int32, c),
                                                            one(ctypes.c_int32, k), one(ctypes.c_int32, k), 1, st), "etm_conv_wgrad_reduce_grouped")
        wgrad(x, None)
        buf32 = torch.empty(K * cout + cout, device=dev)
        nbytes = lib.etm_conv_train_wgrad_workspace_bytes(N, c, hw, hw, cout, k, k, s)
        ws32 = torch.empty(max(nbytes, 8) // 4, device=dev)
        etm_lib.check(lib.etm_conv_train_wgrad(P(x), None, P(dy), P(buf32), P(ws32), nbytes, N, c, hw, hw, cout, k, k, s, st), "etm_conv_train_wgrad")
        cols = F.unfold(x.permute(0, 3, 1, 2).double(), k, stride=s)
        ref_dw = torch.einsum("nkp,npo->ok", cols, dy.double().reshape(N, ho * ho, cout)).reshape(cout, c, k, k).cpu()
        ref_db = dy.double().sum((0, 1, 2)).cpu()
        del cols
        e3, e32 = rel(dw3, ref_dw), rel(buf32[: K * cout].view(cout, c, k, k), ref_dw)
        assert bool(torch.isfinite(dw3).all()) and e3 <= max(e32, 1e-7) and e3 < 3e-7, (li, "weight gradient", e3, e32)
        assert rel(db3, ref_db) < 3e-7, (li, "bias gradient", rel(db3, ref_db))
        if li == 0:
            keep = dw3.clone()
            wgrad(bank, index)
            assert bool((dw3 == keep).all()), "weight gradient through the minibatch index"


def test_train_encoder_minibatch_size_properties():
    """At the minibatch size of BASELINE config 3 (N = 2048, 3 x 84 x 84; too large for the float64 host reference): the features
    agree with the library convolutions, and the weight / bias gradients are ADDITIVE over the batch -- the gradient of the
    full batch equals the sum of the gradients of its eight slices (each slice covered by the float64 test's size range)."""
    from etm import ops
    dev = _dev()
    torch.manual_seed(5)
    N = 2048
    convs = [torch.nn.Conv2d(3, 32, 8, 4).to(dev), torch.nn.Conv2d(32, 64, 4, 2).to(dev), torch.nn.Conv2d(64, 64, 3, 1).to(dev)]
    params = [p for c in convs for p in (c.weight, c.bias)]
    x = torch.rand((N, 3, 84, 84), device=dev)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    got = ops.encoder_train(x_nhwc, *convs)
    with torch.no_grad():
        h = x
        for c in convs:
            h = torch.relu(c(h))
        want = h.permute(0, 2, 3, 1).reshape(N, -1)
    close(got, want.cpu().numpy(), atol=5e-5, rtol=1e-3, what="features vs library convolutions")
    go = torch.randn_like(got)
    full = torch.autograd.grad(got, params, go)
    parts = [torch.zeros_like(p) for p in params]
    for k in range(8):
        sl = slice(k * 256, (k + 1) * 256)
        gs = torch.autograd.grad(ops.encoder_train(x_nhwc[sl].contiguous(), *convs), params, go[sl].contiguous())
        for acc, g in zip(parts, gs):
            acc += g
    for i, (a, b) in enumerate(zip(full, parts)):
        assert _rel(a, b) < 5e-6, (i, _rel(a, b))


def test_poc_memory_env_learns():
    """BASELINE config (1) end to end on the MI355X path: PocMemoryEnv (goal cue visible only in the first two steps)
    through subprocess workers; the policy must learn to use its episodic memory (success >= 0.9 within 30 updates)."""
    from collections import deque
    from yaml_parser import YamlParser
    from trainer import PPOTrainer
    dev = _dev()
    here = os.path.dirname(os.path.abspath(__file__))
    cfg = YamlParser(os.path.join(here, "..", "episodic-transformer-memory-ppo_amd", "configs", "poc_memory_env.yaml")).get_config()
    torch.manual_seed(0)
    tr = PPOTrainer(cfg, run_id="poc", device=dev, tensorboard=False)
    recent = deque(maxlen=100)
    success = 0.0
    for update in range(30):
        lr, beta, clip = tr.schedules(update)
        recent.extend(tr._sample_training_data())
        tr.buffer.prepare_batch_dict()
        tr._train_epochs(lr, clip, beta)
        success = float(np.mean([i["success"] for i in recent]))
        if update >= 10 and success >= 0.95:
            break
    tr.close()
    assert success >= 0.9, success


def test_bench_prints_one_json_line(tmp_path):
    """bench.py's contract with the driver, on the DRIVER'S command shape (--gpus 1, micro-benchmarks, fresh-observation run and CPU
    baseline all ON): rank 0 prints ONE strict-JSON line on stdout, shorter than 8 KB (round 5's 20,013-character line was dropped by
    the driver's parser), carrying the headline, `roofline`, `cpu_baseline` and `config.workload`; the full record is the side file."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = tmp_path / "full.json"
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--full-json", str(full)],
                         capture_output=True, text=True, timeout=1500, cwd=repo)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[:2000]
    assert len(lines[0].encode()) < 8192, len(lines[0])

    def no_constants(name):
        raise AssertionError(f"non-strict JSON constant {name} on the line")

    rec = json.loads(lines[0], parse_constant=no_constants)
    assert rec["n_gpus"] == 1 and rec["unit"] == "env-steps/s" and rec["value"] > 0 and rec["steps"] == 2 and rec["warmup"] == 1
    assert abs(rec["ms_per_step"] * rec["steps"] * 1e-3 * rec["value"] - 2 * 32 * 512) < 0.01 * 2 * 32 * 512
    assert 0 < rec["roofline"]["frac"] < 1.5 and rec["roofline"]["peak"] > 0 and "traffic" in rec["roofline"]
    assert rec["cpu_baseline"]["value"] > 0 and rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cores"] >= 1
    assert "synthetic_minigrid" in rec["config"]["workload"] and len(rec["config"]["workload"]) <= 300
    assert rec["rooflines"]["encoder.all_passes"][1] > 0.3
    assert rec["value_worker_processes"] > 0 and rec["config"]["env_pool"] == 0          # the default environment draws every observation fresh
    side = json.loads(full.read_text())
    assert side["value"] == pytest.approx(rec["value"], rel=1e-5) and "kernels_train" in side and "rooflines" in side
