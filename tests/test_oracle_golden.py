"""The oracle (oracle/ref_model.py, oracle/ref_algo.py) against golden vectors from the real reference.

Fixtures: tests/golden/*.npz, produced by tests/golden/make_golden.py (imports /root/reference).
Tolerances (fp32): forward 1e-5 abs + 1e-5 rel, gradients 1e-4 rel of the tensor norm;
window tables / indices / masks: exact.
"""
import json
import os

import numpy as np
import pytest
import torch

import detgen as dg
from oracle import ref_algo as ra
from oracle import ref_model as rm

torch.set_num_threads(1)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def shapes_of(z, tag):
    keys = [str(k) for k in z[tag + "keys"]]
    shapes = [tuple(int(x) for x in str(s).split(",") if x) for s in z[tag + "shapes"]]
    return keys, shapes


def det_sd(case, keys, shapes, requires_grad=True):
    gen = dg.det_state_dict(case, keys, shapes)
    return {k: torch.from_numpy(v).requires_grad_(requires_grad) for k, v in gen.items()}


def close(a, b, atol=1e-5, rtol=1e-5):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b)
    assert (err <= atol + rtol * np.abs(b)).all(), f"max err {err.max():.3e}"


def grad_close(g, z, tag, key, n_sample=384, rel=1e-4):
    g = g.detach().numpy()
    norm = float(z[tag + "grad_norm/" + key])
    tol = rel * max(norm, 1e-6)
    assert abs(float(np.linalg.norm(g.astype(np.float64))) - norm) <= 10 * tol
    close(dg.sample(g, n_sample), z[tag + "grad_sample/" + key], atol=tol, rtol=1e-4)


# ------------------------------------------------------------------ tables (exact)
def test_tables_bit_exact(golden_dir):
    z = load(golden_dir, "tables.npz")
    names = [k for k in z.files if k.startswith("mask_")]
    assert len(names) >= 8
    for k in names:
        L, T = (int(x[1:]) for x in k.split("_")[1:])
        mask, idx = rm.window_tables(L, T)
        assert mask.dtype == torch.float32 and idx.dtype == torch.int64
        assert np.array_equal(mask.numpy(), z[k])
        assert np.array_equal(idx.numpy(), z[f"index_L{L}_T{T}"])
        for step in range(T):
            row, first = rm.rollout_window(step, L)
            assert row == int(z[f"maskrow_L{L}_T{T}"][step])
            assert first == int(z[f"index_L{L}_T{T}"][step, 0])
            assert rm.last_value_window(step, L) == tuple(int(x) for x in z[f"lastwin_L{L}_T{T}"][step])


def test_tables_docstring_examples():
    # the only reference-authored known answers: trainer.py:79-86 and :91-99
    mask, idx = rm.window_tables(4, 7)
    assert idx.tolist() == [[0, 1, 2, 3]] * 4 + [[1, 2, 3, 4], [2, 3, 4, 5], [3, 4, 5, 6]]
    m6, _ = rm.window_tables(6, 6)
    assert m6.tolist() == [[1.0] * r + [0.0] * (6 - r) for r in range(6)]
    with pytest.raises(ValueError):
        rm.window_tables(8, 7)


# ------------------------------------------------------------------ MHA
def test_mha_forward_backward(golden_dir):
    z = load(golden_dir, "mha.npz")
    cases = sorted({k.split("/")[0] for k in z.files})
    assert len(cases) >= 6
    for case in cases:
        tag = case + "/"
        D, H, L, n = (int(x) for x in z[tag + "dims"])
        keys, shapes = shapes_of(z, tag)
        sd = {"a." + k: v for k, v in det_sd(case, keys, shapes).items()}
        kv = torch.from_numpy(dg.det_normal(case, "kv", (n, L, D)))
        q = torch.from_numpy(dg.det_normal(case, "q", (n, 1, D))).requires_grad_(True)
        mask = torch.from_numpy(dg.leading_mask(case, n, L))
        out, att = rm.mha(sd, "a", H, kv, kv, q, mask)
        close(out.detach(), z[tag + "out"])
        close(att.detach(), z[tag + "att"], atol=1e-6)
        # Q2: fully masked row -> uniform attention
        assert np.allclose(att.detach().numpy()[0], 1.0 / L, atol=1e-6)
        go = torch.from_numpy(dg.det_normal(case, "gout", tuple(out.shape)))
        (out * go).sum().backward()
        close(q.grad, z[tag + "gq"], atol=1e-5, rtol=1e-4)
        for k in keys:
            grad_close(sd["a." + k].grad, z, tag, k)


# ------------------------------------------------------------------ transformer variants
def test_transformer_variants(golden_dir):
    z = load(golden_dir, "transformer.npz")
    cases = sorted({k.split("/")[0] for k in z.files}, key=lambda s: int(s.split("_v")[1]))
    assert len(cases) == 27
    for case in cases:
        tag = case + "/"
        info = json.loads(str(z[tag + "cfg_json"]))
        cfg, T, n = info["cfg"], info["T"], info["n"]
        D, L, nb = cfg["embed_dim"], cfg["memory_length"], cfg["num_blocks"]
        keys, shapes = shapes_of(z, tag)
        sd = {"transformer." + k: v for k, v in det_sd(case, keys, shapes).items()}
        h = torch.from_numpy(dg.det_normal(case, "h", (n, D)))
        mem = torch.from_numpy(dg.det_normal(case, "mem", (n, L, nb, D), 0.5))
        mask = torch.from_numpy(dg.leading_mask(case, n, L))
        idx = torch.from_numpy(dg.window_indices(case, n, L, T))
        out, new_mem, _ = rm.transformer(sd, cfg, T, h, mem, mask, idx)
        assert out.shape == (n, D)
        close(out.detach(), z[tag + "out"], atol=2e-5, rtol=1e-5)
        close(new_mem.detach(), z[tag + "new_mem"])
        go = torch.from_numpy(dg.det_normal(case, "gout", tuple(out.shape)))
        (out * go).sum().backward()
        for k in keys:
            if k.endswith("inv_freqs"):
                continue
            g = sd["transformer." + k].grad
            g = g if g is not None else torch.zeros(shapes[keys.index(k)])
            grad_close(g, z, tag, k, n_sample=96)


# ------------------------------------------------------------------ actor-critic
def test_actor_critic(golden_dir):
    z = load(golden_dir, "model.npz")
    for case in ("model_vec", "model_img", "model_img_post"):
        tag = case + "/"
        info = json.loads(str(z[tag + "cfg_json"]))
        cfg, T, n, obs_shape = info["cfg"], info["T"], info["n"], tuple(info["obs_shape"])
        t = cfg["transformer"]
        keys, shapes = shapes_of(z, tag)
        sd = det_sd(case, keys, shapes)
        obs = torch.from_numpy(np.abs(dg.det_normal(case, "obs", (n,) + obs_shape, 0.4)).clip(0, 1))
        mem = torch.from_numpy(dg.det_normal(case, "mem", (n, t["memory_length"], t["num_blocks"], t["embed_dim"]), 0.3))
        mask = torch.from_numpy(dg.leading_mask(case, n, t["memory_length"]))
        idx = torch.from_numpy(dg.window_indices(case, n, t["memory_length"], T))
        logits, value, new_mem = rm.actor_critic(sd, cfg, obs, mem, mask, idx, T)
        lsm = torch.log_softmax(logits[0], -1)
        close(lsm.detach(), z[tag + "log_probs_all"], atol=2e-5)
        close(value.detach(), z[tag + "value"], atol=2e-5)
        close(new_mem.detach(), z[tag + "new_mem"], atol=2e-5)
        loss = (lsm * torch.from_numpy(dg.det_normal(case, "gl", tuple(lsm.shape)))).sum() + \
               (value * torch.from_numpy(dg.det_normal(case, "gv", tuple(value.shape)))).sum()
        loss.backward()
        for k in keys:
            if k.endswith("inv_freqs"):
                continue
            grad_close(sd[k].grad, z, tag, k, n_sample=96, rel=2e-4)


def test_init_state_dict_keys_match_reference(golden_dir):
    import yaml
    z = load(golden_dir, "model.npz")
    cfg_dir = os.path.join(os.path.dirname(golden_dir), "..", "episodic-transformer-memory-ppo_amd", "configs")
    for cname, obs_shape, n_act, T in (("minigrid", (3, 84, 84), 3, 96), ("cartpole", (4,), 2, 200),
                                        ("poc_memory_env", (3,), 2, 32), ("mortar_mayhem_grid", (3, 84, 84), 4, 128)):
        cfg = yaml.safe_load(open(os.path.join(cfg_dir, cname + ".yaml")))
        sd = rm.init_state_dict(cfg, obs_shape, (n_act,), T)
        want = {str(k): tuple(int(x) for x in str(s).split(",") if x) for k, s in zip(z[f"keys/{cname}"], z[f"shapes/{cname}"])}
        assert {k: tuple(v.shape) for k, v in sd.items()} == want
        n_params = sum(v.numel() for k, v in sd.items() if not k.endswith("inv_freqs"))
        assert n_params == int(z[f"nparams/{cname}"])


# ------------------------------------------------------------------ GAE / loss / decay
def test_gae(golden_dir):
    z = load(golden_dir, "gae.npz")
    for case in sorted({k.split("/")[0] for k in z.files}):
        adv = ra.gae(z[case + "/rewards"], z[case + "/dones"], z[case + "/values"], z[case + "/last_value"],
                     float(z[case + "/gamma"]), float(z[case + "/lamda"]))
        assert np.array_equal(adv.numpy(), z[case + "/adv"]), case  # same op order -> bit exact
    hand = ra.gae(z["hand/rewards"], z["hand/dones"], z["hand/values"], z["hand/last_value"], 0.99, 0.95).numpy()
    assert np.allclose(hand, [[0.5198, -0.4000, 1.5815, 1.7900], [0.8259, 0.7740, 0.7198, -0.4000]], atol=1e-4)


def test_ppo_loss(golden_dir):
    z = load(golden_dir, "loss.npz")
    for case in sorted({k.split("/")[0] for k in z.files}):
        t = lambda k: torch.from_numpy(z[f"{case}/{k}"])
        logits = t("logits").clone().requires_grad_(True)
        value = t("value").clone().requires_grad_(True)
        clip, cv, beta = (float(x) for x in z[f"{case}/hp"])
        loss, stats = ra.ppo_loss([logits], value, t("actions"), t("old_logp"), t("adv"), t("old_v"), clip, cv, beta)
        loss.backward()
        close(stats.numpy(), z[f"{case}/stats"], atol=1e-6, rtol=1e-5)
        close(logits.grad, z[f"{case}/glogits"], atol=1e-8, rtol=1e-4)
        close(value.grad, z[f"{case}/gvalue"], atol=1e-8, rtol=1e-4)


def test_polynomial_decay(golden_dir):
    z = load(golden_dir, "decay.npz")
    for k in [k for k in z.files if k.endswith("/steps")]:
        base = k[:-len("steps")]
        ini, fin, mx, pw = z[base + "params"]
        got = [ra.polynomial_decay(float(ini), float(fin), int(mx), float(pw), int(s)) for s in z[k]]
        assert got == list(z[base + "values"]), base  # python float64: exact


# ------------------------------------------------------------------ teacher-forced rollout + updates
@pytest.mark.parametrize("name", ["vec", "gtrxl", "img", "img32", "cfg2"])
def test_teacher_forced_rollout_and_update(golden_dir, name):
    from environments.synthetic import SyntheticVecEnv
    z = load(golden_dir, f"rollout_{name}.npz")
    info = json.loads(str(z["cfg_json"]))
    cfg, envk = info["cfg"], info["env"]
    env = SyntheticVecEnv(cfg["n_workers"], **{**envk, "obs_shape": tuple(envk["obs_shape"])})
    keys, shapes = shapes_of(z, "")
    sd0 = {k: torch.from_numpy(v) for k, v in dg.det_state_dict("rollout_" + name, keys, shapes).items()}
    if cfg["transformer"]["positional_encoding"] == "relative":
        sd0["transformer.pos_embedding.inv_freqs"] = 1e4 ** (-torch.arange(0, cfg["transformer"]["embed_dim"], 2.0) / cfg["transformer"]["embed_dim"])
    tr = ra.OracleTrainer(cfg, env, state_dict=sd0, seed=0)
    for upd in range(cfg["updates"]):
        tag = f"u{upd}/"
        buf, stats, _ = tr.update(upd, forced_actions=z[tag + "actions"][:, :, 0], perms=z[tag + "perms"])
        # integer / mask bookkeeping: exact
        for k in ("memory_mask", "memory_index", "memory_indices", "dones"):
            assert np.array_equal(np.asarray(buf[k]), z[tag + k]), k
        assert np.array_equal(tr.ep_step.numpy(), z[tag + "ep_step_after"])
        assert np.array_equal(buf["rewards"], z[tag + "rewards"])
        if tag + "obs" in z:
            close(buf["obs"], z[tag + "obs"], atol=0, rtol=0)
        else:      # image observations: subsample + sum (the fixture stays small)
            ob = np.asarray(buf["obs"])
            assert np.array_equal(dg.sample(ob, 8192), z[tag + "obs_sample"]) and np.float64(ob.astype(np.float64).sum()) == z[tag + "obs_sum"]
        close(buf["values"], z[tag + "values"], atol=2e-5)
        close(buf["log_probs"], z[tag + "log_probs"], atol=2e-5)
        close(buf["advantages"], z[tag + "advantages"], atol=1e-4)
        if tag + "memories" in z:
            close(buf["memories"], z[tag + "memories"], atol=2e-5)
        else:      # many episodes: subsample + sum + shape
            mem = np.asarray(buf["memories"])
            assert tuple(mem.shape) == tuple(z[tag + "memories_shape"])
            close(dg.sample(mem, 32768), z[tag + "memories_sample"], atol=2e-5)
            assert abs(float(mem.astype(np.float64).sum()) - float(z[tag + "memories_sum"])) < 1e-3 * max(1.0, abs(float(z[tag + "memories_sum"])))
        close(stats, z[tag + "stats"], atol=2e-5, rtol=1e-3)
        lr, clip, beta = z[tag + "hp"]
        for k, v in tr.sd.items():
            if k.endswith("inv_freqs"):
                continue
            close(dg.sample(v.detach().numpy(), 64), z[tag + "sd_after_sample/" + k], atol=2e-5, rtol=1e-3)


@pytest.mark.parametrize("name", ["cfg3", "cfg5"])
def test_teacher_forced_rollout_at_baseline_model_sizes(golden_dir, name):
    """BASELINE model sizes (config 3: D 384 / H 4 / L 64 / 3 blocks / 3x84x84 / 32 workers; config 5: pre-LN GTrXL, L 128, 4 blocks):
    the oracle's first rollout against the reference's, teacher-forced.  Rollout only -- the optimisation step of these fixtures
    (one minibatch of 2,400 - 2,560 samples) takes minutes on the CPU; the whole-path comparison at these sizes is the GPU
    suite's test_trainer_teacher_forced_vs_reference[cfg3 / cfg5]."""
    from environments.synthetic import SyntheticVecEnv
    z = load(golden_dir, f"rollout_{name}.npz")
    info = json.loads(str(z["cfg_json"]))
    cfg, envk = info["cfg"], info["env"]
    env = SyntheticVecEnv(cfg["n_workers"], **{**envk, "obs_shape": tuple(envk["obs_shape"])})
    keys, shapes = shapes_of(z, "")
    sd0 = {k: torch.from_numpy(v) for k, v in dg.det_state_dict("rollout_" + name, keys, shapes).items()}
    sd0["transformer.pos_embedding.inv_freqs"] = 1e4 ** (-torch.arange(0, cfg["transformer"]["embed_dim"], 2.0) / cfg["transformer"]["embed_dim"])
    tr = ra.OracleTrainer(cfg, env, state_dict=sd0, seed=0)
    tag = "u0/"
    prev = torch.get_num_threads()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    try:
        buf = tr.sample(forced_actions=z[tag + "actions"][:, :, 0])
    finally:
        torch.set_num_threads(prev)
    for k in ("memory_mask", "memory_index", "memory_indices", "dones"):
        assert np.array_equal(np.asarray(buf[k]), z[tag + k]), k
    assert np.array_equal(tr.ep_step.numpy(), z[tag + "ep_step_after"]) and np.array_equal(buf["rewards"], z[tag + "rewards"])
    assert int(z[tag + "memory_indices"].max()) >= cfg["transformer"]["memory_length"], "the fixture's windows slide past L"
    ob = np.asarray(buf["obs"])
    assert np.array_equal(dg.sample(ob, 8192), z[tag + "obs_sample"]) and np.float64(ob.astype(np.float64).sum()) == z[tag + "obs_sum"]
    close(buf["values"], z[tag + "values"], atol=2e-5)
    close(buf["log_probs"], z[tag + "log_probs"], atol=2e-5)
    close(buf["advantages"], z[tag + "advantages"], atol=1e-4)
    mem = np.asarray(buf["memories"])
    assert tuple(mem.shape) == tuple(z[tag + "memories_shape"])
    close(dg.sample(mem, 32768), z[tag + "memories_sample"], atol=2e-5)
