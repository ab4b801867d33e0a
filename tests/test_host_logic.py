"""CPU-side checks of the product's host logic: window tables (bit-exact vs the reference's), schedules, config
surface, the C-ABI library (loads, exports every symbol include/etm_hip.h declares), loud failure without a GPU."""
import os
import re

import numpy as np
import pytest
import torch


def test_window_tables_bit_exact_vs_reference(golden_dir):
    from trainer import build_window_tables
    z = np.load(os.path.join(golden_dir, "tables.npz"))
    for k in [k for k in z.files if k.startswith("mask_")]:
        L, T = (int(x[1:]) for x in k.split("_")[1:])
        mask, idx = build_window_tables(L, T)
        assert mask.dtype == torch.float32 and idx.dtype == torch.int64
        assert np.array_equal(mask.numpy(), z[k]) and np.array_equal(idx.numpy(), z[f"index_L{L}_T{T}"])
    with pytest.raises(ValueError):
        build_window_tables(8, 7)


def test_polynomial_decay_vs_reference(golden_dir):
    from utils import polynomial_decay
    z = np.load(os.path.join(golden_dir, "decay.npz"))
    for k in [k for k in z.files if k.endswith("/steps")]:
        base = k[:-len("steps")]
        ini, fin, mx, pw = z[base + "params"]
        got = [polynomial_decay(float(ini), float(fin), int(mx), float(pw), int(s)) for s in z[k]]
        assert got == list(z[base + "values"]), base


def test_batched_index_select_matches_gather():
    from utils import batched_index_select
    x = torch.randn(5, 9, 2, 4)
    idx = torch.randint(0, 9, (5, 3))
    out = batched_index_select(x, 1, idx)
    assert out.shape == (5, 3, 2, 4)
    for b in range(5):
        assert torch.equal(out[b], x[b, idx[b]])


def test_state_dict_keys_and_param_counts(golden_dir):
    import yaml
    from types import SimpleNamespace
    from model import ActorCriticModel
    here = os.path.dirname(os.path.abspath(__file__))
    cfg_dir = os.path.join(here, "..", "episodic-transformer-memory-ppo_amd", "configs")
    z = np.load(os.path.join(golden_dir, "model.npz"))
    for cname, obs_shape, n_act, T in (("minigrid", (3, 84, 84), 3, 96), ("cartpole", (4,), 2, 200),
                                        ("poc_memory_env", (3,), 2, 32), ("mortar_mayhem_grid", (3, 84, 84), 4, 128)):
        cfg = yaml.safe_load(open(os.path.join(cfg_dir, cname + ".yaml")))
        m = ActorCriticModel(cfg, SimpleNamespace(shape=obs_shape), (n_act,), T)
        sd = m.state_dict()
        assert list(sd.keys()) == [str(k) for k in z[f"keys/{cname}"]]
        assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in z[f"shapes/{cname}"]]
        assert sum(p.numel() for p in m.parameters()) == int(z[f"nparams/{cname}"])


def test_yaml_surface_and_parser():
    from yaml_parser import YamlParser
    here = os.path.dirname(os.path.abspath(__file__))
    cfg = YamlParser(os.path.join(here, "..", "episodic-transformer-memory-ppo_amd", "configs", "minigrid.yaml")).get_config()
    assert cfg["n_workers"] == 16 and cfg["worker_steps"] == 512 and cfg["epochs"] == 5 and cfg["n_mini_batch"] == 8
    assert cfg["transformer"] == {"num_blocks": 3, "embed_dim": 384, "num_heads": 4, "memory_length": 64,
                                  "positional_encoding": "relative", "layer_norm": "post", "gtrxl": False, "gtrxl_bias": 0.0}
    syn = YamlParser(os.path.join(here, "..", "episodic-transformer-memory-ppo_amd", "configs", "synthetic_minigrid.yaml")).get_config()
    assert syn["n_workers"] == 32 and syn["environment"]["type"] == "Synthetic" and syn["transformer"] == cfg["transformer"]


def test_library_exports_every_declared_symbol():
    from etm import lib
    here = os.path.dirname(os.path.abspath(__file__))
    header = open(os.path.join(here, "..", "include", "etm_hip.h")).read()
    declared = set(re.findall(r"\b(etm_[a-z0-9_]+)\s*\(", header))
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    handle = lib.load()                       # dlopen works without a GPU; no kernel is launched here
    for name in declared:
        assert hasattr(handle, name)
    assert handle.etm_abi_version() == lib.ABI_VERSION
    assert b"not supported" in handle.etm_error_string(-2)
    assert handle.etm_mha_bwd_workspace_bytes(2048, 64, 384) > 0 and handle.etm_ppo_loss_workspace_bytes(2048) == 8 * 8 * 4


def test_product_path_refuses_cpu():
    from etm import ops
    with pytest.raises(RuntimeError, match="no CPU"):
        ops.gae(torch.zeros(2, 3), torch.zeros(2, 3, dtype=torch.bool), torch.zeros(2, 3), torch.zeros(2), 0.99, 0.95)
    if not torch.cuda.is_available():
        from trainer import PPOTrainer
        with pytest.raises(RuntimeError, match="no CPU"):
            PPOTrainer({"n_workers": 1}, device=torch.device("cpu"))


def test_product_does_not_import_oracle():
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(here, "..", "episodic-transformer-memory-ppo_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_synthetic_env_single_and_vector_streams_agree():
    from environments.synthetic import SyntheticEnv, SyntheticVecEnv
    kw = dict(obs_shape=(2, 3), num_actions=3, max_episode_steps=7, seed=5, p_done=0.2, p_reward=0.4, pool=5)
    vec = SyntheticVecEnv(3, **kw)
    singles = [SyntheticEnv(worker_id=w, **kw) for w in range(3)]
    obs_v = vec.reset()
    obs_s = np.stack([e.reset() for e in singles])
    assert np.array_equal(obs_v, obs_s)
    for t in range(60):
        o, r, d, info = vec.step(np.zeros(3, dtype=np.int64))
        for w, e in enumerate(singles):
            so, sr, sd, si = e.step([0])
            if si:
                so = e.reset()
            assert np.array_equal(o[w], so) and r[w] == sr and d[w] == sd and info[w] == si


def test_poc_memory_env_protocol():
    from environments.poc_memory_env import PocMemoryEnv
    env = PocMemoryEnv(glob=False, freeze=True, max_episode_steps=32, seed=0)
    obs = env.reset()
    assert obs.shape == (3,) and set(np.abs(obs[[0, 2]])) == {1.0}
    total, steps, done = 0.0, 0, False
    while not done:
        obs, r, done, info = env.step([1])
        steps += 1
        total += r
        if steps > 2 and not done:
            assert obs[0] == 0.0 and obs[2] == 0.0       # goals hidden after the first two steps
    assert info["length"] == steps and abs(info["reward"] - total) < 1e-6 and steps <= 32


def test_kernel_shape_validation_is_early_and_loud():
    from trainer import check_kernel_shapes
    ok = dict(embed_dim=384, num_heads=4, memory_length=64)
    check_kernel_shapes(ok)
    for bad in (dict(ok, embed_dim=100, num_heads=4), dict(ok, num_heads=8), dict(ok, memory_length=129), dict(ok, embed_dim=2048)):
        with pytest.raises(ValueError):
            check_kernel_shapes(bad)


def test_pipe_worker_protocol_and_vec_env():
    """Upstream pipe protocol (cmd, data) through real subprocesses, batched by PipeVecEnv; env errors reach the parent."""
    from environments.vec_env import PipeVecEnv, SerialVecEnv
    from utils import create_env
    from worker import Worker, WorkerException
    cfg = {"type": "Synthetic", "obs_shape": [5], "num_actions": 2, "max_episode_steps": 6, "seed": 3, "p_done": 0.3, "pool": 4}
    w = Worker(cfg, worker_id=1)
    w.child.send(("reset", None))
    first = w.child.recv()
    assert first.shape == (5,)
    w.child.send(("step", [0]))
    obs, r, d, info = w.child.recv()
    assert obs.shape == (5,) and isinstance(d, bool)
    w.child.send(("bogus", None))
    assert isinstance(w.child.recv(), WorkerException)
    w.child.send(("close", None))
    w.child.recv()
    pipe_env = PipeVecEnv(cfg, 3)
    serial = SerialVecEnv([create_env(cfg, worker_id=i) for i in range(3)])
    a, b = pipe_env.reset(), serial.reset()
    assert np.array_equal(a, b)
    for _ in range(20):
        oa, ra, da, ia = pipe_env.step(np.zeros(3, dtype=np.int64))
        ob, rb, db, ib = serial.step(np.zeros(3, dtype=np.int64))
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(da, db) and ia == ib
    pipe_env.close()


def test_vec_env_on_rows_protocol():
    """step(..., on_rows=cb): cb(lo, hi) covers [0, W) in increasing order, the rows are final when it is called, and the
    results equal a step without the callback (synthetic vectorised env and the serial front-end over single envs)."""
    from environments.synthetic import SyntheticEnv, SyntheticVecEnv
    from environments.vec_env import SerialVecEnv
    W = 6

    def make(kind):
        if kind == "vec":
            return SyntheticVecEnv(W, obs_shape=(2, 5), num_actions=3, max_episode_steps=7, seed=4, pool=4)
        return SerialVecEnv([SyntheticEnv(obs_shape=(2, 5), num_actions=3, max_episode_steps=7, seed=4, worker_id=w, pool=4) for w in range(W)])

    for kind in ("vec", "serial"):
        a, b = make(kind), make(kind)
        oa, ob = np.zeros((W, 2, 5), np.float32), np.zeros((W, 2, 5), np.float32)
        a.reset(out=oa); b.reset(out=ob)
        for t in range(20):
            seen = []

            def cb(lo, hi):
                seen.append((lo, hi, ob[lo:hi].copy()))

            ra = a.step(np.zeros(W, dtype=np.int64), out=oa)
            rb = b.step(np.zeros(W, dtype=np.int64), out=ob, on_rows=cb)
            assert [s[0] for s in seen] == sorted(s[0] for s in seen) and seen[0][0] == 0 and seen[-1][1] == W
            assert all(seen[i][1] == seen[i + 1][0] for i in range(len(seen) - 1))
            for lo, hi, rows in seen:
                assert np.array_equal(rows, ob[lo:hi])          # rows did not change after the notification
            assert np.array_equal(oa, ob) and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2])


def test_conv_pack_weights_layout_and_attention_dispatch_rules():
    """conv_pack_weights implements the fragment order documented in include/etm_hip.h; folded_supported matches the shape
    rules of etm_window_fwd; set_attention_impl validates its argument (pure host logic, no kernel call)."""
    from etm import ops
    cout, k = 64, 40
    w = torch.arange(cout * k, dtype=torch.float32).reshape(cout, k)
    packed = ops.conv_pack_weights(w).reshape(-1)
    T = cout // 32
    for g, t, half, col, j in ((0, 0, 0, 0, 0), (1, 1, 1, 5, 3), (4, 0, 1, 31, 2), (2, 1, 0, 17, 1)):
        assert packed[((g * T + t) * 64 + half * 32 + col) * 4 + j] == w[t * 32 + col, g * 8 + half * 4 + j]
    with pytest.raises(ValueError):
        ops.conv_pack_weights(torch.zeros(48, 40))
    assert ops.folded_supported(384, 64, 4) and ops.folded_supported(384, 128, 4) and ops.folded_supported(1024, 64, 8)
    assert not ops.folded_supported(1024, 128, 8) and not ops.folded_supported(48, 8, 1) and not ops.folded_supported(96, 8, 32)
    with pytest.raises(ValueError):
        ops.set_attention_impl("fast")
    ops.set_attention_impl("dense"); ops.set_attention_impl("folded")


def test_composite_vec_env_equals_single_front_end():
    """make_vec_env(groups=2) == one front-end over all workers (same per-worker streams), whether it is stepped as a whole or
    part by part (what the pipelined rollout does)."""
    from environments.vec_env import make_vec_env
    cfg = dict(type="Synthetic", obs_shape=[2, 5], num_actions=3, max_episode_steps=9, seed=3, p_done=0.1, pool=4)
    W = 8
    one, whole, parts = make_vec_env(cfg, W), make_vec_env(cfg, W, groups=2), make_vec_env(cfg, W, groups=2)
    assert hasattr(whole, "parts") and len(whole.parts) == 2 and not hasattr(one, "parts")
    o1, o2, o3 = (np.zeros((W, 2, 5), np.float32) for _ in range(3))
    one.reset(out=o1); whole.reset(out=o2); parts.reset(out=o3)
    assert np.array_equal(o1, o2) and np.array_equal(o1, o3)
    acts = np.zeros(W, dtype=np.int64)
    for t in range(40):
        r1 = one.step(acts, out=o1)
        r2 = whole.step(acts, out=o2)
        rew, don, inf = np.zeros(W, np.float32), np.zeros(W, bool), []
        for p, (lo, hi) in zip(parts.parts, parts.bounds):
            _, rew[lo:hi], don[lo:hi], i_ = p.step(acts[lo:hi], out=o3[lo:hi])
            inf.extend(i_)
        for o, r in ((o2, r2), (o3, (o3, rew, don, inf))):
            assert np.array_equal(o1, o) and np.array_equal(r1[1], r[1]) and np.array_equal(r1[2], r[2]) and r1[3] == r[3]


def test_enjoy_episode_loop_window_rule(golden_dir):
    """enjoy.py: tables equal the reference's, and the episode loop hands the model upstream's per-step window (rows
    memory_indices[t] of the episode memory with the items of the earlier steps in place, mask row clip(t, 0, L-1))."""
    import enjoy
    from torch.distributions import Categorical
    from environments.poc_memory_env import PocMemoryEnv
    z = np.load(os.path.join(golden_dir, "tables.npz"))
    L, T, nb, D = 4, 7, 2, 32
    cfg = {"transformer": {"memory_length": L, "num_blocks": nb, "embed_dim": D}}
    memory, mask, idx = enjoy.init_transformer_memory(cfg["transformer"], T, torch.device("cpu"))
    assert memory.shape == (1, T, nb, D) and not memory.any()
    assert np.array_equal(mask.numpy(), z["mask_L4_T7"]) and np.array_equal(idx.numpy(), z["index_L4_T7"])

    seen = []

    def model(obs, in_memory, m, indices):
        t = len(seen)
        assert obs.shape[0] == 1 and in_memory.shape == (1, L, nb, D) and m.shape == (1, L) and indices.shape == (1, L)
        assert np.array_equal(indices[0].numpy(), z["index_L4_T7"][t]) and np.array_equal(m[0].numpy(), z["mask_L4_T7"][min(t, L - 1)])
        # row j of the window is episode step indices[j]: written steps carry their step number + 1, the others are zero
        expect = torch.tensor([float(s + 1) if s < t else 0.0 for s in indices[0].tolist()])
        assert torch.equal(in_memory[0, :, 0, 0], expect) and torch.equal(in_memory[0, :, nb - 1, D - 1], expect)
        seen.append(t)
        return [Categorical(logits=torch.zeros(1, 3))], torch.zeros(1), torch.full((1, nb, D), float(t + 1))

    env = PocMemoryEnv(glob=False, freeze=True, max_episode_steps=T)
    rewards, info = enjoy.run_episode(model, env, cfg, torch.device("cpu"))
    assert 1 <= len(rewards) <= T and len(seen) == len(rewards)
    assert info is not None and info["length"] == len(rewards)


def test_transposing_reduction_lane_mapping_emulated():
    """Lane-level emulation of csrc/window_attn.hip's TransposeReduce (select form and the v_permlane16/32_swap form of the
    candidate build, with the documented swap semantics: odd rows of the first operand <-> even rows of the second): after the
    butterfly, v[0] of lane l holds the 64-lane total of value bit_reverse(l mod NV); also the candidate's pass 1, which merges
    rows k and k + RG/2 at offset 1 inside the row loop and continues with TransposeReduce<RG/2, 2>."""
    rng = np.random.default_rng(0)
    lanes = np.arange(64)

    def xor(v, off):
        return v[lanes ^ off]

    def swap_sum(a, b, off):
        a2, b2 = a.copy(), b.copy()
        for l in range(64):
            if (l // off) % 2 == 1:
                a2[l], b2[l - off] = b[l - off], a[l]
        return a2 + b2

    def reduce(v, off, swap):
        while off <= 32:
            up = (lanes & off) != 0
            if len(v) > 1:
                half = len(v) // 2
                v = [swap_sum(v[k], v[k + half], off) if (swap and off >= 16) else
                     np.where(up, v[k + half], v[k]) + xor(np.where(up, v[k], v[k + half]), off) for k in range(half)]
            else:
                v = [swap_sum(v[0], v[0], off)] if (swap and off >= 16) else [v[0] + xor(v[0], off)]
            off *= 2
        return v[0]

    def brev(x, bits):
        return int(format(x, f"0{bits}b")[::-1], 2)

    for nv in (8, 16, 32):
        e = [rng.integers(-50, 50, 64).astype(np.float64) for _ in range(nv)]
        for swap in (False, True):
            r = reduce(list(e), 1, swap)
            assert all(r[l] == e[brev(l % nv, nv.bit_length() - 1)].sum() for l in range(64)), (nv, swap)
    rg = 8
    e = [rng.integers(-50, 50, 64).astype(np.float64) for _ in range(rg)]
    up = (lanes & 1) != 0
    ev = [np.where(up, e[k + rg // 2], e[k]) + xor(np.where(up, e[k], e[k + rg // 2]), 1) for k in range(rg // 2)]
    r = reduce(ev, 2, True)
    assert all(r[l] == e[brev(l % rg, 3)].sum() for l in range(64))


def test_library_collective_plumbing_without_a_device():
    """etm_comm_*: RCCL is resolved lazily (dlopen), the rendezvous id can be drawn anywhere, argument errors are reported as
    ETM_EINVAL; joining a communicator needs a HIP device (on the CPU-only build container RCCL reports an error, which must
    come back as ETM_ERCCL_BASE + ncclResult_t, not as a crash)."""
    import ctypes
    from etm import lib as etm_lib
    lib = etm_lib.load()
    buf = ctypes.create_string_buffer(128)
    assert lib.etm_comm_unique_id(buf) == 0 and any(buf.raw)
    comm = ctypes.c_void_p()
    assert lib.etm_comm_init(bytes(buf.raw), 2, 1, ctypes.byref(comm)) == -1          # rank outside the world
    assert lib.etm_allreduce_f32(None, None, None, 0, None) == -1 and lib.etm_comm_destroy(None) == -1
    if not torch.cuda.is_available():
        rc = lib.etm_comm_init(bytes(buf.raw), 0, 1, ctypes.byref(comm))
        assert rc >= 100000 and b"RCCL" in lib.etm_error_string(rc)


def test_host_copier_and_threaded_environment_rows():
    """etm_host_copy (the multi-threaded memcpy behind ``copy_threads``): exact for sizes around its thresholds and chunk edges,
    from one to five threads, and a SyntheticVecEnv that uses it emits the same rows / rewards / dones as the numpy one."""
    import ctypes
    from etm import lib as etm_lib
    from environments.synthetic import SyntheticVecEnv
    lib = etm_lib.load()
    rng = np.random.default_rng(0)
    assert not lib.etm_host_copier_create(0) and not lib.etm_host_copier_create(65)
    for threads in (1, 2, 3, 5):
        h = lib.etm_host_copier_create(threads)
        assert h
        for n in (0, 1, 63, 65535, 65536, 65537, 300001, 1354752, 4 * 1354752 + 7):
            src = rng.integers(0, 256, size=n + 16, dtype=np.uint8)
            dst = np.full(n + 16, 7, dtype=np.uint8)
            assert lib.etm_host_copy(h, dst.ctypes.data + 3, src.ctypes.data + 5, n) == 0
            assert np.array_equal(dst[3:3 + n], src[5:5 + n]) and (dst[:3] == 7).all() and (dst[3 + n:] == 7).all()
        for rep in range(200):                      # back-to-back jobs (helpers still spinning)
            src = rng.integers(0, 256, size=200000, dtype=np.uint8)
            dst = np.empty_like(src)
            assert lib.etm_host_copy(h, dst.ctypes.data, src.ctypes.data, src.size) == 0
            assert np.array_equal(dst, src)
        assert lib.etm_host_copy(h, None, None, 8) != 0 and lib.etm_host_copy(None, None, None, 0) != 0
        lib.etm_host_copier_destroy(h)
    logs = []
    for ct in (1, 3):
        env = SyntheticVecEnv(6, (3, 84, 84), 3, 9, seed=4, pool=5, p_done=0.2, copy_threads=ct)
        out = np.empty((6, 3, 84, 84), np.float32)
        env.reset(out)
        log = [out.copy()]
        seen = []
        for t in range(20):
            _, r, d, info = env.step(np.zeros(6, np.int64), out=out, on_rows=lambda lo, hi: seen.append((lo, hi)))
            log += [out.copy(), r.copy(), d.copy()]
        logs.append(log)
        assert seen and seen[-1][1] == 6
    assert all(np.array_equal(a, b) for a, b in zip(*logs))


def test_rollout_group_count_follows_the_hardware_queues(monkeypatch):
    """``rollout_groups: auto`` (trainer.py): four worker groups only when GPU_MAX_HW_QUEUES >= 8 was in the environment when the
    module was imported (the HIP runtime reads it once; with 4 queues four groups serialise), and only while the groups keep
    ``rollout_min_group_size`` workers; otherwise two, otherwise one."""
    import trainer
    monkeypatch.setattr(trainer, "_HW_QUEUES_AT_IMPORT", 8)
    assert trainer.default_rollout_groups(32, 8) == 4
    assert trainer.default_rollout_groups(16, 8) == 2          # groups of 4 would be below the minimum
    assert trainer.default_rollout_groups(8, 8) == 1
    assert trainer.default_rollout_groups(30, 4) == 2          # not divisible by four
    monkeypatch.setattr(trainer, "_HW_QUEUES_AT_IMPORT", 0)    # runtime defaults (or started before the variable was set)
    assert trainer.default_rollout_groups(32, 8) == 2
    monkeypatch.setattr(trainer, "_HW_QUEUES_AT_IMPORT", 4)
    assert trainer.default_rollout_groups(32, 8) == 2


def test_worker_processes_over_shared_memory_replay_the_in_process_streams():
    """environments/shm_env.py (round 4): environments in worker processes over one shared segment produce exactly the streams of the
    in-process front-end -- observations, rewards, done flags, episode results -- through the host-driven protocol, for the whole
    front-end and group by group, with one or several environments per process; a sequence restart is acknowledged by every worker."""
    import numpy as np
    from environments.shm_env import ShmVecEnv
    from environments.synthetic import SyntheticVecEnv
    kw = dict(obs_shape=(3, 12, 12), num_actions=3, max_episode_steps=9, seed=5, p_done=0.08, p_reward=0.3, pool=8)
    for per_proc, groups in ((1, 2), (2, 2), (4, 1)):
        env = ShmVecEnv({"type": "Synthetic", **{**kw, "obs_shape": list(kw["obs_shape"])}}, 8, first_worker_id=3, groups=groups,
                        envs_per_proc=per_proc, steps_per_rollout=16)
        try:
            ref = SyntheticVecEnv(8, first_worker_id=3, **kw)
            assert np.array_equal(env.reset().copy(), ref.reset())
            rng = np.random.default_rng(0)
            for t in range(40):
                a = rng.integers(0, 3, size=8)
                if t % 2 == 0:
                    ob, r, d, inf = env.step(a)
                else:      # group by group, like the trainer's pipelined rollout
                    outs = [p.step(a[lo:hi]) for p, (lo, hi) in zip(env.parts, env.bounds)]
                    ob = np.concatenate([o[0] for o in outs])
                    r, d = np.concatenate([o[1] for o in outs]), np.concatenate([o[2] for o in outs])
                    inf = [i for o in outs for i in o[3]]
                ob2, r2, d2, inf2 = ref.step(a)
                assert np.array_equal(ob, ob2) and np.array_equal(r, r2) and np.array_equal(d, d2) and inf == inf2, (per_proc, groups, t)
            env.restart_sequence()
            assert (env.v["ready"][:, 0] == 0).all()
        finally:
            env.close()


def test_worker_processes_step_a_python_environment():
    """The worker processes of environments/shm_env.py build ANY configured environment through utils.create_env (upstream
    worker.py:20-34): PocMemoryEnv (pure python, upstream's config 1) steps behind the shared segment, episodes finish with upstream's
    info keys, and the front-end reports the environment's spaces."""
    import numpy as np
    from environments.shm_env import ShmVecEnv
    env = ShmVecEnv({"type": "PocMemoryEnv"}, 4, groups=2, envs_per_proc=1, steps_per_rollout=8)
    try:
        assert env.observation_space_shape == (3,) and env.num_actions == 2 and env.max_episode_steps == 32
        obs = env.reset()
        assert obs.shape == (4, 3) and np.isfinite(obs).all()
        rng = np.random.default_rng(1)
        finished = 0
        for _ in range(80):
            ob, r, d, infos = env.step(rng.integers(0, 2, size=4))
            assert ob.shape == (4, 3) and r.shape == (4,) and d.shape == (4,)
            for w in np.flatnonzero(d):
                finished += 1
                assert {"reward", "length"} <= set(infos[w]) and 1 <= infos[w]["length"] <= 32
            assert all(i is None for w, i in enumerate(infos) if not d[w])
        assert finished >= 4
    finally:
        env.close()


def test_parity_ratchet_file_is_consistent_and_covers_every_teacher_forced_case():
    """tests/golden/tf_measured_baseline.json (the ratchet under the teacher-forced GPU test): every entry that sits above the
    quietest path of its (case, update) is covered by a known_flips item with an existing evidence file (tools/tf_ratchet.py
    check), and every (case, path, update) the GPU test runs has an entry."""
    import ast
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(here)
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "tf_ratchet.py"), "check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    doc = json.load(open(os.path.join(here, "golden", "tf_measured_baseline.json")))
    # the case list of the GPU test, read from its source (importing the module would need nothing more, but keep this test free
    # of the GPU test module's side effects)
    tree = ast.parse(open(os.path.join(here, "test_gpu_parity.py")).read())
    cases = None
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "_TF_CASES":
            cases = ast.literal_eval(node.value)
    assert cases
    for name, path in cases:
        z = np.load(os.path.join(here, "golden", f"rollout_{name}.npz"))
        updates = json.loads(str(z["cfg_json"]))["cfg"]["updates"]
        for u in range(updates):
            assert f"{name}/{path}/{u}" in doc["entries"], f"no ratchet entry for {name}/{path}/{u}"


def test_worker_processes_park_wake_protocol_has_no_missed_wakeups():
    """ADVICE round 4 (environments/shm_env.py): (1) activate() issued exactly when a worker decides to park must still end with every
    worker active AND holding -- the activation epoch is acknowledged by the worker, a worker seen PARKED later is woken again;
    (2) reset() with the workers held active must not hang (held workers never read their pipe): the hold is dropped for the reset
    and restored; (3) a host-driven step issued at the parking edge completes (wakes are re-sent while waiting)."""
    import time
    import numpy as np
    from environments import shm_env
    from environments.shm_env import ShmVecEnv
    kw = dict(obs_shape=[2, 4, 4], num_actions=3, max_episode_steps=9, seed=5, p_done=0.08, p_reward=0.3, pool=4)
    env = ShmVecEnv({"type": "Synthetic", **kw}, 4, groups=2, envs_per_proc=1, steps_per_rollout=8)
    try:
        env.reset()
        a = np.zeros(4, dtype=np.int64)
        for i in range(12):
            env.activate(hold=False)
            # land on the workers' parking decision (IDLE_PARK_S after their last activity), a little earlier / later each time
            time.sleep(shm_env.IDLE_PARK_S + (i - 6) * 0.002)
            if i % 2 == 0:
                env.activate(hold=True)
                time.sleep(2.5 * shm_env.IDLE_PARK_S)          # a worker that missed the hold would have parked by now
                assert (env.v["state"][:, 0] == shm_env.ST_ACTIVE).all(), i
                assert (env.v["state"][:, 1] == env.v["ctl"][2]).all(), i
            env.step(a)                                            # at the edge (odd i) or under hold
        # reset under hold: returns, and the hold is in force again afterwards
        env.activate(hold=True)
        t0 = time.perf_counter()
        env.reset()
        assert time.perf_counter() - t0 < 10.0
        assert env.v["ctl"][1] == 1 and (env.v["state"][:, 0] == shm_env.ST_ACTIVE).all()
        env.step(a)
        env.park()
    finally:
        env.close()


def test_host_cpu_plan_keeps_busy_threads_within_the_rank_share():
    """etm/hostcpu.py (round 5): with CPUs to spare nothing changes; below the wanted number of busy threads the copier shrinks and its
    helpers sleep between jobs, worker processes grow (fewer of them) and stop spinning when even those do not fit, and the trainer
    thread waits politely below ~1.5 CPUs; the budget follows the affinity mask, the cgroup quota and the ranks per node."""
    from etm import hostcpu

    def bud(per_rank, world=8):
        return {"affinity": 256, "cgroup_quota": None, "usable": 256, "local_world": world, "per_rank": per_rank}

    p = hostcpu.plan_host_threads(4, False, 32, 1, budget=bud(32), quiet=True)
    assert p["copy_threads"] == 4 and p["copier_spin"] and not p["polite_wait"] and p["busy_threads"] == 4
    p = hostcpu.plan_host_threads(4, False, 32, 1, budget=bud(2), quiet=True)          # a 16-CPU quota shared by 8 ranks
    assert p["copy_threads"] == 2 and not p["copier_spin"] and not p["polite_wait"] and p["busy_threads"] <= 2
    p = hostcpu.plan_host_threads(4, False, 32, 1, budget=bud(1), quiet=True)
    assert p["copy_threads"] == 1 and p["polite_wait"] and p["busy_threads"] == 0
    p = hostcpu.plan_host_threads(4, True, 32, 1, groups=4, budget=bud(16, 1), quiet=True)      # round 4's box: 32 one-env workers on 16 CPUs
    assert p["envs_per_process"] >= 4 and 32 // p["envs_per_process"] + 1 <= 16 and p["busy_threads"] <= 16
    p = hostcpu.plan_host_threads(4, True, 32, 4, groups=4, budget=bud(2), quiet=True)
    assert p["envs_per_process"] == 8 and not p["worker_spin"]                                # four processes do not fit two CPUs: they back off
    b = hostcpu.host_cpu_budget(local_world=2)
    assert b["local_world"] == 2 and 0 < b["per_rank"] <= b["affinity"] and b["usable"] <= b["affinity"]
    mine = sorted(os.sched_getaffinity(0))
    if len(mine) >= 2:
        os.sched_setaffinity(0, set(mine[:2]))
        try:
            assert hostcpu.host_cpu_budget(local_world=1)["per_rank"] <= 2
        finally:
            os.sched_setaffinity(0, set(mine))


def test_fresh_observation_draws_are_the_same_streams_in_both_environment_forms():
    """environments/synthetic.py, pool = 0 (SURVEY 8d to the letter): every observation is a fresh draw of the worker's own generator;
    the single-environment form, the vector form and the vector form with drawing threads emit identical streams, and two steps
    never show the same frame."""
    from environments.synthetic import SyntheticEnv, SyntheticVecEnv
    kw = dict(obs_shape=(2, 5, 5), num_actions=3, max_episode_steps=7, seed=5, p_done=0.2, p_reward=0.4, pool=0)
    vec, vec_t = SyntheticVecEnv(4, **kw), SyntheticVecEnv(4, gen_threads=3, **kw)
    singles = [SyntheticEnv(worker_id=w, **kw) for w in range(4)]
    o = vec.reset()
    assert np.array_equal(o, vec_t.reset()) and np.array_equal(o, np.stack([e.reset() for e in singles]))
    seen = [o.copy()]
    for t in range(40):
        rows = []
        o, r, d, info = vec.step(np.zeros(4, dtype=np.int64), out=np.empty_like(o), on_rows=lambda a, b: rows.append((a, b)))
        o2, r2, d2, info2 = vec_t.step(np.zeros(4, dtype=np.int64))
        assert np.array_equal(o, o2) and np.array_equal(r, r2) and np.array_equal(d, d2) and info == info2
        assert rows and rows[-1][1] == 4
        for w, e in enumerate(singles):
            so, sr, sd, si = e.step([0])
            if si:
                so = e.reset()
            assert np.array_equal(o[w], so) and r[w] == sr and d[w] == sd and info[w] == si
        assert not any(np.array_equal(o, s) for s in seen)
        seen.append(o.copy())
    vec_t.close()


def test_two_ranks_on_four_cpus_do_not_slow_each_other_down_under_the_host_cpu_plan():
    """the host side of a rollout (environment stepping + the multi-threaded observation copier + the wait
    for the device) of TWO ranks restricted to four CPUs: with the plan of etm/hostcpu.py (copy_threads cut to the rank's share,
    helpers asleep between jobs) a rank's host time per step stays within 1.3 x of the rank running alone on the same mask."""
    import json
    import subprocess
    import sys
    import time
    cpus = sorted(os.sched_getaffinity(0))
    if len(cpus) < 4:
        pytest.skip("needs four CPUs")
    mask = f"{cpus[0]}-{cpus[3]}"
    if cpus[3] - cpus[0] != 3:
        pytest.skip("needs four consecutive CPUs")
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "host_plumbing_probe.py")

    def run(ranks, procs):
        start = time.time() + 4.0
        ps = [subprocess.Popen([sys.executable, tool, mask, str(ranks), "1500", "plan", str(start)], stdout=subprocess.PIPE, text=True)
              for _ in range(procs)]
        return [json.loads(p.communicate(timeout=300)[0].strip().splitlines()[-1]) for p in ps]

    best = None
    for attempt in range(3):                      # shared CI hosts are noisy: the best of three attempts counts
        alone = run(2, 1)[0]                       # the same plan (two ranks' share), but nobody else on the mask
        both = run(2, 2)
        ratio = max(r["us_per_step"] for r in both) / alone["us_per_step"]
        best = ratio if best is None else min(best, ratio)
        if best <= 1.3:
            break
    assert both[0]["copy_threads"] == 2, both
    assert best <= 1.3, (best, alone, both)


def test_bench_line_summary_is_short_strict_json():
    """bench.py's stdout line is a summary of the full record: < 8 KB of strict JSON with the contract's keys, `roofline`,
    `cpu_baseline` and one {kernel: [avg_ms, frac]} table -- checked here on the committed full record of round 5 (20,013
    characters, the line the driver's parser dropped), on a record with non-finite numbers, and on an oversized one."""
    import importlib.util
    import json
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with open(os.path.join(repo, "profiles", "r05_bench.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 19000

    def strict(text):
        def bad(name):
            raise AssertionError(name)
        return json.loads(text, parse_constant=bad)

    text = bench.compact_line(full, "bench_full.json")
    assert "\n" not in text and len(text.encode()) < 8000
    rec = strict(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert rec[key] == pytest.approx(full[key], rel=1e-5) if isinstance(full[key], float) else rec[key] == full[key], key
    assert rec["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5) and rec["roofline"]["bound"] == "hbm"
    assert rec["roofline"]["traffic"] == pytest.approx(full["roofline"]["traffic"], rel=1e-5)
    assert rec["cpu_baseline"]["value"] == pytest.approx(full["cpu_baseline"]["value"], rel=1e-5) and rec["cpu_baseline"]["cores"] == 64
    assert len(rec["config"]["workload"]) <= 300 and len(rec["cpu_baseline"]["sample"]) <= 200
    assert rec["rooflines"]["encoder.all_passes"][1] == pytest.approx(full["rooflines"]["encoder"]["all_passes"]["frac"], rel=1e-5)
    assert rec["full_json"] == "bench_full.json"
    # non-finite numbers never reach the line (json.dumps would print NaN / Infinity, which strict parsers reject)
    full["roofline"]["frac_unique"] = float("nan")
    full["phase_s_per_step"]["rollout"] = float("inf")
    rec = strict(bench.compact_line(full, "x"))
    assert rec["roofline"].get("frac_unique") is None and rec["phase_s_per_step"]["rollout"] is None
    # an oversized optional table is shed before the headline is
    full["rooflines"] = {f"k{i}": {"avg_launch_ms": 1.0, "frac": 0.5} for i in range(2000)}
    text = bench.compact_line(full, "x")
    assert len(text.encode()) < 8000 and "rooflines" not in strict(text) and strict(text)["value"] > 0


def test_native_pcg64_stream_is_numpys():
    """libetm_envgen.so (csrc/envgen.cc, include/etm_envgen.h): every symbol the header declares is exported, and the fill is numpy's
    ``default_rng(seed).random(n, dtype=float32)`` bit for bit -- floats AND the generator state afterwards -- in the AVX-512 form (where
    the host has it) and in the portable scalar form, over lengths around the lane counts, chained calls, several seeds; the
    multi-threaded row fill equals the per-row fills."""
    import ctypes
    from environments import envgen
    here = os.path.dirname(os.path.abspath(__file__))
    header = open(os.path.join(here, "..", "include", "etm_envgen.h")).read()
    declared = set(re.findall(r"\b(etm_[a-z0-9_]+)\s*\(", header))
    assert declared == set(envgen.SIGNATURES), declared ^ set(envgen.SIGNATURES)
    lib = envgen.load(required=True)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.etm_envgen_abi_version() == envgen.ABI_VERSION
    assert lib.etm_pcg64_fill_f32(None, None, 4) != 0 and lib.etm_pcg64_fill_f32(np.zeros(4, np.uint64).ctypes.data, np.zeros(4, np.float32).ctypes.data, 3) != 0
    try:
        for vec in (1, 0):
            lib.etm_envgen_set_vector(vec)
            for seed in (0, 7, (5, 1), 2 ** 40 + 3):
                for n in (0, 2, 14, 16, 18, 62, 64, 66, 126, 128, 130, 510, 1026, 2 * 5 * 5, 3 * 84 * 84):
                    native, ref = np.random.default_rng(seed), np.random.default_rng(seed)
                    st = envgen.state_of(native)
                    for _ in range(3):
                        out = np.full(n, -1.0, dtype=np.float32)
                        assert lib.etm_pcg64_fill_f32(st.ctypes.data, out.ctypes.data, n) == 0
                        assert np.array_equal(out, ref.random(n, dtype=np.float32)), (vec, seed, n)
                        assert np.array_equal(st, envgen.state_of(ref)), (vec, seed, n)
                    envgen.set_state(native, st)          # numpy continues where the native stream stands
                    assert np.array_equal(native.random(9, dtype=np.float32), ref.random(9, dtype=np.float32))
    finally:
        lib.etm_envgen_set_vector(1)
    # rows of a step side by side on the library's threads
    pool = lib.etm_envgen_pool_create(3)
    assert pool
    gens = [np.random.default_rng(100 + w) for w in range(7)]
    states = np.ascontiguousarray(np.stack([envgen.state_of(g) for g in gens]))
    for _ in range(20):
        out = np.empty((7, 3 * 84 * 84), dtype=np.float32)
        assert lib.etm_pcg64_fill_rows_f32(pool, states.ctypes.data, out.ctypes.data, out.shape[1], 7) == 0
        for w, g in enumerate(gens):
            assert np.array_equal(out[w], g.random(out.shape[1], dtype=np.float32))
    lib.etm_envgen_pool_destroy(pool)
