"""Every replay of the captured optimisation step against an eager evaluation of the SAME step (round 5).

A trainer takes its minibatch steps through the captured graph as always.  After every replay the gradient the graph left in the
flat arena is kept, the parameters are put back to what they were before the step, the same minibatch is evaluated eagerly
(`_train_body_a`: the code the graph was captured from) and the two gradients are compared tensor by tensor; then the parameters
of the replayed step are restored and training goes on.  Same kernels in the same order: the difference is rounding-level (library
GEMMs may pick another algorithm outside capture) unless a node of the graph is not replay-safe -- DESIGN.md section 4: torch's
column sum was not, from replay ~300 on, and returned garbage for one tensor.

    python tools/graph_replay_soak.py [updates] [layout ...]        layouts: trxl_post (visual), gtrxl_pre (vector), gtrxl_pre_visual,
                                                                    or the name of a configs/*.yaml (synthetic_minigrid = BASELINE config 3,
                                                                    synthetic_mortar_gtrxl = config 5's shape, synthetic_cartpole = config 2's)
    python tools/graph_replay_soak.py 120 gtrxl_pre --framework-bias-grad
        positive control: fc_out as a plain nn.Linear again (its bias gradient = torch's column sum inside the captured step) on
        minibatches of 1,024 samples -- the check must report the replays in which that one tensor is wrong
"""
import os
import sys

os.environ.setdefault("ETM_TUNABLE_GEMM", "0")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "episodic-transformer-memory-ppo_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

BASE = dict(gamma=0.99, lamda=0.95, updates=1, epochs=2, n_workers=8, worker_steps=32, n_mini_batch=2, value_loss_coefficient=0.5,
            hidden_layer_size=128, max_grad_norm=0.5, tunable_gemm=False,
            learning_rate_schedule=dict(initial=3e-4, final=3e-4, power=1.0, max_decay_steps=10),
            beta_schedule=dict(initial=1e-3, final=1e-3, power=1.0, max_decay_steps=10),
            clip_range_schedule=dict(initial=0.1, final=0.1, power=1.0, max_decay_steps=10))
LAYOUTS = {
    "trxl_post": dict(environment=dict(type="Synthetic", obs_shape=[3, 84, 84], num_actions=3, max_episode_steps=24, seed=0, p_done=0.05, pool=8),
                      transformer=dict(num_blocks=2, embed_dim=128, num_heads=4, memory_length=16, positional_encoding="relative",
                                       layer_norm="post", gtrxl=False, gtrxl_bias=0.0)),
    "gtrxl_pre": dict(environment=dict(type="Synthetic", obs_shape=[4], num_actions=2, max_episode_steps=24, seed=0, p_done=0.05, pool=8),
                      transformer=dict(num_blocks=4, embed_dim=128, num_heads=4, memory_length=16, positional_encoding="",
                                       layer_norm="pre", gtrxl=True, gtrxl_bias=0.0)),
    "gtrxl_pre_visual": dict(environment=dict(type="Synthetic", obs_shape=[3, 84, 84], num_actions=4, max_episode_steps=48, seed=0, p_done=0.05, pool=8),
                             transformer=dict(num_blocks=2, embed_dim=384, num_heads=4, memory_length=32, positional_encoding="relative",
                                              layer_norm="pre", gtrxl=True, gtrxl_bias=0.0)),
}


def run(layout, updates, quiet=False, framework_bias_grad=False):
    from trainer import PPOTrainer
    dev = torch.device("cuda", 0)
    if layout in LAYOUTS:
        cfg = dict(BASE, **LAYOUTS[layout])
    else:
        from yaml_parser import YamlParser
        cfg = YamlParser(os.path.join(REPO, "episodic-transformer-memory-ppo_amd", "configs", layout + ".yaml")).get_config()
        cfg["tunable_gemm"] = False
    if framework_bias_grad:
        from etm import ops
        ops.linear_bias = lambda lin, x: lin(x)
        cfg.update(worker_steps=256)                # minibatches of 1,024 rows: torch sums them in two stages (semaphore)
        cfg["environment"] = dict(cfg["environment"], max_episode_steps=300)
    torch.manual_seed(0)
    tr = PPOTrainer(cfg, run_id=f"soak_{layout}", device=dev, tensorboard=False)
    names = [k for k, p in tr.model.named_parameters() if p.requires_grad]
    views = tr._grad_views
    assert len(names) == len(views) == len(tr.params)
    mbs = tr.buffer.batch_size // cfg["n_mini_batch"]
    replays, worst, worst_at, loud = 0, 0.0, None, 0
    for u in range(updates):
        tr._sample_training_data()
        tr.buffer.prepare_batch_dict()
        with torch.no_grad():
            tr._bank_pos, tr._obs_train = tr._bank_with_positions(), tr._observations_channels_last()
        for e in range(cfg["epochs"]):
            perm = torch.randperm(tr.buffer.batch_size, device=dev).view(-1, mbs).sort(dim=1).values
            for idx in perm:
                before = [p.detach().clone() for p in tr.params]
                was_captured = tr._train_graph is not None
                tr._train_step_graph(idx, 3e-4, 0.1, 1e-3, bool(cfg.get("monitor_gradients", True)))
                if not was_captured:
                    continue                                    # an eager warm-up step or the capture itself
                replays += 1
                g_graph = [v.clone() for v in views]
                after = [p.detach().clone() for p in tr.params]
                with torch.no_grad():
                    torch._foreach_copy_([p.data for p in tr.params], before)
                tr._train_body_a(tr._tg_idx, 0.1, 1e-3, tr._tg_stats3)
                # (the step's clip + AdamW kernel leaves the arena scaled by the clip coefficient: one factor for all tensors)
                dots = torch.stack([torch.stack(((gg * ge).sum(), (ge * ge).sum())) for gg, ge in zip(g_graph, views)]).double()
                alpha = float((dots[:, 0] / dots[:, 1].clamp_min(1e-300)).median())          # (median: one wrong tensor must not move it)
                for k, gg, ge in zip(names, g_graph, views):
                    den = abs(alpha) * float(ge.norm())
                    d = float((gg - alpha * ge).norm()) / den if den > 0 else float((gg - alpha * ge).norm())
                    if d > worst:
                        worst, worst_at = d, (replays, k)
                    if d > 1e-3:
                        loud += 1
                        if not quiet and loud <= 5:
                            print(f"[{layout}] replay {replays}: {k}: graph gradient differs from the eager one by {d:.3e} of its norm", flush=True)
                with torch.no_grad():
                    torch._foreach_copy_([p.data for p in tr.params], after)
        tr._bank_pos = tr._row_stats = None
    assert tr._train_graph is not None
    tr.close()
    return dict(layout=layout, updates=updates, replays=replays, worst=worst, worst_at=worst_at, loud=loud)


if __name__ == "__main__":
    args = sys.argv[1:]
    updates = int(args[0]) if args and args[0].isdigit() else 150
    layouts = [a for a in args if a in LAYOUTS or a.startswith("synthetic_")] or list(LAYOUTS)
    control = "--framework-bias-grad" in args
    for layout in layouts:
        r = run(layout, updates, framework_bias_grad=control)
        print(("POSITIVE CONTROL (fc_out bias gradient by torch's column sum) " if control else "") + f"{layout}: {r['replays']} replays of the captured step checked against eager evaluations: worst tensor difference {r['worst']:.3e} of its norm "
              + (f"(replay {r['worst_at'][0]}, {r['worst_at'][1]})" if r["worst_at"] else "(every tensor of every replay identical up to the clip factor)")
              + f"; (replay, tensor) pairs above 1e-3: {r['loud']}", flush=True)
