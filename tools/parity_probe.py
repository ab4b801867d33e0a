"""Which tensors carry the HIP path's gradient-evaluation noise?  (GPU side of the round-4 parity decomposition.)

For a BASELINE-size teacher-forced fixture (tests/golden/rollout_cfg*.npz) the fixture generator records, per optimiser step,
the reference's fp32 gradient AND the gradient of the reference's own loss code evaluated in float64 at the same parameters
("exact").  This tool runs the HIP trainer teacher-forced through the first rollout, evaluates ITS gradient of the first
minibatch (same parameters: the fixture's procedural initial weights) under several kernel selections and prints, per tensor,

    ||reference - exact|| / ||exact||      (the reference's own evaluation noise)
    ||HIP - exact||       / ||exact||

Whole tensors are compared when tools/scratch/parity/ship/xgrad_<case>_u0s0.npz exists (written in the build container by
`tools/parity_decompose.py ship <case>`; scratch, not committed), else the fixture's 256-element samples.

    python tools/parity_probe.py cfg3 [variant,variant,...]   -> gpurun_out/parity/probe_<case>.json (+ optional dumps)
"""
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("ETM_HW_QUEUES_SET_EARLY", "1")   # marker for trainer.py: the line above ran before the HIP runtime started
os.environ.setdefault("ETM_TUNABLE_GEMM", "0")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import detgen as dg  # noqa: E402

VARIANTS = {
    "default": {},
    "separate_heads": {"fused_heads_loss": False, "grouped_dw_train": False, "grouped_colsum_train": False},
    "no_grouped_dw": {"grouped_dw_train": False},
    "no_fused_heads": {"fused_heads_loss": False},
    "library_convs": {"fused_train_encoder": False},
    "unsorted": {"sort_minibatch": False},
    "dense": {"_attention": "dense"},
}


def run(name, variant, full, dump_dir=None):
    from etm import ops
    from trainer import PPOTrainer
    z = np.load(os.path.join(REPO, "tests", "golden", f"rollout_{name}.npz"))
    info = json.loads(str(z["cfg_json"]))
    cfg, envk = info["cfg"], info["env"]
    extra = dict(VARIANTS[variant])
    ops.set_attention_impl(extra.pop("_attention", "folded"))
    cfg = {**cfg, **extra, "environment": {"type": "Synthetic", **envk}}
    tr = PPOTrainer(cfg, run_id="probe", device=torch.device("cuda", 0), tensorboard=False)
    keys = [str(k) for k in z["keys"]]
    shapes = [tuple(int(x) for x in str(s).split(",") if x) for s in z["shapes"]]
    gen = dg.det_state_dict("rollout_" + name, keys, shapes)
    sd = tr.model.state_dict()
    tr.model.load_state_dict({k: (torch.from_numpy(gen[k]) if k in gen else sd[k]) for k in keys})
    tag = "u0/"
    tr._sample_training_data(forced_actions=z[tag + "actions"][:, :, 0])
    tr.buffer.prepare_batch_dict()
    lr, clip, beta = (float(x) for x in z[tag + "hp"])
    mbs = (cfg["n_workers"] * cfg["worker_steps"]) // cfg["n_mini_batch"]
    if dump_dir is not None:
        # the rollout buffer fields the loss reads (rollout forward pass) and the optimisation-phase forward pass of the SAME samples
        # under the same weights: values / log-probs of the two HIP paths against each other and against the reference's rollout
        from etm.ops import WindowSpec
        os.makedirs(dump_dir, exist_ok=True)
        b, flat = tr.buffer, tr.buffer.samples_flat
        with torch.no_grad():
            spec = WindowSpec.from_bank(b.memories, flat["memory_index"], flat["memory_indices"], flat["memory_indices"], flat["memory_mask"])
            logits, value, _ = tr.model.forward_logits(flat["obs"], spec)
            lsm = torch.log_softmax(logits[0], dim=-1)
            logp = lsm.gather(1, flat["actions"][:, :1]).squeeze(1)
        np.savez(os.path.join(dump_dir, f"hip_buffer_{name}_{variant}.npz"), values=b.values.cpu().numpy(), log_probs=b.log_probs.cpu().numpy(),
                 advantages=b.advantages.cpu().numpy(), train_values=value.cpu().numpy().reshape(b.values.shape),
                 train_log_probs=logp.cpu().numpy().reshape(b.values.shape))
    grads = tr.minibatch_gradients(z[tag + "perms"][0][:mbs], clip, beta)
    rows = []
    num_r = num_h = den = num_hr = 0.0
    for k, g in grads.items():
        got = g.detach().cpu().numpy().astype(np.float64)
        if full is not None:
            x, r = full["xgrad/" + k], full["grad/" + k].astype(np.float64)
        else:
            x, r = z[f"{tag}s0/xgrad_sample/{k}"], z[f"{tag}s0/grad_sample/{k}"].astype(np.float64)
            got = dg.sample(got, 256)
        nx = float(np.linalg.norm(x))
        er, eh, ehr = float(np.linalg.norm(r.reshape(-1) - x.reshape(-1))), float(np.linalg.norm(got.reshape(-1) - x.reshape(-1))), \
            float(np.linalg.norm(got.reshape(-1) - r.reshape(-1)))
        rows.append({"tensor": k, "numel": int(g.numel()), "norm": nx, "ref_err": er / max(nx, 1e-300), "hip_err": eh / max(nx, 1e-300),
                     "ratio": eh / max(er, 1e-300)})
        num_r, num_h, num_hr, den = num_r + er * er, num_h + eh * eh, num_hr + ehr * ehr, den + nx * nx
    res = {"case": name, "variant": variant, "whole_tensors": full is not None,
           "all_tensors": {"ref_err": (num_r / den) ** 0.5, "hip_err": (num_h / den) ** 0.5, "hip_vs_ref": (num_hr / den) ** 0.5,
                           "ratio": (num_h / max(num_r, 1e-300)) ** 0.5},
           "tensors": rows}
    print(f"[{name}/{variant}] all tensors: ref-vs-exact {res['all_tensors']['ref_err']:.2e}  HIP-vs-exact {res['all_tensors']['hip_err']:.2e}"
          f"  HIP-vs-ref {res['all_tensors']['hip_vs_ref']:.2e}  ratio {res['all_tensors']['ratio']:.1f}", flush=True)
    for row in sorted(rows, key=lambda r: -r["hip_err"] * r["norm"])[:12]:
        print(f"    {row['tensor']:58s} n={row['numel']:8d} |g|={row['norm']:.2e} ref {row['ref_err']:.2e} hip {row['hip_err']:.2e} x{row['ratio']:.1f}")
    if dump_dir is not None:
        os.makedirs(dump_dir, exist_ok=True)
        np.savez(os.path.join(dump_dir, f"hip_grad_{name}_{variant}.npz"), **{k: g.detach().cpu().numpy() for k, g in grads.items()})
        # parameters after every optimiser step of the first update (eager steps: two warm-up steps precede any capture)
        after = []
        real = tr.optimizer.step

        def step(*a, **k):
            out = real(*a, **k)
            after.append({n: p.detach().cpu().numpy().copy() for n, p in tr.model.named_parameters()})
            return out

        tr.optimizer.step = step
        tr.config["hip_graph_train"] = False
        tr._use_train_graph = False
        tr._train_epochs(lr, clip, beta, perms=z[tag + "perms"])
        for s, d in enumerate(after):
            np.savez(os.path.join(dump_dir, f"hip_params_{name}_{variant}_s{s}.npz"), **d)
    tr.close()
    return res


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    variants = (sys.argv[2] if len(sys.argv) > 2 else "default").split(",")
    dump = os.environ.get("ETM_PROBE_DUMP")
    ship = os.path.join(REPO, "tools", "scratch", "parity", "ship", f"xgrad_{name}_u0s0.npz")
    full = dict(np.load(ship)) if os.path.exists(ship) else None
    out = [run(name, v, full, dump_dir=(os.path.join(REPO, "gpurun_out", "parity") if dump and v == variants[0] else None)) for v in variants]
    os.makedirs(os.path.join(REPO, "gpurun_out", "parity"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "parity", f"probe_{name}.json"), "w") as f:
        json.dump(out, f)
