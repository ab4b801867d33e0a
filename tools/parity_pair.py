"""Two rollout paths of ONE fixture against each other on the device: is a gradient shift a flipped ReLU unit or a defect?

Round 4 closed with the teacher-forced cfg5 fixture at 1.37e-3 parameter-movement error on the default path and 4.1e-5 on the
eager path (profiles/r04/tf_measured.jsonl) after `csrc/conv3_hidden.hip` changed the summation order of `lin_hidden` in the ROLLOUT.
The optimisation phase is the same code on both paths, with the same observations and parameters; what differs is the episode
bank (memory items written by the rollout: different rounding) and the buffer's values / log-probs / advantages.  So every ReLU
whose input depends on the memory window (the blocks' fc layers, the hidden heads) can sit on different sides of its kink in the two
runs, and nothing else can.  This tool shows which it is, without the reference:

  1. runs the HIP trainer teacher-forced through the first rollout under each named path (tests/test_gpu_parity._ROLLOUT_PATHS keys),
  2. evaluates the gradient of the update's first minibatch (eager) while recording, for every relu(linear) layer that runs as its own
     op, the layer's input and its activation pattern, and the input of the fused hidden heads,
  3. per pair (first path vs each other path): the (sample, unit) pairs whose activation differs, with their float64 pre-activations
     under both runs; per gradient tensor: ||A - B|| / ||A||, the two largest singular values of A - B (rank-1 test) and the largest
     element's share of the squared error of the bias tensors (one-hot test),
  4. each path's gradient against the float64 evaluation of the reference's own loss code held by the fixture (64-element samples).

    python tools/parity_pair.py cfg5 default,kslice_hidden,eager [--step=k]   -> gpurun_out/parity/pair_<case>[_stepk].txt / .json
(--step=k: the update's first k optimiser steps are taken first -- eagerly, on the fixture's minibatches -- and the gradient of step k is probed)
"""
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("ETM_HW_QUEUES_SET_EARLY", "1")
os.environ.setdefault("ETM_TUNABLE_GEMM", "0")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"), os.path.join(REPO, "tests", "golden"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import detgen as dg  # noqa: E402

PATHS = {
    "default": {},
    "eager": {"hip_graph_rollout": False},
    "kslice_hidden": {"fused_conv3_hidden": False},
    "multi_launch_blocks": {"fused_rollout_block": False},
    "window_row_stats": {"bank_row_stats": False},      # (optimisation phase: norm_kv statistics per window row)
    "window_row_stats_eager": {"bank_row_stats": False, "hip_graph_rollout": False},
    "fp32_encoder": {"encoder_products": "fp32"},      # (optimisation phase: the fp32-MFMA encoder kernels instead of csrc/conv_b3*.hip)
    "fp32_encoder_eager": {"encoder_products": "fp32", "hip_graph_rollout": False},
}
OUT = []


def say(*a):
    line = " ".join(str(x) for x in a)
    OUT.append(line)
    print(line, flush=True)


STEP = 0      # optimiser steps of the update taken (eagerly, on the fixture's minibatches) before the probed gradient: --step k


def run(name, path):
    from etm import ops
    from trainer import PPOTrainer
    z = np.load(os.path.join(REPO, "tests", "golden", f"rollout_{name}.npz"))
    info = json.loads(str(z["cfg_json"]))
    cfg, envk = info["cfg"], info["env"]
    cfg = {**cfg, **PATHS[path], "environment": {"type": "Synthetic", **envk}}
    tr = PPOTrainer(cfg, run_id="pair", device=torch.device("cuda", 0), tensorboard=False)
    keys = [str(k) for k in z["keys"]]
    shapes = [tuple(int(x) for x in str(s).split(",") if x) for s in z["shapes"]]
    gen = dg.det_state_dict("rollout_" + name, keys, shapes)
    sd = tr.model.state_dict()
    tr.model.load_state_dict({k: (torch.from_numpy(gen[k]) if k in gen else sd[k]) for k in keys})
    tr._sample_training_data(forced_actions=z["u0/actions"][:, :, 0])
    tr.buffer.prepare_batch_dict()
    lr, clip, beta = (float(x) for x in z["u0/hp"])
    mbs = (cfg["n_workers"] * cfg["worker_steps"]) // cfg["n_mini_batch"]
    names = {id(m): n for n, m in tr.model.named_modules()}
    rec = {}
    real_lr, real_heads = ops.linear_relu, ops.heads_ppo_loss

    def rec_linear_relu(lin, x, out=None):
        y = real_lr(lin, x, out)
        if torch.is_grad_enabled() and id(lin) in names:
            rec[names[id(lin)]] = {"x": x.detach().double().cpu(), "on": (y.detach() > 0).cpu(), "w": lin.weight.detach().double().cpu(),
                                   "b": lin.bias.detach().double().cpu()}
        return y

    def rec_heads(h, lin_policy, lin_value, *a, **k):
        for lin in (lin_policy, lin_value):
            x64 = h.detach().double()
            pre32 = torch.nn.functional.linear(h.detach(), lin.weight.detach(), lin.bias.detach())
            rec[names[id(lin)]] = {"x": x64.cpu(), "on": (pre32 > 0).cpu(), "w": lin.weight.detach().double().cpu(), "b": lin.bias.detach().double().cpu(),
                                   "note": "activation pattern recomputed by a library product (the fused heads + loss pass keeps none)"}
        return real_heads(h, lin_policy, lin_value, *a, **k)

    ops.linear_relu, ops.heads_ppo_loss = rec_linear_relu, rec_heads
    n_mb = cfg["n_mini_batch"]
    p0 = {k: v.detach().double().cpu().clone() for k, v in tr.model.named_parameters()}
    for st in range(STEP):          # the update's first STEP optimiser steps, in the fixture's minibatch order
        perm = z["u0/perms"][st // n_mb]
        idx = torch.as_tensor(perm[(st % n_mb) * mbs: (st % n_mb + 1) * mbs], device=tr.device, dtype=torch.long).sort().values
        tr._train_mini_batch(tr.buffer.gather(idx), lr, clip, beta)
    probe_perm = z["u0/perms"][STEP // n_mb][(STEP % n_mb) * mbs: (STEP % n_mb + 1) * mbs]
    try:
        grads = tr.minibatch_gradients(probe_perm, clip, beta)
    finally:
        ops.linear_relu, ops.heads_ppo_loss = real_lr, real_heads
    b = tr.buffer
    fields = {f: getattr(b, f).detach().double().cpu().numpy() for f in ("values", "log_probs", "advantages")}
    res = {"grads": {k: g.detach().double().cpu() for k, g in grads.items()}, "rec": rec, "fields": fields,
           "params": {k: v.detach().double().cpu().clone() for k, v in tr.model.named_parameters()}, "params0": p0}
    # against the float64 evaluation of the reference's loss code (fixture samples of step 0 of update 0)
    pnames = [str(k) for k in z["param_keys"]]
    xs, rs, xnorm = z[f"u0/s{STEP}/xgrad_samples"], z[f"u0/s{STEP}/grad_samples"], z[f"u0/s{STEP}/xgrad_norm"]
    num_h = num_r = den = 0.0
    rows = []
    for i, k in enumerate(pnames):
        x = xs[i][~np.isnan(xs[i])].astype(np.float64)
        r = rs[i][~np.isnan(rs[i])].astype(np.float64)
        got = dg.sample(grads[k].cpu().numpy(), 64).astype(np.float64)
        scale = float(xnorm[i]) * (x.size / grads[k].numel()) ** 0.5
        eh, er = float(np.linalg.norm(got - x)), float(np.linalg.norm(r - x))
        rows.append((k, eh / max(scale, 1e-300), er / max(scale, 1e-300)))
        num_h, num_r, den = num_h + eh * eh, num_r + er * er, den + float(np.sum(x ** 2))
    res["vs_exact"] = {"hip": (num_h / den) ** 0.5, "ref": (num_r / den) ** 0.5, "rows": rows}
    say(f"[{name}/{path}] gradient of optimiser step {STEP} vs the float64 evaluation AT THE REFERENCE'S parameters of that step (fixture samples): HIP {res['vs_exact']['hip']:.2e}, "
        f"reference {res['vs_exact']['ref']:.2e}")
    for k, eh, er in sorted(rows, key=lambda t: -t[1])[:5]:
        say(f"      {k:56s} HIP {eh:.2e}   reference {er:.2e}")
    tr.close()
    return res


def compare(name, pa, ra, pb, rb):
    say(f"\n==== {name}: {pa} vs {pb}")
    for f in ("values", "log_probs", "advantages"):
        d = np.abs(ra["fields"][f] - rb["fields"][f])
        say(f"  buffer.{f}: max |A - B| {d.max():.2e}, rms {np.sqrt((d ** 2).mean()):.2e}")
    if STEP:
        num = sum(float((ra["params"][k] - rb["params"][k]).norm()) ** 2 for k in ra["params"])
        den = sum(float((ra["params"][k] - ra["params0"][k]).norm()) ** 2 for k in ra["params"])
        say(f"  parameters after {STEP} optimiser step(s): ||A - B|| / ||A - initial|| = {(num / max(den, 1e-300)) ** 0.5:.2e} (whole tensors)")
    flips = []
    for lname in ra["rec"]:
        A, B = ra["rec"][lname], rb["rec"].get(lname)
        if B is None:
            continue
        diff = (A["on"] != B["on"]).nonzero()
        dx = float((A["x"] - B["x"]).norm() / max(float(A["x"].norm()), 1e-300))
        say(f"  layer {lname:44s} input ||A - B|| / ||A|| {dx:.2e}; units with a different activation: {len(diff)} of {A['on'].numel()}")
        for n, u in diff.tolist()[:8]:
            pa64 = float(A["x"][n] @ A["w"][u] + A["b"][u])
            pb64 = float(B["x"][n] @ B["w"][u] + B["b"][u])
            say(f"      sample {n} (sorted minibatch position), unit {u}: float64 pre-activation {pa64:+.3e} under {pa}, {pb64:+.3e} under {pb}"
                f"  (active: {bool(A['on'][n, u])} / {bool(B['on'][n, u])})")
            flips.append({"layer": lname, "sample": n, "unit": u, "pre_a": pa64, "pre_b": pb64})
    num = den = 0.0
    rows = []
    for k, ga in ra["grads"].items():
        d = ga - rb["grads"][k]
        e, nrm = float(d.norm()), float(ga.norm())
        num, den = num + e * e, den + nrm * nrm
        row = {"tensor": k, "rel": e / max(nrm, 1e-300)}
        if d.dim() >= 2 and e > 0:
            s = torch.linalg.svdvals(d.reshape(d.shape[0], -1).cuda()).cpu()
            row["s1_over_s0"] = float(s[1] / s[0]) if s.numel() > 1 else 0.0
            rowsq = (d.reshape(d.shape[0], -1) ** 2).sum(1)
            row["top_row_share"], row["top_row"] = float(rowsq.max() / rowsq.sum()), int(rowsq.argmax())
        elif e > 0:
            sq = d.reshape(-1) ** 2
            row["top_elem_share"], row["top_elem"] = float(sq.max() / sq.sum()), int(sq.argmax())
        rows.append(row)
    say(f"  gradient of the first minibatch, all tensors: ||A - B|| / ||A|| = {(num / den) ** 0.5:.2e}")
    for row in sorted(rows, key=lambda r: -r["rel"])[:14]:
        extra = ""
        if "s1_over_s0" in row:
            extra = f"second / first singular value of A - B {row['s1_over_s0']:.1e}; row {row['top_row']} holds {row['top_row_share']:.3f} of the squared error"
        elif "top_elem" in row:
            extra = f"element {row['top_elem']} holds {row['top_elem_share']:.3f} of the squared error"
        say(f"      {row['tensor']:56s} {row['rel']:.2e}  {extra}")
    return {"pair": [pa, pb], "flips": flips, "all": (num / den) ** 0.5, "tensors": rows}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--step")]
    for a in sys.argv[1:]:
        if a.startswith("--step="):
            STEP = int(a.split("=")[1])
    name = args[0] if args else "cfg5"
    paths = (args[1] if len(args) > 1 else "default,kslice_hidden,eager").split(",")
    results = {p: run(name, p) for p in paths}
    summary = [compare(name, paths[0], results[paths[0]], p, results[p]) for p in paths[1:]]
    if len(paths) > 2:
        summary.append(compare(name, paths[1], results[paths[1]], paths[2], results[paths[2]]))
    os.makedirs(os.path.join(REPO, "gpurun_out", "parity"), exist_ok=True)
    tag = f"{name}_step{STEP}" if STEP else name
    with open(os.path.join(REPO, "gpurun_out", "parity", f"pair_{tag}.txt"), "w") as f:
        f.write("\n".join(OUT) + "\n")
    with open(os.path.join(REPO, "gpurun_out", "parity", f"pair_{tag}.json"), "w") as f:
        json.dump({"case": name, "vs_exact": {p: {"hip": r["vs_exact"]["hip"], "ref": r["vs_exact"]["ref"]} for p, r in results.items()},
                   "pairs": summary}, f)
