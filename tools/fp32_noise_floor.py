"""How far apart are two CORRECT fp32 evaluations of the same PPO update at a BASELINE model size?

The teacher-forced fixtures compare post-update parameters with the reference's.  From the SECOND AdamW step on, the movement of a
parameter is lr * m_hat / (sqrt(v_hat) + eps): scale-free in the gradient, so an element whose gradient is small against the
evaluation noise of that gradient moves by +-lr whatever its magnitude -- the relative error of the MOVEMENT of such elements is O(1)
however accurate the gradient is relative to its tensor's norm.  This tool measures that floor without any GPU: it runs the CPU
oracle (oracle/ref_algo.py, test infrastructure) on a fixture with the forward / backward pass in float64 (exact up to 1e-16)
and the reference's fp32 clipping + AdamW arithmetic on fp32 master parameters, and reports the parameter-movement error of
the REFERENCE's own fp32 result (the fixture) against it, per update, in the metric of
tests/test_gpu_parity.py::movement_error.  The GPU path cannot be expected to sit closer to the reference than the reference sits
to the exact result.

    python tools/fp32_noise_floor.py cfg3 [threads]        -> profiles/r03/fp32_noise_floor_<name>.txt
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"), os.path.join(REPO, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import detgen as dg  # noqa: E402
from environments.synthetic import SyntheticVecEnv  # noqa: E402
from oracle import ref_algo as ra  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
torch.set_num_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
z = np.load(os.path.join(REPO, "tests", "golden", f"rollout_{name}.npz"))
info = json.loads(str(z["cfg_json"]))
cfg, envk = info["cfg"], info["env"]
keys = [str(k) for k in z["keys"]]
shapes = [tuple(int(x) for x in str(s).split(",") if x) for s in z["shapes"]]
torch.set_default_dtype(torch.float64)
env = SyntheticVecEnv(cfg["n_workers"], **{**envk, "obs_shape": tuple(envk["obs_shape"])})
sd0 = {k: torch.from_numpy(v).double() for k, v in dg.det_state_dict("rollout_" + name, keys, shapes).items()}
D = cfg["transformer"]["embed_dim"]
sd0["transformer.pos_embedding.inv_freqs"] = 1e4 ** (-torch.arange(0, D, 2.0) / D)
tr = ra.OracleTrainer(cfg, env, state_dict=sd0, seed=0)
# fp32 master parameters and fp32 AdamW / clipping exactly as in the reference (torch.optim.AdamW's single-tensor path on fp32
# tensors): ONLY the forward / backward evaluation is exact -- the parameters are up-cast inside the graph, so `.grad` is the
# float64 gradient rounded once to fp32.  What is left between this run and the fixture is the reference's own fp32
# forward / backward evaluation noise, propagated through the same optimiser arithmetic.
tr.memory = tr.memory.double()
from oracle import ref_model as rm  # noqa: E402
tr._forward = lambda obs, window, mask, idx: rm.actor_critic({k: (v.double() if v.is_floating_point() else v) for k, v in tr.sd.items()},
                                                             tr.cfg, obs.double(), window.double(), mask, idx, tr.T)
prev = {k: dg.sample(v.numpy(), 64).astype(np.float64) for k, v in sd0.items() if not k.endswith("inv_freqs")}
lines = [f"fixture rollout_{name}: the reference's fp32 result against the float64 evaluation of the same update (CPU oracle, teacher-forced)"]
for upd in range(cfg["updates"]):
    tag = f"u{upd}/"
    buf, stats, _ = tr.update(upd, forced_actions=z[tag + "actions"][:, :, 0], perms=z[tag + "perms"])
    fwd_err = {f: float((np.abs(np.asarray(buf[f], dtype=np.float64).reshape(z[tag + f].shape) - z[tag + f]) / np.maximum(1.0, np.abs(z[tag + f]))).max())
               for f in ("values", "log_probs", "advantages")}
    num = den = worst = 0.0
    worst_key = ""
    for k in prev:
        got = z[tag + "sd_after_sample/" + k].astype(np.float64)           # the reference's fp32 parameters after the update
        ref = dg.sample(tr.sd[k].detach().numpy(), 64).astype(np.float64)  # float64 evaluation
        mv, err = ref - prev[k], got - ref
        n_mv = float(np.linalg.norm(mv))
        if n_mv > 0 and float(np.linalg.norm(err)) / n_mv > worst:
            worst, worst_key = float(np.linalg.norm(err)) / n_mv, k
        num, den = num + float(np.sum(err ** 2)), den + float(np.sum(mv ** 2))
        prev[k] = ref
    gerr = None
    if tag + "grad0_norm/" + next(iter(prev)) in z:
        pass
    line = (f"update {upd}: forward (scaled) " + ", ".join(f"{f} {v:.1e}" for f, v in fwd_err.items()) +
            f"; loss statistics max rel {float((np.abs(stats - z[tag + 'stats']) / (np.abs(z[tag + 'stats']) + 1e-3)).max()):.1e}"
            f"; parameter movement: all tensors {np.sqrt(num / max(den, 1e-300)):.2e}, worst tensor {worst:.2e} ({worst_key})")
    print(line, flush=True)
    lines.append(line)
open(os.path.join(REPO, "profiles", "r03", f"fp32_noise_floor_{name}.txt"), "w").write("\n".join(lines) + "\n")
