"""Round 6 soak of the direct observation rows (host writes into device memory, trainer.py / etm/ops.py:host_view): every float of every
rollout's observations as the device holds them afterwards against a host-side copy of what the environments emitted.

    python tools/direct_rows_soak.py [rollouts] [train=0|1]

The environments' step() is wrapped: the rows are produced into a host buffer first (kept per step), then copied into the trainer's
output buffer -- the staging row in device memory -- by numpy; the rollout itself (graphs, groups, fences) is the product's.  train=1 runs
the optimisation phase between rollouts (the staging array's lines are then long gone from the device's caches: the production
pattern); train=0 samples back to back (the copy out of the staging array has just read every line)."""
import os, sys, time
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("ETM_HW_QUEUES_SET_EARLY", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import numpy as np, torch
from trainer import PPOTrainer
from yaml_parser import YamlParser

n_roll = int(sys.argv[1]) if len(sys.argv) > 1 else 12
train = (sys.argv[2].split("=")[1] == "1") if len(sys.argv) > 2 else True
cfg = YamlParser(os.path.join(REPO, "episodic-transformer-memory-ppo_amd", "configs", "synthetic_minigrid.yaml")).get_config()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr = PPOTrainer(cfg, run_id="soak", device=dev, tensorboard=False)
W, S = cfg["n_workers"], cfg["worker_steps"]
shape = tuple(cfg["environment"]["obs_shape"])
emitted = np.zeros((S, W) + shape, dtype=np.float32)        # what the environments emitted for rows 1 .. S - 1 of the staging array
tmp = np.zeros((W,) + shape, dtype=np.float32)
step_of = {}
for gi, g in enumerate(tr._groups):
    orig = g.env.step

    def wrapped(actions, out=None, on_rows=None, orig=orig, g=g):
        t = step_of.get(id(g), 0) + 1
        step_of[id(g)] = t
        res = orig(actions, out=tmp[g.lo:g.hi])
        if t < S:
            emitted[t, g.lo:g.hi] = tmp[g.lo:g.hi]
        np.copyto(out, tmp[g.lo:g.hi])
        return (out,) + tuple(res[1:])
    g.env.step = wrapped
bad_total, t0 = 0, time.time()
for r in range(n_roll):
    step_of.clear()
    first = tr._obs_pin.numpy().copy()                        # observation 0 of this rollout = the last one of the previous
    tr._sample_training_data()
    tr.buffer.prepare_batch_dict()
    emitted[0] = first
    got = tr.buffer.obs                                       # [W, S, ...] on the device
    ref = torch.from_numpy(emitted).to(dev).transpose(0, 1)
    bad = int((got != ref).sum().item())
    bad_total += bad
    if bad:
        print(f"rollout {r}: {bad} floats differ", flush=True)
    if train:
        lr, beta, clip = tr.schedules(r)
        tr._train_epochs(lr, clip, beta)
print(f"direct rows in use: {tr._direct_rows}; {n_roll} rollouts x {W * S} observations x {int(np.prod(shape))} floats "
      f"({'with' if train else 'without'} the optimisation phase in between): {bad_total} floats differ; {time.time() - t0:.1f} s")
tr.close()
