"""Where the part of an update that is NOT rollout steps or minibatch steps goes (round 6): every piece of `_sample_training_data`
around its step loop and of `_train_epochs` around its minibatch loop, timed with a device synchronisation on both sides.

    python tools/update_overheads.py [updates] [config name]
"""
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("ETM_HW_QUEUES_SET_EARLY", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "episodic-transformer-memory-ppo_amd")
for p in (REPO, PKG):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from trainer import PPOTrainer  # noqa: E402
from yaml_parser import YamlParser  # noqa: E402

cfg = YamlParser(os.path.join(PKG, "configs", (sys.argv[2] if len(sys.argv) > 2 else "synthetic_minigrid") + ".yaml")).get_config()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr = PPOTrainer(cfg, run_id="ovh", device=dev, tensorboard=False)
acc = {}


def timed(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def wrapper(*a, **k):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(dev)
        acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, wrapper)


timed(tr.buffer, "begin_rollout")
timed(tr, "_refresh_kv_cache")
timed(tr.model, "refresh_rollout_weights")
timed(tr, "get_last_value")
timed(tr.buffer, "calc_advantages")
timed(tr.buffer, "prepare_batch_dict")
timed(tr, "_bank_with_positions")
timed(tr, "_observations_channels_last")
timed(tr, "_sample_training_data", "rollout_total")
timed(tr, "_train_epochs", "train_total")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for u in range(n + 3):
    if u == 3:
        acc.clear()
    lr, beta, clip = tr.schedules(u)
    tr._sample_training_data()
    tr.buffer.prepare_batch_dict()
    tr._train_epochs(lr, clip, beta)
W, S = cfg["n_workers"], cfg["worker_steps"]
print(f"per update over {n} updates (ms), every item bracketed by device synchronisations:")
for k, v in acc.items():
    print(f"  {k:32s} {1e3 * v / n:8.3f}")
inner = sum(v for k, v in acc.items() if k not in ("rollout_total", "train_total", "prepare_batch_dict", "_bank_with_positions", "_observations_channels_last"))
print(f"  rollout_total - listed pieces = step loop + staging copies: {1e3 * (acc['rollout_total'] - inner) / n:.3f} ms  ({1e6 * (acc['rollout_total'] - inner) / n / S:.1f} us per step)")
tr.close()
