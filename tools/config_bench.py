"""Time full PPO updates for any config of configs/ (whole-path env-steps/s, phases).
python tools/config_bench.py NAME [updates] [key=value ...]   (config keys, dotted for a section: 0/1 -> bool unless the key holds an integer, integers otherwise; ETM_DIAG_LIB=... selects
another build of the library)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the HIP runtime starts: rollout_groups "auto" = 4 (trainer.py)
os.environ.setdefault("ETM_HW_QUEUES_SET_EARLY", "1")   # marker for trainer.py: the line above ran before the HIP runtime started
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import torch
from yaml_parser import YamlParser
from trainer import PPOTrainer
from etm import lib as _etm_lib
if os.environ.get("ETM_DIAG_LIB"):
    _etm_lib.LIB_PATH = os.environ["ETM_DIAG_LIB"]
name = sys.argv[1]
rest = [a for a in sys.argv[2:] if "=" not in a]
n_upd = int(rest[0]) if rest else 3
cfg = YamlParser(os.path.join(REPO, "episodic-transformer-memory-ppo_amd", "configs", name + ".yaml")).get_config()
for kv in (a for a in sys.argv[2:] if "=" in a):
    key, val = kv.split("=")
    node = cfg
    *path, key = key.split(".")                       # environment.copy_threads=8 reaches into a section
    for part in path:
        node = node[part]
    old = node.get(key)
    as_int = isinstance(old, int) and not isinstance(old, bool)       # an integer key stays one (copy_threads=1)
    node[key] = (val == "1") if val in ("0", "1") and not as_int else (int(val) if val.lstrip("-").isdigit() else val)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr = PPOTrainer(cfg, run_id="cfgbench", device=dev, tensorboard=False)
def upd(i):
    lr, beta, clip = tr.schedules(i)
    t0 = time.perf_counter(); tr._sample_training_data(); tr.buffer.prepare_batch_dict(); torch.cuda.synchronize()
    t1 = time.perf_counter(); tr._train_epochs(lr, clip, beta); torch.cuda.synchronize()
    return t1 - t0, time.perf_counter() - t1
upd(0)
r = t = 0.0
for i in range(n_upd):
    a, b = upd(1 + i); r += a; t += b
steps = cfg["n_workers"] * cfg["worker_steps"]
print(f"{name}: {steps * n_upd / (r + t):.0f} env-steps/s  (rollout {r / n_upd:.3f} s, train {t / n_upd:.3f} s per update of {steps} steps; "
      f"D={cfg['transformer']['embed_dim']} L={cfg['transformer']['memory_length']} blocks={cfg['transformer']['num_blocks']} "
      f"gtrxl={cfg['transformer']['gtrxl']} ln={cfg['transformer']['layer_norm']!r})")
tr.close()                          # (worker processes: stops them and unlinks the shared segment)
