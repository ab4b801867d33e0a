"""Optimisation-phase profile: one rollout (untimed), then N epochs of minibatch updates (BASELINE config 3)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import torch
from yaml_parser import YamlParser
from trainer import PPOTrainer
cfg = YamlParser(os.path.join(REPO, "episodic-transformer-memory-ppo_amd", "configs", "synthetic_minigrid.yaml")).get_config()
cfg["epochs"] = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if os.environ.get("ETM_ENCODER_PRODUCTS"):
    cfg["encoder_products"] = os.environ["ETM_ENCODER_PRODUCTS"]      # (A/B of csrc/conv_b3.hip against the fp32-MFMA encoder kernels)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr = PPOTrainer(cfg, run_id="prof", device=dev, tensorboard=False)
tr._sample_training_data()
tr.buffer.prepare_batch_dict()
tr._train_epochs(3e-4, 0.1, 1e-3)       # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
tr._train_epochs(3e-4, 0.1, 1e-3)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"train: {dt / (cfg['epochs'] * cfg['n_mini_batch']) * 1e3:.3f} ms per minibatch")
