"""torch.profiler op-level view of one training minibatch (BASELINE config 3): which aten ops launch the small kernels."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from yaml_parser import YamlParser
from trainer import PPOTrainer
cfg = YamlParser(os.path.join(REPO, "episodic-transformer-memory-ppo_amd", "configs", "synthetic_minigrid.yaml")).get_config()
cfg["epochs"] = 1
dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr = PPOTrainer(cfg, run_id="prof", device=dev, tensorboard=False)
tr._sample_training_data(); tr.buffer.prepare_batch_dict()
tr._train_epochs(3e-4, 0.1, 1e-3); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr._train_epochs(3e-4, 0.1, 1e-3); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
