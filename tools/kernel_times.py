"""Per-kernel timings (library's HIP-event facility) for one update of any config.  python tools/kernel_times.py NAME"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import torch
from yaml_parser import YamlParser
from trainer import PPOTrainer
from etm import lib as etm_lib
name = sys.argv[1]
cfg = YamlParser(os.path.join(REPO, "episodic-transformer-memory-ppo_amd", "configs", name + ".yaml")).get_config()
cfg["hip_graph_rollout"] = False      # events cannot be recorded inside a captured graph
dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr = PPOTrainer(cfg, run_id="kt", device=dev, tensorboard=False)
lib = etm_lib.load()
def upd(i, prof):
    lr, beta, clip = tr.schedules(i)
    lib.etm_profile_set_tag(0); tr._sample_training_data(); tr.buffer.prepare_batch_dict()
    lib.etm_profile_set_tag(1); tr._train_epochs(lr, clip, beta); torch.cuda.synchronize()
upd(0, False)
lib.etm_profile_enable(1)
upd(1, True)
lib.etm_profile_enable(0)
res = etm_lib.profile_collect()
for tag, label in ((1, "training"), (0, "rollout")):
    print(f"--- {name} {label}")
    for (t, k), (ms, cnt) in sorted(res.items(), key=lambda kv: -kv[1][0]):
        if t == tag:
            print(f"   {k:24s} launches={cnt:6d} avg_us={ms / cnt * 1e3:9.1f} total_ms={ms:8.2f}")
