"""Feasibility probe for a two-group pipelined rollout: do the head graphs of two 16-worker groups overlap on the GPU when
they are replayed on two streams of ONE process?"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import torch
from yaml_parser import YamlParser
from trainer import PPOTrainer
cfg = YamlParser(os.path.join(REPO, "episodic-transformer-memory-ppo_amd", "configs", "synthetic_minigrid.yaml")).get_config()
cfg["n_workers"] = 16
cfg["worker_steps"] = 64
cfg["rollout_groups"] = 1      # each trainer of this probe IS one group
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
trs = []
for g in range(2):
    with torch.cuda.stream(streams[g]):
        tr = PPOTrainer(cfg, run_id=f"g{g}", device=dev, tensorboard=False, first_worker_id=16 * g)
        tr._sample_training_data()
        trs.append(tr)
torch.cuda.synchronize()
def run(which, n=60):
    for tr in trs: tr._groups[0].t_dev.zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for g in which:
            with torch.cuda.stream(streams[g]):
                trs[g]._step_graph[0].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for _ in range(2):
    print(f"head A alone: {run([0]):.1f} us | head B alone: {run([1]):.1f} us | A and B on two streams: {run([0, 1]):.1f} us per pair")
