import os, sys, time, math
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # as bench.py / train.py: four worker groups (trainer.py, rollout_groups: auto)
os.environ.setdefault("ETM_HW_QUEUES_SET_EARLY", "1")   # marker for trainer.py: the line above ran before the HIP runtime started
sys.path.insert(0, "episodic-transformer-memory-ppo_amd")
import torch
from yaml_parser import YamlParser
from trainer import PPOTrainer
# python tools/soak.py [key=value ...]   e.g. worker_processes=1 envs_per_process=4: the same soak through worker processes + the native driver
over = {}
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    over[k] = (v == "1") if v in ("0", "1") else (int(v) if v.lstrip("-").isdigit() else v)
for name, n in (("synthetic_minigrid", 60), ("synthetic_mortar_gtrxl", 20), ("synthetic_cartpole", 60)):
    cfg = YamlParser(f"episodic-transformer-memory-ppo_amd/configs/{name}.yaml").get_config()
    cfg.update(over)
    torch.manual_seed(0)
    tr = PPOTrainer(cfg, run_id="soak", device=torch.device("cuda", 0), tensorboard=False)
    t0 = time.time(); bad = 0
    for u in range(n):
        lr, beta, clip = tr.schedules(u)
        tr._sample_training_data(); tr.buffer.prepare_batch_dict()
        stats, norms = tr._train_epochs(lr, clip, beta)
        s = torch.stack([torch.as_tensor(r) for r in stats]).float()
        if not torch.isfinite(s).all(): bad += 1
    torch.cuda.synchronize()
    p = torch.cat([q.detach().reshape(-1) for q in tr.model.parameters()])
    native = bool(getattr(tr, "_native_rollout", False))
    print(f"{name} {over if over else ''} (native rollout driver: {native}): {n} updates in {time.time()-t0:.1f} s, non-finite stat rows in {bad} updates, params finite: {bool(torch.isfinite(p).all())}, last loss {float(s[-1][2]):.4f}", flush=True)
    tr.close()
