"""Where does a rollout step go?  Host-side timers around graph replay / sync / env step (BASELINE config 3)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import numpy as np, torch
from yaml_parser import YamlParser
from trainer import PPOTrainer

cfg = YamlParser(os.path.join(REPO, "episodic-transformer-memory-ppo_amd", "configs", "synthetic_minigrid.yaml")).get_config()
for k in sys.argv[1:]:
    name, val = k.split("=")
    cfg[name] = (val == "1")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr = PPOTrainer(cfg, run_id="prof", device=dev, tensorboard=False)
tr._sample_training_data()          # warm-up: MIOpen find, graph capture
tr.buffer.prepare_batch_dict()
torch.cuda.synchronize()

# instrumented copy of the rollout loop
buf, W, S = tr.buffer, tr.num_workers, cfg["worker_steps"]
stream = torch.cuda.current_stream(dev)
t_replay = t_sync = t_env = t_book = 0.0
buf.begin_rollout(tr._slot_dev)
tr.worker_episode_slot[:] = np.arange(W)
tr._slot_dev.copy_(tr._slot_pin, non_blocking=True)
if tr._use_kv_cache:
    tr._refresh_kv_cache()
tr._t_dev.zero_()
torch.cuda.synchronize()
t00 = time.perf_counter()
for t in range(S):
    a = time.perf_counter()
    if tr._step_graph is not None:
        tr._step_graph[0].replay()
        tr._act_ready.record(stream)
        tr._step_graph[1].replay()
    else:
        with torch.no_grad():
            carry = tr._rollout_step_head()
            tr._act_ready.record(stream)
            tr._rollout_step_tail(carry)
    b = time.perf_counter()
    tr._act_ready.synchronize()
    c = time.perf_counter()
    _, rewards, dones, infos = tr.env.step(tr._act_pin.numpy()[:, 0], out=tr.obs)
    d = time.perf_counter()
    buf.rewards[:, t] = rewards
    buf.dones[:, t] = dones
    tr.worker_current_episode_step += 1
    if dones.any():
        for w in np.flatnonzero(dones):
            tr.worker_current_episode_step[w] = 0
            slot = buf.open_episode()
            tr.worker_episode_slot[w] = slot
            if t < S - 1:
                buf.memory_index_host[w, t + 1:] = slot
    e = time.perf_counter()
    t_replay += b - a; t_sync += c - b; t_env += d - c; t_book += e - d
total = time.perf_counter() - t00
print(f"per step (us): replay-call {t_replay / S * 1e6:.0f}  wait-for-gpu {t_sync / S * 1e6:.0f}  env.step {t_env / S * 1e6:.0f}  bookkeeping {t_book / S * 1e6:.0f}  total {total / S * 1e6:.0f}")
