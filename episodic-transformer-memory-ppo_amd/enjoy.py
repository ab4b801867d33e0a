"""Run one episode with a trained checkpoint (upstream enjoy.py: --model flag, ``pickle((state_dict, config))`` format of
``PPOTrainer._save_model`` / upstream trainer.py:356-362).

    python enjoy.py --model ./models/run.nn

The model runs on the MI355X (there is no CPU path in this build); the episode loop is upstream's (enjoy.py:62-90): one
environment, one zero-initialised episodic memory [1, T, blocks, D], per step the window rows of ``memory_indices[t]``,
mask row ``clip(t, 0, L-1)``, the model's upstream-signature ``forward`` and the new item written at row t.
"""
import argparse
import pickle

import numpy as np
import torch

from model import ActorCriticModel
from trainer import build_window_tables
from utils import create_env


def init_transformer_memory(trxl_conf: dict, max_episode_steps: int, device):
    """-> (memory [1, T, blocks, D] zeros, memory_mask [L, L] float, memory_indices [T, L] int64), upstream enjoy.py:9-27;
    the tables are the trainer's (bit-exact with upstream trainer.py:78, :88-90)."""
    mask, indices = build_window_tables(trxl_conf["memory_length"], max_episode_steps)
    memory = torch.zeros((1, max_episode_steps, trxl_conf["num_blocks"], trxl_conf["embed_dim"]), dtype=torch.float32, device=device)
    return memory, mask.to(device), indices.to(device)


def run_episode(model, env, config, device, max_steps=None):
    """-> (list of rewards, last info dict).  Sampling uses torch's generator of ``device``."""
    memory, memory_mask, memory_indices = init_transformer_memory(config["transformer"], env.max_episode_steps, device)
    memory_length = config["transformer"]["memory_length"]
    rewards, info, done, t = [], None, False, 0
    obs = env.reset()
    while not done and (max_steps is None or t < max_steps):
        obs_t = torch.tensor(np.expand_dims(obs, 0), dtype=torch.float32, device=device)
        indices = memory_indices[t].unsqueeze(0)
        in_memory = memory[0, indices]                                  # [1, L, blocks, D]
        mask = memory_mask[max(0, min(t, memory_length - 1))].unsqueeze(0)
        if hasattr(env, "render"):
            env.render()
        policy, _value, new_memory = model(obs_t, in_memory, mask, indices)
        memory[:, t] = new_memory.detach()
        action = [branch.sample().item() for branch in policy]
        obs, reward, done, info = env.step(action)
        rewards.append(reward)
        t += 1
    return rewards, info


def main():
    ap = argparse.ArgumentParser(description="Run one episode with a trained model on the MI355X")
    ap.add_argument("--model", default="./models/run.nn", help="Path to the trained model")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("no HIP device visible: the model runs on the MI355X kernels (there is no CPU fallback)")
    device = torch.device("cuda", 0)
    with open(args.model, "rb") as f:
        state_dict, config = pickle.load(f)
    env = create_env(config["environment"], render=True)
    model = ActorCriticModel(config, env.observation_space, (env.action_space.n,), env.max_episode_steps)
    model.load_state_dict(state_dict)
    model.to(device)
    model.eval()
    rewards, info = run_episode(model, env, config, device)
    if hasattr(env, "render"):
        env.render()
    if info:
        print("Episode length: " + str(info["length"]))
        print("Episode reward: " + str(info["reward"]))
    else:
        print("Episode length: " + str(len(rewards)))
        print("Episode reward: " + str(float(np.sum(rewards))))
    env.close()


if __name__ == "__main__":
    main()
