"""Helpers with the upstream names (utils.py): create_env, polynomial_decay, batched_index_select,
process_episode_info, Module."""
import numpy as np
import torch
from torch import nn


def create_env(config: dict, render: bool = False, worker_id: int = 0):
    """One environment instance for ``config["type"]`` (upstream utils.py:10-30).

    "Synthetic" is this build's workload generator (environments/synthetic.py).  The simulator-backed types need
    third-party packages that are not part of this build; they raise a clear ImportError when absent.
    """
    kind = config["type"]
    if kind == "Synthetic":
        from environments.synthetic import SyntheticEnv
        keys = ("obs_shape", "num_actions", "max_episode_steps", "seed", "p_reward", "p_done", "pool")
        kw = {k: config[k] for k in keys if k in config}
        if "obs_shape" in kw:
            kw["obs_shape"] = tuple(kw["obs_shape"])
        return SyntheticEnv(worker_id=worker_id, **kw)
    if kind == "PocMemoryEnv":
        from environments.poc_memory_env import PocMemoryEnv
        return PocMemoryEnv(glob=False, freeze=True, max_episode_steps=32)
    raise ImportError(f"environment type {kind!r} needs its simulator package (gym / gym-minigrid / memory-gym), which is "
                      "outside this build; use type 'Synthetic' with the same observation shape for throughput runs")


def polynomial_decay(initial: float, final: float, max_decay_steps: int, power: float, current_step: int) -> float:
    """Polynomial schedule; ``final`` once current_step exceeds max_decay_steps (strictly) or if nothing decays."""
    if current_step > max_decay_steps or initial == final:
        return final
    frac = 1 - current_step / max_decay_steps
    return (initial - final) * (frac ** power) + final


def batched_index_select(input, dim, index):
    """input [B, ...], index [B, K] -> input gathered along ``dim`` per batch row: [B, K, ...] for dim == 1.

    Kept for API compatibility; the trainer itself never materialises windows (the kernels gather in place)."""
    view = [index.shape[0]] + [1] * (input.dim() - 1)
    view[dim] = index.shape[1]
    expand = list(input.shape)
    expand[dim] = index.shape[1]
    return torch.gather(input, dim, index.reshape(view).expand(expand))


def process_episode_info(episode_info: list) -> dict:
    """mean/std per key of the finished-episode dicts (plus ``success_percent``)."""
    result = {}
    if not episode_info:
        return result
    for key in episode_info[0].keys():
        vals = [info[key] for info in episode_info]
        if key == "success":
            result[key + "_percent"] = np.sum(vals) / len(vals)
        result[key + "_mean"] = np.mean(vals)
        result[key + "_std"] = np.std(vals)
    return result


class Module(nn.Module):
    """nn.Module with gradient norm / mean helpers."""

    def _flat_grads(self):
        grads = [p.grad.view(-1) for _, p in self.named_parameters() if p.grad is not None]
        return torch.cat(grads) if grads else None

    def grad_norm(self):
        g = self._flat_grads()
        return torch.linalg.norm(g).item() if g is not None else None

    def grad_mean(self):
        g = self._flat_grads()
        return torch.mean(g).item() if g is not None else None
