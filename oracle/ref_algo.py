"""Oracle (CPU, fp32): GAE, PPO loss, schedules and a whole-update CPU trainer.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Pinned by
``tests/golden/*.npz`` via ``tests/test_oracle_golden.py``.
Citations are into ``/root/reference``.
"""
import time

import numpy as np
import torch

from . import ref_model as rm


def polynomial_decay(initial: float, final: float, max_decay_steps: int, power: float, current_step: int) -> float:
    """utils.py:32-50 (note the strict ``>``: at step == max the formula itself yields ``final``)."""
    if current_step > max_decay_steps or initial == final:
        return final
    return (initial - final) * ((1 - current_step / max_decay_steps) ** power) + final


def gae(rewards, dones, values, last_value, gamma: float, lamda: float):
    """buffer.py:95-113.  rewards/dones/values [W,S], last_value [W] -> advantages [W,S].

    Same operation order as the reference loop (so it is the bit-level
    definition the HIP scan is compared against).
    """
    rewards = torch.as_tensor(rewards, dtype=torch.float32)
    alive = torch.as_tensor(dones).logical_not()
    values = torch.as_tensor(values, dtype=torch.float32)
    adv = torch.zeros_like(values)
    nxt_value = torch.as_tensor(last_value, dtype=torch.float32)
    nxt_adv = 0
    for t in range(values.shape[1] - 1, -1, -1):
        nxt_value = nxt_value * alive[:, t]
        nxt_adv = nxt_adv * alive[:, t]
        delta = rewards[:, t] + gamma * nxt_value - values[:, t]
        nxt_adv = delta + gamma * lamda * nxt_adv
        adv[:, t] = nxt_adv
        nxt_value = values[:, t]
    return adv


def ppo_loss(logits_list, value, actions, old_log_probs, advantages, old_values, clip_range, vf_coef, beta):
    """trainer.py:276-304, 315-316.

    logits_list: one [N, A_b] tensor per action branch; actions [N,B] int64;
    old_log_probs [N,B]; advantages/old_values/value [N].
    Returns (loss, stats[6]) with stats = (policy, value, loss, entropy, kl, clip_fraction),
    the order of trainer.py:318-323.
    """
    logp, ent = [], []
    for b, logits in enumerate(logits_list):
        lsm = torch.log_softmax(logits, dim=-1)
        logp.append(lsm.gather(1, actions[:, b:b + 1]).squeeze(1))
        ent.append(-(lsm.exp() * lsm).sum(-1))
    logp = torch.stack(logp, dim=1)
    entropy = torch.stack(ent, dim=1).sum(1).reshape(-1)
    norm_adv = (advantages - advantages.mean()) / (advantages.std() + 1e-8)  # unbiased std, per minibatch
    norm_adv = norm_adv.unsqueeze(1).repeat(1, len(logits_list))
    log_ratio = logp - old_log_probs
    ratio = torch.exp(log_ratio)
    surr1 = ratio * norm_adv
    surr2 = torch.clamp(ratio, 1.0 - clip_range, 1.0 + clip_range) * norm_adv
    policy = torch.min(surr1, surr2).mean()
    ret = old_values + advantages
    clipped = old_values + (value - old_values).clamp(min=-clip_range, max=clip_range)
    vf = torch.max((value - ret) ** 2, (clipped - ret) ** 2).mean()
    ent_bonus = entropy.mean()
    loss = -(policy - vf_coef * vf + beta * ent_bonus)
    kl = ((ratio - 1.0) - log_ratio).mean()
    clip_frac = (abs(ratio - 1.0) > clip_range).float().mean()
    return loss, torch.stack([policy.detach(), vf.detach(), loss.detach(), ent_bonus.detach(), kl.detach(), clip_frac])


class OracleTrainer:
    """Whole-update CPU restatement of ``PPOTrainer`` (trainer.py:17-323) over a vectorised env.

    Used for (1) teacher-forced end-to-end parity against the MI355X trainer and
    (2) the ``cpu_baseline`` leg of ``bench.py``.  ``env`` follows the
    ``VecEnv`` protocol of the build (reset() -> obs[W,...];
    step(actions[W]) -> obs, rewards, dones, infos) which is the batched form of
    the reference's per-worker pipe protocol (worker.py:20-34).
    """

    def __init__(self, config: dict, env, state_dict=None, seed: int = 0):
        self.cfg = config
        self.env = env
        t = config["transformer"]
        self.W, self.S = config["n_workers"], config["worker_steps"]
        self.L, self.nb, self.D = t["memory_length"], t["num_blocks"], t["embed_dim"]
        self.T = env.max_episode_steps
        self.obs_shape = tuple(env.observation_space_shape)
        self.branches = (env.num_actions,)  # Q8: always one branch (trainer.py:47)
        sd = state_dict if state_dict is not None else rm.init_state_dict(config, self.obs_shape, self.branches, self.T, seed)
        self.sd = {k: v.clone().float().requires_grad_(v.is_floating_point() and not k.endswith("inv_freqs"))
                   for k, v in sd.items()}
        self.params = [v for v in self.sd.values() if v.requires_grad]
        self.opt = torch.optim.AdamW(self.params, lr=config["learning_rate_schedule"]["initial"])
        self.mask_table, self.index_table = rm.window_tables(self.L, self.T)
        self.memory = torch.zeros((self.W, self.T, self.nb, self.D))
        self.ep_step = torch.zeros((self.W,), dtype=torch.int64)
        self.obs = np.asarray(env.reset(), dtype=np.float32)
        self.gen = torch.Generator().manual_seed(seed)

    # -- forward helper
    def _forward(self, obs, window, mask, indices):
        return rm.actor_critic(self.sd, self.cfg, obs, window, mask, indices, self.T)

    def sample(self, forced_actions=None):
        """trainer.py:145-225.  Returns the rollout dict (reference buffer fields, flattened later)."""
        W, S, L = self.W, self.S, self.L
        buf = {
            "obs": torch.zeros((W, S) + self.obs_shape),
            "actions": torch.zeros((W, S, 1), dtype=torch.int64),
            "log_probs": torch.zeros((W, S, 1)),
            "values": torch.zeros((W, S)),
            "rewards": np.zeros((W, S), dtype=np.float32),
            "dones": np.zeros((W, S), dtype=bool),
            "memory_mask": torch.zeros((W, S, L), dtype=torch.bool),
            "memory_index": torch.zeros((W, S), dtype=torch.int64),
            "memory_indices": torch.zeros((W, S, L), dtype=torch.int64),
        }
        episodes = [self.memory[w] for w in range(W)]  # views of the live memories (trainer.py:154)
        buf["memory_index"][:] = torch.arange(W)[:, None]
        infos = []
        for t in range(S):
            with torch.no_grad():
                obs_t = torch.from_numpy(self.obs.copy())
                buf["obs"][:, t] = obs_t
                buf["memory_mask"][:, t] = self.mask_table[torch.clip(self.ep_step, 0, L - 1)].bool()
                buf["memory_indices"][:, t] = self.index_table[self.ep_step]
                window = rm.gather_window(self.memory, buf["memory_indices"][:, t])
                logits, value, item = self._forward(obs_t, window, buf["memory_mask"][:, t], buf["memory_indices"][:, t])
                self.memory[torch.arange(W), self.ep_step] = item
                lsm = torch.log_softmax(logits[0], dim=-1)
                if forced_actions is None:
                    act = torch.multinomial(lsm.exp(), 1, generator=self.gen).squeeze(1)
                else:
                    act = torch.as_tensor(forced_actions[:, t]).reshape(W).long()
                buf["actions"][:, t, 0] = act
                buf["log_probs"][:, t, 0] = lsm.gather(1, act[:, None]).squeeze(1)
                buf["values"][:, t] = value
            obs, rew, done, info = self.env.step(act.numpy())
            buf["rewards"][:, t] = rew
            buf["dones"][:, t] = done
            for w in range(W):
                if done[w]:
                    self.ep_step[w] = 0
                    infos.append(info[w])
                    slot = int(buf["memory_index"][w, t])
                    episodes[slot] = episodes[slot].clone()      # freeze the finished episode (trainer.py:205-206)
                    self.memory[w] = 0.0
                    if t < S - 1:
                        episodes.append(self.memory[w])
                        buf["memory_index"][w, t + 1:] = len(episodes) - 1
                else:
                    self.ep_step[w] += 1
            self.obs = np.asarray(obs, dtype=np.float32)
        buf["last_value"] = self.last_value(buf)
        buf["advantages"] = gae(buf["rewards"], buf["dones"], buf["values"], buf["last_value"],
                                self.cfg["gamma"], self.cfg["lamda"])
        buf["memories"] = torch.stack(episodes, dim=0)
        buf["episode_infos"] = infos
        return buf

    def last_value(self, buf):
        """trainer.py:227-237 incl. quirk Q5 (different window; positions from the last stored step)."""
        L = self.L
        rows = []
        for w in range(self.W):
            s, e = rm.last_value_window(int(self.ep_step[w]), L)
            rows.append(torch.arange(s, e))
        idx = torch.stack(rows).long()
        with torch.no_grad():
            window = rm.gather_window(self.memory, idx)
            mask = self.mask_table[torch.clip(self.ep_step, 0, L - 1)]
            _, v, _ = self._forward(torch.from_numpy(self.obs.copy()), window, mask, buf["memory_indices"][:, -1])
        return v

    @staticmethod
    def flatten(buf):
        keys = ("actions", "values", "log_probs", "advantages", "obs", "memory_mask", "memory_index", "memory_indices")
        return {k: buf[k].reshape(buf[k].shape[0] * buf[k].shape[1], *buf[k].shape[2:]) for k in keys}

    def train_minibatch(self, flat, memories, idx, lr, clip, beta):
        """trainer.py:258-323 on the sample rows ``idx``."""
        ep_mem = memories[flat["memory_index"][idx]]
        window = rm.gather_window(ep_mem, flat["memory_indices"][idx])
        logits, value, _ = self._forward(flat["obs"][idx], window, flat["memory_mask"][idx], flat["memory_indices"][idx])
        loss, stats = ppo_loss(logits, value, flat["actions"][idx], flat["log_probs"][idx], flat["advantages"][idx],
                               flat["values"][idx], clip, self.cfg["value_loss_coefficient"], beta)
        for pg in self.opt.param_groups:
            pg["lr"] = lr
        self.opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.params, max_norm=self.cfg["max_grad_norm"])
        self.opt.step()
        return stats.numpy()

    def update(self, update_idx: int = 0, forced_actions=None, perms=None):
        """One full PPO update.  ``perms``: optional list (one per epoch) of sample permutations."""
        c = self.cfg
        sched = lambda s: polynomial_decay(s["initial"], s["final"], s["max_decay_steps"], s["power"], update_idx)
        lr, beta, clip = sched(c["learning_rate_schedule"]), sched(c["beta_schedule"]), sched(c["clip_range_schedule"])
        t0 = time.perf_counter()
        buf = self.sample(forced_actions)
        t1 = time.perf_counter()
        flat = self.flatten(buf)
        batch = self.W * self.S
        mb = batch // c["n_mini_batch"]
        stats = []
        for ep in range(c["epochs"]):
            perm = torch.as_tensor(perms[ep]) if perms is not None else torch.randperm(batch, generator=self.gen)
            for start in range(0, batch, mb):
                stats.append(self.train_minibatch(flat, buf["memories"], perm[start:start + mb], lr, clip, beta))
        t2 = time.perf_counter()
        return buf, np.asarray(stats), {"rollout_s": t1 - t0, "train_s": t2 - t1}
