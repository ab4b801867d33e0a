"""HBM roofline of the two scan/loss kernels at config size and at a scaled size (SURVEY.md section 8d asks for both: at
config size they are launch-latency-bound).

    python tools/scan_roofline.py            # prints one line per (kernel, size): us per launch, GB/s, fraction of 8 TB/s

GAE: 13 algorithmic bytes per (worker, step) -- rewards 4 + values 4 + dones 1 read, advantages 4 written.
PPO loss: 28 + 8 A bytes per sample (logits 4A + action 8 + old log-prob 4 + advantage 4 + old value 4 + value 4 read,
d_logits 4A + d_value 4 written); the advantage-statistics and finalize kernels are single workgroups sized for the
2048-sample minibatch and are listed separately.  Times are the library's per-launch HIP events (etm_profile_*).
Bit-exactness of GAE at the scaled size is checked here against a numpy loop over time (vectorised over workers), which is the
reference recurrence of buffer.py:95-113 in its original operation order.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import torch  # noqa: E402

from etm import lib as etm_lib  # noqa: E402
from etm import ops  # noqa: E402

PEAK = 8000.0  # GB/s
dev = torch.device("cuda", 0)


def timed(fn, n):
    """{kernel name: us per launch} from the library's per-launch HIP events (each kernel on its own, no launch gaps)."""
    lib = etm_lib.load()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    etm_lib.profile_collect()
    lib.etm_profile_enable(1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    lib.etm_profile_enable(0)
    return {k: ms / cnt * 1e3 for (_tag, k), (ms, cnt) in etm_lib.profile_collect().items()}


def gae_numpy(r, d, v, last, gamma, lamda):
    g, gl = np.float32(gamma), np.float32(gamma * lamda)
    W, S = r.shape
    adv = np.empty_like(r)
    lv, la = last.copy(), np.zeros(W, dtype=np.float32)
    for t in range(S - 1, -1, -1):
        m = (~d[:, t]).astype(np.float32)
        lv = lv * m
        la = la * m
        delta = r[:, t] + g * lv - v[:, t]
        la = delta + gl * la
        adv[:, t] = la
        lv = v[:, t]
    return adv


def report(name, size, us, nbytes):
    gbs = nbytes / us * 1e-3
    print(f"{name:10s} {size:28s} {us:9.1f} us  {nbytes / 1e6:9.2f} MB  {gbs:8.1f} GB/s  frac {gbs / PEAK:5.3f}", flush=True)


def run_gae(W, S, check):
    gen = torch.Generator(device="cpu").manual_seed(0)
    r = (torch.rand(W, S, generator=gen) < 0.05).float()
    d = torch.rand(W, S, generator=gen) < 0.02
    v = torch.randn(W, S, generator=gen)
    last = torch.randn(W, generator=gen)
    rd, dd, vd, ld = r.to(dev), d.to(dev), v.to(dev), last.to(dev)
    out = torch.empty_like(vd)
    us = timed(lambda: ops.gae(rd, dd, vd, ld, 0.995, 0.95, out=out), 20)
    report("gae", f"W={W} S={S}", us["gae_kernel"], 13 * W * S)
    if check:
        ref = gae_numpy(r.numpy(), d.numpy(), v.numpy(), last.numpy(), 0.995, 0.95)
        same = np.array_equal(out.cpu().numpy(), ref)
        print(f"           bit-exact against the numpy recurrence: {same}", flush=True)
        if not same:
            raise SystemExit(1)


def run_loss(N, A):
    gen = torch.Generator(device="cpu").manual_seed(1)
    logits = torch.randn(N, A, generator=gen).to(dev).requires_grad_(True)
    value = torch.randn(N, generator=gen).to(dev).requires_grad_(True)
    actions = torch.randint(0, A, (N, 1), generator=gen).to(dev)
    old_logp = (-torch.rand(N, 1, generator=gen) - 0.5).to(dev)
    adv = torch.randn(N, generator=gen).to(dev)
    old_value = torch.randn(N, generator=gen).to(dev)
    stats3 = ops.adv_stats(adv)

    def fwd():
        with torch.no_grad():      # the kernel computes the loss and both gradients in the same pass either way
            ops.ppo_loss([logits], value, actions, old_logp, adv, old_value, 0.1, 0.5, 0.001, stats3)
    us = timed(lambda: (ops.adv_stats(adv), fwd()), 20)
    report("ppo_loss", f"N={N} A={A}", us["ppo_loss_kernel"], (28 + 8 * A) * N)
    # single-workgroup kernels sized for a 2048-sample minibatch (launch-bound there); listed for completeness
    report("adv_stats", f"N={N} (one workgroup)", us["adv_stats_kernel"], 4 * N)
    print(f"           ppo_finalize_kernel {us['ppo_finalize_kernel']:.1f} us ({(N + 255) // 256} partial rows, one wave)", flush=True)


if __name__ == "__main__":
    run_gae(32, 512, True)              # config 3
    run_gae(65536, 512, True)           # scaled: 436 MB
    run_loss(2048, 3)                   # config 3 minibatch
    run_loss(1 << 24, 3)                # scaled: 872 MB
