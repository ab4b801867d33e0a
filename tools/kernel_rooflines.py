"""Kernel micro-benchmarks behind the roofline fractions of SURVEY.md section 8d (imported by bench.py, runnable on its own
for rocprofv3 passes):

    python tools/kernel_rooflines.py [window_cold|window_train|window_sorted|mfma3|mfma5|gae|ppo|all] [launches]

Every figure is algorithmic work per launch (stated below) / average launch duration from the library's per-launch HIP
events (etm_profile_*: an event pair on the launch stream around each kernel), against the MI355X peaks of
/opt/skills/guides/MI355X_MICROARCH.md (HBM 8 TB/s; fp32 MFMA 157.3 TFLOP/s).

  window pass   HBM, N * L * D * 4 bytes per launch (the un-deduplicated window read of SURVEY 8d; the folded vectors, the
                attention weights and the outputs are reported separately as extra_bytes)
      cold      every sample reads rows no other sample reads, from a bank of 906 MB, and consecutive launches walk the three
                blocks of the bank (604 MB between two reads of a row > the 256 MB Infinity Cache): every byte comes from HBM
      train     the access pattern of the optimisation phase: N of the W * S (worker, step) pairs of a 32 x 512 rollout with
                sliding windows -- unique rows per block <= 61 MB, so L2 / Infinity-Cache hits are part of the rate
      sorted    the same minibatch in ascending (worker, step) order, as the trainer hands it over (sort_minibatch): every XCD
                takes a contiguous chunk of the samples, neighbours in time share most of their rows in that XCD's L2
  dense MFMA    fp32 MFMA, forward N * 2 * (2 L D^2 + 2 L D) flop, dW N * 2 * (2 L D^2) flop per launch, config 3 and config 5 dims
  GAE           HBM, 13 bytes per (worker, step): config size (32 x 512) and 65,536 x 512
  PPO loss      HBM, 28 + 8 A bytes per sample: minibatch size (2048) and 2^24 samples
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "episodic-transformer-memory-ppo_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)
import torch  # noqa: E402

from etm import lib as etm_lib  # noqa: E402
from etm import ops  # noqa: E402

HBM_PEAK_GBS = 8000.0
FP32_MFMA_PEAK_TFLOPS = 157.3


def _timed(fn, launches, warm=3):
    """{kernel: (avg_ms, launches)} from the library's per-launch HIP events."""
    lib = etm_lib.load()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    etm_lib.profile_collect()
    lib.etm_profile_enable(1)
    for _ in range(launches):
        fn()
    torch.cuda.synchronize()
    lib.etm_profile_enable(0)
    out = {}
    for (_tag, k), (ms, cnt) in etm_lib.profile_collect().items():
        out[k] = (ms / cnt, cnt)
    return out


def _hbm(name, nbytes, avg_ms, launches, **extra):
    gbs = nbytes / (avg_ms * 1e-3) / 1e9
    return dict(kernel=name, bound="hbm", bytes_per_launch=nbytes, avg_launch_ms=avg_ms, launches=launches, achieved=gbs, peak=HBM_PEAK_GBS,
                unit="GB/s", frac=gbs / HBM_PEAK_GBS, **extra)


def _mfma(name, flops, avg_ms, launches, **extra):
    tf = flops / (avg_ms * 1e-3) / 1e12
    return dict(kernel=name, bound="mfma", flops_per_launch=flops, avg_launch_ms=avg_ms, launches=launches, achieved=tf,
                peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / FP32_MFMA_PEAK_TFLOPS, dtype="f32 (v_mfma_f32_32x32x2_f32)", **extra)


def window(mode, dev, launches=24, N=2048, L=64, D=384, H=4):
    """Folded window pass, forward and backward, ``mode`` = "cold", "train" or "sorted" (see the module docstring)."""
    gen = torch.Generator(device="cpu").manual_seed(0)
    nb = 3
    if mode == "cold":
        T = L + 32
        bank = torch.randn((N, T, nb, D), device=dev)                      # one episode per sample: no shared rows
        ep = torch.randperm(N, generator=gen).to(dev)
        start = torch.randint(0, T - L + 1, (N, 1), generator=gen).to(dev)
        win = start + torch.arange(L, device=dev)[None, :]
        cnt = torch.randint(0, L, (N,), generator=gen).to(dev)
    else:
        W, S, T = 32, 512, 96                                             # a rollout of config 3: ~13 episodes per worker
        E = W * (S // 40 + 1)
        bank = torch.randn((E, T, nb, D), device=dev)
        pairs = torch.randperm(W * S, generator=gen)[:N].to(dev)           # a minibatch: N of the (worker, step) pairs
        if mode == "sorted":
            pairs = torch.sort(pairs).values
        w, s = pairs // S, pairs % S
        ep = w * (S // 40 + 1) + s // 40                                   # episodes of 40 steps
        step = s % 40
        win = torch.clamp(step - (L - 1), min=0)[:, None] + torch.arange(L, device=dev)[None, :]
        cnt = torch.clamp(step, max=L - 1)
    mask = torch.arange(L, device=dev)[None, :] < cnt[:, None]
    spec = ops.WindowSpec.from_bank(bank, ep, win, None, mask)
    spec.pos_included = True
    u = torch.randn((H, N, D), device=dev).requires_grad_(True)
    gz = torch.randn((H, N, D), device=dev)
    state = {"b": 0}

    def step_fn():
        b = state["b"] = (state["b"] + 1) % nb                            # walk the blocks: 3 x 201 MB between two reads of a row
        z, _att = ops._WindowFn.apply(u, None, None, None, spec, b, 1e-5)
        z.backward(gz)
        u.grad = None

    t = _timed(step_fn, launches)
    alg = N * L * D * 4
    unique = int(torch.unique(ep * (bank.shape[1]) + win[:, 0]).numel())   # lower bound of distinct windows (by first row)
    res = {}
    for k, extra in (("window_fwd_kernel", 4 * N * (2 * H * D + H * L)), ("window_bwd_kernel", 4 * N * (2 * H * D + 2 * H * L))):
        if k in t:
            res[k] = _hbm(k, alg, t[k][0], t[k][1], extra_bytes_per_launch=extra, access=mode, shape=dict(N=N, L=L, D=D, H=H),
                          bank_bytes=bank.numel() * 4, distinct_windows=unique)
    return res


def mfma(dev, launches=20, N=2048, L=64, D=384, H=4):
    """Dense (north-star) formulation: K/V projections of the window as fp32-MFMA contractions; forward and dW kernels."""
    gen = torch.Generator(device="cpu").manual_seed(1)
    T, nb, E = L + 32, 3, 416
    bank = torch.randn((E, T, nb, D), device=dev)
    ep = torch.randint(0, E, (N,), generator=gen).to(dev)
    win = torch.randint(0, T - L + 1, (N, 1), generator=gen).to(dev) + torch.arange(L, device=dev)[None, :]
    mask = torch.arange(L, device=dev)[None, :] < torch.randint(0, L, (N,), generator=gen).to(dev)[:, None]
    spec = ops.WindowSpec.from_bank(bank, ep, win, None, mask)
    wk = (torch.randn((D, D), device=dev) / D ** 0.5).requires_grad_(True)
    wv = (torch.randn((D, D), device=dev) / D ** 0.5).requires_grad_(True)
    q = torch.randn((N, D), device=dev).requires_grad_(True)
    g = torch.randn((N, D), device=dev)

    def step_fn():
        out, _ = ops.mha(q, wk, wv, spec, 1, H, impl="dense")
        out.backward(g)
        q.grad = wk.grad = wv.grad = None

    t = _timed(step_fn, launches)
    shape = dict(N=N, L=L, D=D, H=H)
    res = {}
    if "mha_fwd_kernel" in t:
        res["mha_fwd_kernel"] = _mfma("mha_fwd_kernel", N * 2.0 * (2 * L * D * D + 2 * L * D), *t["mha_fwd_kernel"], shape=shape)
    if "bwd_dw_kernel" in t:
        res["bwd_dw_kernel"] = _mfma("bwd_dw_kernel", N * 2.0 * (2 * L * D * D), *t["bwd_dw_kernel"], shape=shape)
    return res


def gae(dev, W, S, launches=20):
    gen = torch.Generator(device="cpu").manual_seed(2)
    r = (torch.rand(W, S, generator=gen) < 0.05).float().to(dev)
    d = (torch.rand(W, S, generator=gen) < 0.02).to(dev)
    v = torch.randn(W, S, generator=gen).to(dev)
    last = torch.randn(W, generator=gen).to(dev)
    out = torch.empty_like(v)
    t = _timed(lambda: ops.gae(r, d, v, last, 0.995, 0.95, out=out), launches)
    return _hbm("gae_kernel", 13 * W * S, *t["gae_kernel"], shape=dict(W=W, S=S), bytes_per_element=13)


def ppo(dev, N, A=3, launches=20):
    gen = torch.Generator(device="cpu").manual_seed(3)
    logits = torch.randn(N, A, generator=gen).to(dev)
    value = torch.randn(N, generator=gen).to(dev)
    actions = torch.randint(0, A, (N, 1), generator=gen).to(dev)
    old_logp = (-torch.rand(N, 1, generator=gen) - 0.5).to(dev)
    adv = torch.randn(N, generator=gen).to(dev)
    old_value = torch.randn(N, generator=gen).to(dev)
    stats3 = ops.adv_stats(adv)

    def step_fn():
        with torch.no_grad():       # the kernel computes the loss and both gradients in one pass either way
            ops.adv_stats(adv)
            ops.ppo_loss([logits], value, actions, old_logp, adv, old_value, 0.1, 0.5, 0.001, stats3)

    t = _timed(step_fn, launches)
    res = _hbm("ppo_loss_kernel", (28 + 8 * A) * N, *t["ppo_loss_kernel"], shape=dict(N=N, A=A), bytes_per_element=28 + 8 * A)
    res["adv_stats_kernel_ms"] = t["adv_stats_kernel"][0]
    res["adv_stats_gbs"] = 4 * N / (t["adv_stats_kernel"][0] * 1e-3) / 1e9
    res["ppo_finalize_kernel_ms"] = t["ppo_finalize_kernel"][0]
    return res


def all_rooflines(dev, quick=False):
    """Everything bench.py reports next to the throughput line (outside its timed region)."""
    n = 10 if quick else 24
    out = {"window": {"cold_hbm": window("cold", dev, n), "train_like": window("train", dev, n), "train_sorted": window("sorted", dev, n),
                      "cold_hbm_L128": window("cold", dev, n, L=128)}}
    torch.cuda.empty_cache()
    out["mfma"] = {"config3_dims": mfma(dev, n, 2048, 64, 384, 4), "config5_dims": mfma(dev, n, 2048, 128, 384, 4)}
    torch.cuda.empty_cache()
    out["gae"] = {"config_size": gae(dev, 32, 512, n), "scaled": gae(dev, 65536, 512, n)}
    out["ppo_loss"] = {"config_size": ppo(dev, 2048, 3, n), "scaled": ppo(dev, 1 << 24, 3, n)}
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    import json
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    dev = torch.device("cuda", 0)
    if what == "all":
        res = all_rooflines(dev)
    elif what == "window_cold":
        res = window("cold", dev, n)
    elif what == "window_cold128":
        res = window("cold", dev, n, L=128)
    elif what == "window_train":
        res = window("train", dev, n)
    elif what == "window_sorted":
        res = window("sorted", dev, n)
    elif what == "mfma3":
        res = mfma(dev, n, 2048, 64, 384, 4)
    elif what == "mfma5":
        res = mfma(dev, n, 2048, 128, 384, 4)
    elif what == "gae":
        res = {"config_size": gae(dev, 32, 512, n), "scaled": gae(dev, 65536, 512, n)}
    elif what == "ppo":
        res = {"config_size": ppo(dev, 2048, 3, n), "scaled": ppo(dev, 1 << 24, 3, n)}
    else:
        raise SystemExit(__doc__)
    print(json.dumps(res, indent=1))
