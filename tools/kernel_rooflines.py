"""Kernel micro-benchmarks behind the roofline fractions of SURVEY.md section 8d (imported by bench.py, runnable on its own
for rocprofv3 passes):

    python tools/kernel_rooflines.py [window_cold|window_train|window_sorted|mfma3|mfma5|gae|ppo|encoder|grouped_dw|rollout_step|all] [launches]

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
  encoder       fp32 MFMA, 2 * N * Ho * Wo * k * k * C * Cout flop per layer and pass (forward / backward-data / backward-weight incl.
                its slice reduction) of the three convolutions at N = 2048, 3 x 84 x 84
  rollout_step  the kernels of one rollout step of one worker group in their captured order; the step kernel's streamed bytes
                (W x every matrix of the chain + the workers' K | V window columns) / time, and the length of its dependency chain
"""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # (as bench.py: the rollout_step target builds a trainer -- rollout_groups: auto)
os.environ.setdefault("ETM_HW_QUEUES_SET_EARLY", "1")   # marker for trainer.py: the line above ran before the HIP runtime started

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
BF16_MFMA_PEAK_TFLOPS = 2500.0      # dense (MI355X_MICROARCH.md): v_mfma_f32_32x32x16_bf16, 16 x the fp32 MFMA rate
B3_DTYPE = "f32 results, every product as 6 x v_mfma_f32_32x32x16_bf16 of exactly split operands (3 bf16 terms each)"
B3_NOTE = ("achieved / frac: the algorithmic fp32 flops against the fp32 MFMA peak (what a kernel on v_mfma_f32_32x32x2_f32 is bounded by -- "
           "frac > 1 is past that bound); pipe_*: the bf16 products really issued (6 per fp32 product) against the dense bf16 MFMA peak")


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


def b3_fields(flops, avg_ms):
    """The bf16-pipe view of a kernel that takes its fp32 products as six bf16 MFMA products (csrc/conv_b3*.hip)."""
    tf = 6.0 * flops / (avg_ms * 1e-3) / 1e12
    return dict(dtype=B3_DTYPE, pipe_flops_per_launch=6.0 * flops, pipe_achieved=tf, pipe_peak=BF16_MFMA_PEAK_TFLOPS, pipe_frac=tf / BF16_MFMA_PEAK_TFLOPS, note=B3_NOTE)


def _mfma(name, flops, avg_ms, launches, b3=False, **extra):
    tf = flops / (avg_ms * 1e-3) / 1e12
    d = dict(kernel=name, bound="mfma", flops_per_launch=flops, avg_launch_ms=avg_ms, launches=launches, achieved=tf,
             peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / FP32_MFMA_PEAK_TFLOPS, dtype="f32 (v_mfma_f32_32x32x2_f32)", **extra)
    if b3:
        d.update(b3_fields(flops, avg_ms))
    return d


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
    # the loss cannot run without the statistics pass in front of it and the finalize behind it: the three launches together
    # against the bytes of all three (the advantages are read once more by the statistics pass)
    total_ms = t["ppo_loss_kernel"][0] + t["adv_stats_kernel"][0] + t["ppo_finalize_kernel"][0]
    nbytes = (28 + 8 * A + 4) * N
    res["with_stats"] = dict(kernel="adv_stats + ppo_loss + ppo_finalize", bound="hbm", bytes_per_launch=nbytes, avg_launch_ms=total_ms,
                             achieved=nbytes / (total_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=nbytes / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
    return res


ENCODER_LAYERS = ((3, 84, 84, 32, 8, 4), (32, 20, 20, 64, 4, 2), (64, 9, 9, 64, 3, 1))      # (C, H, W, Cout, k, stride): model.py:29-31 on 3x84x84


def encoder_flops(N, layers=ENCODER_LAYERS):
    """Algorithmic flops of one pass of each layer: 2 * (N * Ho * Wo) * (k * k * C) * Cout (the same for forward, backward-data
    and backward-weight)."""
    return [2.0 * N * ((h - k) // s + 1) * ((w - k) // s + 1) * cout * k * k * c for (c, h, w, cout, k, s) in layers]


def encoder(dev, launches=20, N=2048, products=None):
    """Training-side encoder kernels layer by layer and pass by pass through the C ABI at the minibatch shape.  ``products``:
    "bf16x3" (csrc/conv_b3.hip, conv_b3_wgrad.hip: the default of the trainer since round 6) or "fp32" (csrc/conv_train.hip and the
    LDS-resident forms on v_mfma_f32_32x32x2_f32); None: what ops is set to.  Weight-gradient figures include the slice reduction."""
    import ctypes
    lib = etm_lib.load()
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: None if t is None else t.data_ptr()
    products = products or ops._encoder_products
    b3 = products == "bf16x3"
    torch.manual_seed(0)
    bufs = []
    for (c, h, w, cout, k, s) in ENCODER_LAYERS:
        ho, wo = (h - k) // s + 1, (w - k) // s + 1
        wt = torch.randn((cout, c, k, k), device=dev) * 0.05
        K = k * k * c
        if b3:
            slices = lib.etm_conv_b3_wgrad_slices(N, c, h, w, cout, k, k, s)
            nbytes = slices * (K * cout + cout) * 4
            packs = ops.conv_b3_pack([wt, wt] if c != 3 else [wt], [0, 1] if c != 3 else [0], [s, s] if c != 3 else [s])
        else:
            slices, nbytes = 0, lib.etm_conv_train_wgrad_workspace_bytes(N, c, h, w, cout, k, k, s)
            packs = [ops.conv_pack_weights(wt.permute(0, 2, 3, 1).reshape(cout, -1)), ops.conv_pack_dgrad_weights(wt, s) if c != 3 else None]
        bufs.append(dict(x=torch.rand((N, h, w, c), device=dev), b=torch.randn(cout, device=dev), y=torch.empty((N, ho, wo, cout), device=dev),
                         dy=torch.randn((N, ho, wo, cout), device=dev), packed=packs[0], pd=packs[1] if len(packs) > 1 else None,
                         dx=torch.empty((N, h, w, c), device=dev), dw=torch.empty(K * cout + cout, device=dev), ws=torch.empty(max(nbytes, 8) // 4, device=dev),
                         nbytes=nbytes, slices=slices, ybits=torch.empty((N, ho, wo, cout // 32), dtype=torch.int32, device=dev),
                         xbits=torch.randint(-2 ** 31, 2 ** 31 - 1, (N, h, w, max(c // 32, 1)), dtype=torch.int32, device=dev)))
    one = lambda ct, v: (ct * 1)(v)

    def step_fn():
        for (c, h, w, cout, k, s), b in zip(ENCODER_LAYERS, bufs):
            if b3:
                etm_lib.check(lib.etm_conv_b3_fwd(P(b["x"]), None, P(b["packed"]), P(b["b"]), P(b["y"]), P(b["ybits"]) if c != 64 else None, N, c, h, w, cout, k, k, s, st), "fwd")
                if b["pd"] is not None:
                    etm_lib.check(lib.etm_conv_b3_dgrad(P(b["dy"]), None, P(b["pd"]), None, P(b["xbits"]), P(b["dx"]), N, c, h, w, cout, k, k, s, st), "dgrad")
                etm_lib.check(lib.etm_conv_b3_wgrad(P(b["x"]), None, P(b["dy"]), None, P(b["ws"]), b["nbytes"], N, c, h, w, cout, k, k, s, st), "wgrad")
                K = k * k * c
                etm_lib.check(lib.etm_conv_wgrad_reduce_grouped(one(ctypes.c_void_p, P(b["ws"])), one(ctypes.c_int32, b["slices"]), one(ctypes.c_void_p, P(b["dw"])),
                                                                one(ctypes.c_void_p, b["dw"].data_ptr() + K * cout * 4), one(ctypes.c_int32, cout), one(ctypes.c_int32, c),
                                                                one(ctypes.c_int32, k), one(ctypes.c_int32, k), 1, st), "reduce")
            else:
                etm_lib.check(lib.etm_conv_train_fwd(P(b["x"]), None, N, P(b["packed"]), P(b["b"]), P(b["y"]), N, c, h, w, cout, k, k, s, 0, st), "fwd")
                if b["pd"] is not None:
                    etm_lib.check(lib.etm_conv_train_dgrad(P(b["dy"]), P(b["pd"]), P(b["x"]), P(b["dx"]), N, c, h, w, cout, k, k, s, st), "dgrad")
                etm_lib.check(lib.etm_conv_train_wgrad(P(b["x"]), None, P(b["dy"]), P(b["dw"]), P(b["ws"]), b["nbytes"], N, c, h, w, cout, k, k, s, st), "wgrad")

    t = _timed(step_fn, launches)
    fl = encoder_flops(N)
    # b3: the three slice reductions are launches of the generic weight-gradient id (one per layer and call): their mean rides on every layer's figure
    red = t.get("conv_train_wgrad_kernel", (0.0, 0))[0] if b3 else 0.0
    res, tot_ms, tot_fl = {}, 0.0, 0.0
    for li in range(3):
        for pas in ("fwd", "dgrad", "wgrad"):
            name = f"conv_{pas}_layer{li + 1}"
            if name not in t:
                continue
            avg, cnt = t[name]
            if b3:
                per_pass, n = avg + (red if pas == "wgrad" else 0.0), cnt
            else:
                per_pass, n = avg * (2 if pas == "wgrad" else 1), cnt // (2 if pas == "wgrad" else 1)      # wgrad: main kernel + slice reduction are two timed launches
            res[name] = _mfma(name, fl[li], per_pass, n, b3=b3, shape=dict(N=N, layer=ENCODER_LAYERS[li]))
            tot_ms += per_pass
            tot_fl += fl[li]
    res["all_passes"] = dict(flops=tot_fl, ms=tot_ms, achieved=tot_fl / (tot_ms * 1e-3) / 1e12, peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                             frac=tot_fl / (tot_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, passes=8, products=products)
    if b3:
        res["all_passes"].update({k: v for k, v in b3_fields(tot_fl, tot_ms).items() if k != "pipe_flops_per_launch"})
    return res


def grouped_dw(dev, launches=20, N=2048, D=384, H=4, nb=3, hid=384):
    """All dense-layer weight gradients of one minibatch step at config-3 dims as ONE launch (csrc/grouped_dw.hip): per block
    queries, fc_out, fc ([D, D] each) and the per-head key / value folds (H problems of [hd, D] each), plus linear_embedding,
    lin_policy, lin_value -- 2 * N * out * in flop per layer, against the fp32 MFMA peak."""
    import ctypes
    lib = etm_lib.load()
    torch.manual_seed(0)
    hd = D // H
    probs, keep = [], []
    for _ in range(nb * 3 + 1):                                         # plain [D, D] layers
        a, b, c = torch.randn((N, D), device=dev), torch.randn((N, D), device=dev), torch.empty((D, D), device=dev)
        keep += [a, b, c]
        probs.append((a.data_ptr(), b.data_ptr(), c.data_ptr(), D, D, D, D, D))
    for _ in range(2):                                                  # lin_policy, lin_value
        a, b, c = torch.randn((N, hid), device=dev), torch.randn((N, D), device=dev), torch.empty((hid, D), device=dev)
        keep += [a, b, c]
        probs.append((a.data_ptr(), b.data_ptr(), c.data_ptr(), hid, D, hid, D, D))
    for _ in range(nb * 2):                                             # key / value folds: H problems each
        a, b, c = torch.randn((N, D), device=dev), torch.randn((H, N, D), device=dev), torch.empty((D, D), device=dev)
        keep += [a, b, c]
        for h in range(H):
            probs.append((a.data_ptr() + 4 * h * hd, b[h].data_ptr(), c[h * hd:].data_ptr(), hd, D, D, D, D))
    k = len(probs)
    pa = (ctypes.c_void_p * k)(*[p[0] for p in probs])
    pb = (ctypes.c_void_p * k)(*[p[1] for p in probs])
    pc = (ctypes.c_void_p * k)(*[p[2] for p in probs])
    dims = (ctypes.c_int32 * (5 * k))(*[v for p in probs for v in p[3:8]])
    st = torch.cuda.current_stream().cuda_stream
    t = _timed(lambda: etm_lib.check(lib.etm_grouped_dw(pa, pb, pc, dims, k, N, st), "etm_grouped_dw"), launches)
    flops = sum(2.0 * N * p[3] * p[4] for p in probs)
    tiles = sum((p[3] // 96) * (p[4] // 128) for p in probs)
    return _mfma("grouped_dw_kernel", flops, *t["grouped_dw_kernel"], shape=dict(N=N, D=D, H=H, blocks=nb, problems=k, workgroups=tiles),
                 replaces="the dW GEMMs of these layers, one library launch each (15 us x 19 at config 3, 0.26 of the peak)")


def rollout_step_model(cfg, W, hidden_features, group=False):
    """What one launch of the rollout step kernel (csrc/rollout_fused.hip) moves and how long its dependency chain is.
    Every worker's team streams every matrix of the chain once (weights are shared by all workers, but a team serves ONE worker:
    matrix-VECTOR products), plus the worker's K | V cache columns of the window; the tail streams the K | V projection.
    ``group``: the group form (csrc/rollout_group.hip, gated layouts): the chain's matrices once per GROUP; the tail (a unit projects its
    own worker's items onto its head's K | V columns) still once per worker; every exchange is an all-gather of [8, D] activations."""
    t = cfg["transformer"]
    if group:
        D, nb, L, H = t["embed_dim"], t["num_blocks"], t["memory_length"], t["num_heads"]
        hid = cfg["hidden_layer_size"]
        w_chain = D * D + nb * 14 * D * D + D * 2 * hid        # (fc_out is folded into the first gate's maps of y: 14 maps per block)
        w_tail = nb * D * 2 * D
        kv = nb * L * 2 * D
        exchanges = 7 * nb + 3
        piece = 8 * (D // 32)                                   # floats a workgroup publishes per exchange; 16-byte packets carry 2 each
        ex_written, ex_read = exchanges * 32 * piece * 8, exchanges * 32 * 32 * piece * 8      # bytes (one polling pass over every piece)
        return dict(weight_bytes_chain_once_per_group=4 * w_chain, weight_bytes_per_worker_tail=4 * w_tail, kv_bytes_per_worker=4 * kv,
                    bytes_per_launch=4 * (w_chain + W * (w_tail + kv)), bytes_per_launch_to_handover=4 * (w_chain + W * kv),
                    exchange_bytes_written=ex_written, exchange_bytes_read_one_pass=ex_read,
                    unique_weight_bytes=4 * (w_chain + w_tail), unique_bytes_per_launch=4 * (w_chain + w_tail + W * kv),
                    dependent_products=2 + 8 * nb, team_exchanges=exchanges, dependent_phases=2 + 8 * nb + exchanges + nb * 2 + 1, workers=W,
                    workgroups=32, form="group (csrc/rollout_group.hip)")
    D, nb, L, H = t["embed_dim"], t["num_blocks"], t["memory_length"], t["num_heads"]
    hid = cfg["hidden_layer_size"]
    gates = 2 * 6 * D * D if t.get("gtrxl") else 0
    w_chain = D * D + nb * (3 * D * D + gates) + D * 2 * hid              # embedding, per block q / fc_out / fc (+ gates), hidden heads
    w_tail = nb * D * 2 * D
    kv = nb * L * 2 * D
    exchanges = 2 + nb * (6 if t.get("gtrxl") else 2)
    products = 2 + nb * (15 if t.get("gtrxl") else 3)
    return dict(weight_bytes_per_worker_to_handover=4 * w_chain, weight_bytes_per_worker_tail=4 * w_tail, kv_bytes_per_worker=4 * kv,
                bytes_per_launch=4 * W * (w_chain + w_tail + kv), bytes_per_launch_to_handover=4 * W * (w_chain + kv),
                unique_weight_bytes=4 * (w_chain + w_tail), unique_bytes_per_launch=4 * (w_chain + w_tail + W * kv),
                dependent_products=products, team_exchanges=exchanges,
                dependent_phases=products + exchanges + nb * 2 + 1, workers=W)


def rollout_step(trainer, launches=40):
    """The kernels of ONE rollout step of one worker group (conv x3, lin_hidden K-slice sums, the step kernel), launched eagerly
    in their captured order with the library's per-launch HIP events.  Call after a rollout has run (graphs captured, weights
    refreshed); the step counter is held at 0, the bank / cache rows it writes are not used afterwards by the caller."""
    lib = etm_lib.load()
    g = trainer._groups[0]
    so, hf = trainer._stream_obs, trainer._host_flag

    def step_fn():
        g.t_dev.zero_()
        g.flag_np[0] = 0
        with torch.no_grad():
            trainer._rollout_step_device(g, so, hf)

    stream = g.stream if g.stream is not None else torch.cuda.current_stream()
    with torch.cuda.stream(stream):
        t = _timed(step_fn, launches)
    torch.cuda.synchronize()
    res = {"kernels": {k: dict(avg_us=v[0] * 1e3, launches_per_step=v[1] / launches) for k, v in t.items()}}
    res["step_sum_us"] = sum(v[0] * 1e3 * v[1] / launches for v in t.values())
    if "rollout_trxl_kernel" in t:
        feats = trainer.model.lin_hidden.in_features
        m = rollout_step_model(trainer.config, g.W, feats, group=bool(getattr(g, "group_kernel", False)))
        us = t["rollout_trxl_kernel"][0] * 1e3
        gbs = m["bytes_per_launch"] / (us * 1e-6) / 1e9
        res["rollout_trxl_kernel"] = dict(kernel="rollout_trxl_kernel", bound="latency (dependent chain of matrix-vector products; bytes are L2 / "
                                          "Infinity-Cache streams, not HBM)", avg_launch_ms=us * 1e-3, launches=t["rollout_trxl_kernel"][1],
                                          bytes_per_launch=m["bytes_per_launch"], achieved=gbs, unit="GB/s", peak=HBM_PEAK_GBS,
                                          frac=gbs / HBM_PEAK_GBS, l2_aggregate_peak_gbs=34500.0, frac_of_l2_peak=gbs / 34500.0,
                                          # every team re-reads the SHARED weight set: the bytes that are distinct addresses (weights once
                                          # + every worker's own K | V window columns) against the same time and peak
                                          unique_bytes_per_launch=m["unique_bytes_per_launch"],
                                          frac_unique=m["unique_bytes_per_launch"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                          us_per_dependent_phase=us / m["dependent_phases"], model=m, placement="team_xcd")
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
    out["encoder"] = encoder(dev, n, products="bf16x3")         # (the trainer's default since round 6)
    torch.cuda.empty_cache()
    out["encoder_fp32_mfma"] = encoder(dev, n, products="fp32")  # (rounds 2 - 3: encoder_products: fp32)
    torch.cuda.empty_cache()
    out["block_weight_gradients"] = grouped_dw(dev, n)
    torch.cuda.empty_cache()
    return out


def _bench_trainer(dev, overrides=()):
    """A PPOTrainer on BASELINE config 3 that has run one rollout (graphs captured): the fixture of the rollout_step target."""
    from yaml_parser import YamlParser
    from trainer import PPOTrainer
    cfg = YamlParser(os.path.join(PKG, "configs", os.environ.get("ETM_PROFILE_CONFIG", "synthetic_minigrid") + ".yaml")).get_config()
    for kv in overrides:
        k, v = kv.split("=")
        cfg[k] = (v == "1") if isinstance(cfg.get(k, None), bool) or k in ("rollout_group_kernel",) else (int(v) if v.lstrip("-").isdigit() else v)
    torch.manual_seed(0)
    tr = PPOTrainer(cfg, run_id="roofline", device=dev, tensorboard=False)
    tr._sample_training_data()
    tr.buffer.prepare_batch_dict()
    torch.cuda.synchronize()
    return tr


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
    elif what in ("encoder", "encoder_fp32"):
        res = encoder(dev, n, products="fp32" if what == "encoder_fp32" else "bf16x3")
    elif what == "grouped_dw":
        res = grouped_dw(dev, n)
    elif what == "rollout_step":
        tr = _bench_trainer(dev, sys.argv[3:])
        res = rollout_step(tr, n)
        tr.close()
    else:
        raise SystemExit(__doc__)
    print(json.dumps(res, indent=1))
