"""Forward+backward time of ops.mha at the training shape for both kernel families (folded / dense), with the library's
per-kernel HIP-event breakdown.  python tools/window_time.py [L] [D] [H]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import torch
from etm import lib as etm_lib
if os.environ.get("ETM_DIAG_LIB"):
    etm_lib.LIB_PATH = os.environ["ETM_DIAG_LIB"]   # diagnostic variants (tools/diag_variants.sh)
from etm import ops
dev = torch.device("cuda"); torch.manual_seed(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 64
D = int(sys.argv[2]) if len(sys.argv) > 2 else 384
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4
N, T, nb, E = 2048, L + 32, 3, 416
bank = torch.randn((E, T, nb, D), device=dev)
mode = os.environ.get("ETM_WIN_SAMPLES", "random")
if mode == "random":          # unrelated windows: no two samples share a row (the un-deduplicated roofline case)
    ep = torch.randint(0, E, (N,), device=dev)
    win = torch.randint(0, T - L + 1, (N, 1), device=dev) + torch.arange(L, device=dev)[None, :]
    mask = torch.arange(L, device=dev)[None, :] < torch.randint(0, L, (N,), device=dev)[:, None]
else:                         # training-like: N of the E * T (episode, step) pairs, sliding windows of trainer.py:88-90;
    pairs = torch.randperm(E * T, device=dev)[:N]      # "sorted": minibatch ordered by bank address, "shuffled": as drawn
    if mode == "sorted":
        pairs = pairs.sort().values
    ep, step = pairs // T, pairs % T
    win = torch.clamp(step - (L - 1), min=0)[:, None] + torch.arange(L, device=dev)[None, :]
    mask = torch.arange(L, device=dev)[None, :] < torch.clamp(step, max=L - 1)[:, None]
print(f"samples: {mode}  (ETM_WIN_SAMPLES=random|shuffled|sorted; ETM_WIN_XCD_MAP=1 with a v2 library: contiguous sample chunk per XCD)")
wk = (torch.randn((D, D), device=dev) / D ** 0.5).requires_grad_(True); wv = (torch.randn((D, D), device=dev) / D ** 0.5).requires_grad_(True)
q = torch.randn((N, D), device=dev).requires_grad_(True); g = torch.randn((N, D), device=dev)
spec = ops.WindowSpec.from_bank(bank, ep, win, None, mask)
l = etm_lib.load()
res = {}
for impl in ("folded", "dense"):
    ops.set_attention_impl(impl)
    for it in range(3):
        out, att = ops.mha(q, wk, wv, spec, 1, H); (out * g).sum().backward()
    res[impl] = (out.detach().clone(), q.grad.clone(), wk.grad.clone(), wv.grad.clone())
    q.grad = wk.grad = wv.grad = None
    torch.cuda.synchronize()
    t0, t1, t2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    fw = bw = 0.0
    for it in range(10):
        t0.record(); out, att = ops.mha(q, wk, wv, spec, 1, H); t1.record(); (out * g).sum().backward(); t2.record()
        torch.cuda.synchronize(); fw += t0.elapsed_time(t1); bw += t1.elapsed_time(t2)
    l.etm_profile_enable(1)
    for it in range(5):
        out, att = ops.mha(q, wk, wv, spec, 1, H); (out * g).sum().backward()
    torch.cuda.synchronize(); l.etm_profile_enable(0)
    ks = "  ".join(f"{k}={ms / c * 1e3:.1f}us" for (tag, k), (ms, c) in sorted(etm_lib.profile_collect().items()))
    print(f"{impl:7s} N={N} L={L} D={D} H={H}: forward {fw / 10 * 1e3:7.1f} us  backward(+loss) {bw / 10 * 1e3:7.1f} us | {ks}", flush=True)
import zlib
print("  folded result checksums (crc32 of the fp32 bytes; equal across library variants = bit-identical): " +
      "  ".join(f"{name}={zlib.crc32(x.cpu().numpy().tobytes()):08x}" for name, x in zip(("ctx", "dq", "dwk", "dwv"), res["folded"])))
a, b = res["folded"], res["dense"]
for name, x, y in zip(("ctx", "dq", "dwk", "dwv"), a, b):
    print(f"  folded vs dense {name}: max abs diff {(x - y).abs().max().item():.3e}  rel-to-norm {((x - y).norm() / y.norm()).item():.3e}")
gb = N * L * D * 4 / 1e9
print(f"  window bytes per pass (un-deduplicated) {gb * 1e3:.1f} MB")

if os.environ.get("ETM_DIAG_LIB", "").endswith("trace.so"):
    import ctypes, numpy as np
    ops.set_attention_impl("folded")
    out, att = ops.mha(q, wk, wv, spec, 1, H); torch.cuda.synchronize()      # last launch = one forward pass
    buf = np.zeros(2048 * 8 * 16, dtype=np.uint64)
    f = l.etm_diag_win_trace_read; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    assert f(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(2048, 8, 16).astype(np.int64)
    names = ["", "start", "bookkeeping loaded", "rows loaded", "pass1+reduce", "barrier", "softmax", "barrier", "pass2", "barrier", "sum+store", "end"]
    print("  per-wave phase durations (s_memtime ticks, mean / p90):")
    for i in range(2, 12):
        d = t[:, :, i] - t[:, :, i - 1]
        print(f"    {names[i]:20s} {d.mean():8.0f} {np.percentile(d, 90):8.0f}")
    life = t[:, :, 11] - t[:, :, 1]
    print(f"    lifetime {life.mean():.0f}; kernel span {t[:, :, 11].max() - t[:, :, 1].min()} ticks")
