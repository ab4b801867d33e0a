"""Times the forward and dW kernels of one diagnostic library variant (tools/diag_variants.sh) at the training shape and,
for the `trace` variant, dumps / summarises the per-wave s_memtime samples.  Diagnostic tool only.

    python tools/diag_run.py <variant> [out_dir]
"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import numpy as np
import torch
from etm import lib as etm_lib
variant = sys.argv[1]
out_dir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "gpurun_out")
etm_lib.LIB_PATH = os.path.join(REPO, "tools", "diag_build", f"libetm_{variant}.so")
from etm import ops
dev = torch.device("cuda"); torch.manual_seed(0)
N, L, D, H, T, nb, E = 2048, 64, 384, 4, 96, 3, 416
bank = torch.randn((E, T, nb, D), device=dev)
ep = torch.randint(0, E, (N,), device=dev)
win = torch.randint(0, T - L + 1, (N, 1), device=dev) + torch.arange(L, device=dev)[None, :]
mask = torch.arange(L, device=dev)[None, :] < torch.randint(0, L, (N,), device=dev)[:, None]
wk = (torch.randn((D, D), device=dev) / D ** 0.5).requires_grad_(True); wv = (torch.randn((D, D), device=dev) / D ** 0.5).requires_grad_(True)
q = torch.randn((N, D), device=dev).requires_grad_(True); g = torch.randn((N, D), device=dev)
spec = ops.WindowSpec.from_bank(bank, ep, win, None, mask)
l = etm_lib.load()
for it in range(10):
    if it == 3: l.etm_profile_enable(1)
    out, _ = ops.mha(q, wk, wv, spec, 1, H); (out * g).sum().backward()
torch.cuda.synchronize(); l.etm_profile_enable(0)
res = {k: ms / c * 1e3 for (tag, k), (ms, c) in etm_lib.profile_collect().items()}
print(f"VARIANT {variant:16s} fwd_us={res.get('mha_fwd_kernel', 0):8.1f} dw_us={res.get('bwd_dw_kernel', 0):8.1f}", flush=True)

if not variant.startswith("trace"):
    sys.exit(0)


def read(fn, words):
    buf = np.zeros(words, dtype=np.uint64)
    f = getattr(l, fn); f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    rc = f(buf.ctypes.data, buf.nbytes)
    assert rc == 0, rc
    return buf


def ids(hw):
    xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
    h = hw & np.uint64(0xFFFFFFFF)
    return dict(xcc=xcc.astype(int), simd=((h >> np.uint64(4)) & np.uint64(3)).astype(int), cu=((h >> np.uint64(8)) & np.uint64(15)).astype(int),
                sh=((h >> np.uint64(12)) & np.uint64(1)).astype(int), se=((h >> np.uint64(13)) & np.uint64(7)).astype(int),
                wave=(h & np.uint64(15)).astype(int))


def pct(x, name):
    x = np.asarray(x, dtype=np.float64)
    print(f"  {name:34s} mean {x.mean():9.0f}  p10 {np.percentile(x, 10):9.0f}  p50 {np.percentile(x, 50):9.0f}  p90 {np.percentile(x, 90):9.0f}  max {x.max():9.0f}")


def union_len(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs


os.makedirs(out_dir, exist_ok=True)
# ---------------- forward
fw = read("etm_diag_fw_trace_read", 4096 * 4 * 64).reshape(4096, 4, 64).astype(np.int64)
np.savez_compressed(os.path.join(out_dir, "fw_trace.npz"), t=fw)
t_in, t_loop_end, t_end = fw[:, :, 1], fw[:, :, 2], fw[:, :, 3]
ch = fw[:, :, 4:4 + 48].reshape(4096, 4, 12, 4)
span = t_end.max() - t_in.min()
print(f"FORWARD  kernel span {span} cycles (s_memtime units)")
pct(ch[:, :, 0, 0] - t_in, "prologue (entry -> loop)")
pct(ch[:, :, :, 1] - ch[:, :, :, 0], "stage: A finish + LDS store + bar1")
pct(ch[:, :, :, 2] - ch[:, :, :, 1], "mfma segment (issue)")
pct(ch[:, :, :, 3] - ch[:, :, :, 2], "bar2 wait")
pct(ch[:, :, :, 3] - ch[:, :, :, 0], "chunk total")
pct(t_loop_end - ch[:, :, 0, 0], "main loop total")
pct(t_end - t_loop_end, "epilogue")
pct(t_end - t_in, "workgroup wave lifetime")
idf = ids(fw[:, :, 0].astype(np.uint64))
key = ((idf["xcc"] * 8 + idf["se"]) * 2 + idf["sh"]) * 16 + idf["cu"]
print("  distinct CUs seen:", len(np.unique(key)), " distinct (CU,SIMD):", len(np.unique(key * 4 + idf["simd"])))
fr_any, fr_sum, wg_per = [], [], []
for k in np.unique(key)[:64]:
    for sd in range(4):
        sel = (key == k) & (idf["simd"] == sd)
        if not sel.any(): continue
        seg = [(int(a), int(b)) for a, b in zip(ch[sel][:, :, 1].ravel(), ch[sel][:, :, 2].ravel())]
        lo, hi = t_in[sel].min(), t_end[sel].max()
        fr_any.append(union_len(seg) / (hi - lo)); fr_sum.append(sum(b - a for a, b in seg) / (hi - lo)); wg_per.append(sel.sum())
print(f"  per SIMD: waves {np.mean(wg_per):.1f}; fraction of its span with >=1 wave inside an MFMA segment {np.mean(fr_any):.3f}; "
      f"sum of MFMA-segment time / span {np.mean(fr_sum):.3f}")
# ---------------- dW
dw = read("etm_diag_dw_trace_read", 256 * 8 * 512).reshape(256, 8, 512).astype(np.int64)
np.savez_compressed(os.path.join(out_dir, "dw_trace.npz"), t=dw)
used = dw[:, 0, 1] != 0
dw = dw[used]
n_max = dw[:, :, 2]
it = dw[:, :, 8:8 + 480].reshape(-1, 8, 80, 6)
print(f"dW  workgroups traced {used.sum()}  phases per group {n_max.min()}..{n_max.max()}  kernel span {dw[:, :, 3].max() - dw[:, :, 1].min()}")
for grp in (0, 1):
    s = it[:, grp * 4:(grp + 1) * 4, 2:78]
    print(f" group {grp}")
    pct(s[..., 1] - s[..., 0], "stage: vmcnt wait + LDS stores")
    pct(s[..., 2] - s[..., 1], "issue loads")
    pct(s[..., 3] - s[..., 2], "bar1 wait")
    pct(s[..., 4] - s[..., 3], "mfma segment (issue)")
    pct(s[..., 5] - s[..., 4], "bar2 wait")
    pct(s[..., 5] - s[..., 0], "iteration (= 2 phases)")
