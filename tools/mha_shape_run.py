"""Run kernel #1 forward + backward a few times at the BASELINE config (3) training shape (for rocprofv3 passes).

    python tools/mha_shape_run.py [iters] [N] [L] [D] [H]        (env ETM_ATTENTION=folded|dense, ETM_POS=1 for in-kernel positions,
    ETM_WIN_SAMPLES=random|shuffled|sorted, ETM_DIAG_LIB / ETM_WIN_XCD_MAP for the candidate build)
Prints the average wall time per fwd / bwd call measured with torch events."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import torch
from etm import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N, L, D, H = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (2048, 64, 384, 4)
dev = torch.device("cuda")
torch.manual_seed(0)
T, nb, E = max(96, L), 3, 416
bank = torch.randn((E, T, nb, D), device=dev)
mode = os.environ.get("ETM_WIN_SAMPLES", "random")      # same access patterns as tools/window_time.py
if mode == "random":
    ep = torch.randint(0, E, (N,), device=dev)
    win = torch.randint(0, T - L + 1, (N, 1), device=dev) + torch.arange(L, device=dev)[None, :]
    mask = torch.arange(L, device=dev)[None, :] < torch.randint(0, L, (N,), device=dev)[:, None]
else:                                                    # training-like: N of the E * T (episode, step) pairs, sliding windows
    pairs = torch.randperm(E * T, device=dev)[:N]
    if mode == "sorted":
        pairs = pairs.sort().values
    ep, step = pairs // T, pairs % T
    win = torch.clamp(step - (L - 1), min=0)[:, None] + torch.arange(L, device=dev)[None, :]
    mask = torch.arange(L, device=dev)[None, :] < torch.clamp(step, max=L - 1)[:, None]
pos = torch.randn((T, D), device=dev) * 0.5
wk = (torch.randn((D, D), device=dev) / D ** 0.5).requires_grad_(True)
wv = (torch.randn((D, D), device=dev) / D ** 0.5).requires_grad_(True)
q = torch.randn((N, D), device=dev).requires_grad_(True)
gout = torch.randn((N, D), device=dev)
ops.set_attention_impl(os.environ.get("ETM_ATTENTION", "folded"))
use_pos = os.environ.get("ETM_POS", "0") == "1"      # the trainer pre-adds the positional rows to the bank (no in-kernel positions)
spec = ops.WindowSpec.from_bank(bank, ep, win, win if use_pos else None, mask)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for it in range(iters + 2):
    ev[0].record()
    out, att = ops.mha(q, wk, wv, spec, 1, H, pos=pos if use_pos else None)
    ev[1].record()
    (out * gout).sum().backward()
    ev[2].record()
    torch.cuda.synchronize()
    if it >= 2:
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
fl = N * 2.0 * (2 * L * D * D + 2 * L * D)
print(f"N={N} L={L} D={D} H={H}: fwd {tf / iters:.3f} ms ({fl / (tf / iters * 1e-3) / 1e12:.1f} TF), bwd(all) {tb / iters:.3f} ms")
