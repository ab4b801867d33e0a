"""Forward kernel at the training shape with and without the K/V save epilogue (no-grad = no save)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import torch
from etm import ops
dev = torch.device("cuda"); torch.manual_seed(0)
N, L, D, H, T, nb, E = 2048, 64, 384, 4, 96, 3, 416
bank = torch.randn((E, T, nb, D), device=dev)
ep = torch.randint(0, E, (N,), device=dev)
win = torch.randint(0, T - L + 1, (N, 1), device=dev) + torch.arange(L, device=dev)[None, :]
mask = torch.arange(L, device=dev)[None, :] < torch.randint(0, L, (N,), device=dev)[:, None]
wk = torch.randn((D, D), device=dev) / D ** 0.5; wv = torch.randn((D, D), device=dev) / D ** 0.5
q = torch.randn((N, D), device=dev)
spec = ops.WindowSpec.from_bank(bank, ep, win, None, mask)
def run(grad):
    qq = q.clone().requires_grad_(grad)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    tot = 0
    for i in range(12):
        ev[0].record()
        if grad:
            ops.mha(qq, wk, wv, spec, 1, H)
        else:
            with torch.no_grad():
                ops.mha(qq, wk, wv, spec, 1, H)
        ev[1].record(); torch.cuda.synchronize()
        if i >= 2: tot += ev[0].elapsed_time(ev[1])
    return tot / 10
print(f"fwd with K/V save {run(True):.3f} ms, without {run(False):.3f} ms (no positional rows)")
