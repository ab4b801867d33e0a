"""dW kernel time at the training shape through the library's event facility (pre-added positions = trainer's variant)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import torch
from etm import ops, lib as etm_lib
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
for (tag, k), (ms, c) in sorted(etm_lib.profile_collect().items()):
    print(f"{k:24s} avg_us={ms / c * 1e3:8.1f}")
