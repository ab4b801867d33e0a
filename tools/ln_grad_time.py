"""Timing of the norm_kv gradient pass (csrc/window_ln_grad.hip) at BASELINE config 5's training shape: N = 2048 samples, L = 128,
D = 384, H = 4, sorted sliding windows over a block-major bank, a third of the window rows without weight (masked).  HIP events
through the C ABI; bytes = the window rows that carry weight, read once."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "episodic-transformer-memory-ppo_amd"))
from etm import lib as etm_lib  # noqa: E402

dev = torch.device("cuda:0")
lib = etm_lib.load()
N, L, D, H, E, T = 2048, 128, 384, 4, 64, 512
torch.manual_seed(0)
bank = torch.randn((E, T, D), device=dev)
ep = torch.arange(N, device=dev) // (N // E)
start = (torch.arange(N, device=dev) % (N // E)) * ((T - L) // (N // E))
win = (start[:, None] + torch.arange(L, device=dev)[None, :]).contiguous()
live = torch.arange(L, device=dev)[None, :] < torch.randint(L // 3, L + 1, (N, 1), device=dev)          # ~2/3 of the rows carry weight
att = (torch.rand((N, H, L), device=dev) * live[:, None, :]).contiguous()
d_e = (torch.randn((N, H, L), device=dev) * live[:, None, :]).contiguous()
u, gz = torch.randn((H, N, D), device=dev), torch.randn((H, N, D), device=dev)
stats = torch.stack((torch.zeros((N, L), device=dev), torch.ones((N, L), device=dev)), dim=-1).contiguous()
rows = lib.etm_window_ln_grad_rows(N)
partial = torch.empty((rows, 2 * D), device=dev)
st = torch.cuda.current_stream().cuda_stream


def call():
    rc = lib.etm_window_ln_grad(bank.data_ptr(), T * D, D, ep.data_ptr(), win.data_ptr(), None, None, stats.data_ptr(), att.data_ptr(), d_e.data_ptr(),
                                u.data_ptr(), gz.data_ptr(), N * D, D, partial.data_ptr(), N, L, D, H, st)
    assert rc == 0, rc


for _ in range(5):
    call()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    call()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
nbytes = float(live.sum()) * D * 4
print(f"window_ln_grad_kernel N={N} L={L} D={D} H={H}: {us:.1f} us per launch, {nbytes / 1e6:.0f} MB of live window rows -> {nbytes / us / 1e6:.2f} TB/s "
      f"(checksum {float(partial.double().sum()):.6e})")
