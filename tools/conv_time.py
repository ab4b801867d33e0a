"""Training-side encoder: hand-written kernels vs the library convolutions at the minibatch shape (N = 2048, 3 x 84 x 84).
python tools/conv_time.py [N]      prints per-kernel HIP-event times, TFLOP/s against the 157.3 TFLOP/s fp32 MFMA peak, and the
forward+backward wall time of both implementations."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import torch
from etm import lib as etm_lib
if os.environ.get("ETM_DIAG_LIB"):
    etm_lib.LIB_PATH = os.environ["ETM_DIAG_LIB"]
from etm import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda", 0); torch.manual_seed(0)
convs = [torch.nn.Conv2d(3, 32, 8, 4).to(dev), torch.nn.Conv2d(32, 64, 4, 2).to(dev), torch.nn.Conv2d(64, 64, 3, 1).to(dev)]
x = torch.rand((N, 3, 84, 84), device=dev)
x_nhwc = x.permute(0, 2, 3, 1).contiguous()
x_cl = x_nhwc.permute(0, 3, 1, 2)          # NCHW view, channels_last memory
params = [p for c in convs for p in (c.weight, c.bias)]
def mine():
    f = ops.encoder_train(x_nhwc, *convs); return f
def lib_():
    h = x_cl
    for c in convs: h = torch.relu(c(h))
    return h.permute(0, 2, 3, 1).reshape(N, -1)
go = torch.randn((N, 64 * 7 * 7), device=dev)
for name, fn in (("hand-written", mine), ("library", lib_)):
    for _ in range(3):
        torch.autograd.grad(fn(), params, go)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(10):
        e[0].record(); f = fn(); e[1].record(); torch.autograd.grad(f, params, go); e[2].record(); torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print(f"{name:13s} N={N}: forward {tf / 10 * 1e3:8.1f} us   backward {tb / 10 * 1e3:8.1f} us   total {(tf + tb) / 10 * 1e3:8.1f} us", flush=True)
l = etm_lib.load(); etm_lib.profile_collect(); l.etm_profile_enable(1)
for _ in range(5):
    torch.autograd.grad(mine(), params, go)
torch.cuda.synchronize(); l.etm_profile_enable(0)
fl = {"fwd": [N * 400 * 32 * 192 * 2, N * 81 * 64 * 512 * 2, N * 49 * 64 * 576 * 2]}
tot_f = sum(fl["fwd"]); tot_all = fl["fwd"][0] * 2 + fl["fwd"][1] * 3 + fl["fwd"][2] * 3
for (tag, k), (ms, c) in sorted(etm_lib.profile_collect().items()):
    per = ms / 5
    extra = ""
    if k == "conv_train_fwd_kernel": extra = f"  {tot_f / (per * 1e-3) / 1e12:6.1f} TFLOP/s of 157.3"
    if k == "conv_train_wgrad_kernel": extra = f"  {tot_f / (per * 1e-3) / 1e12:6.1f} TFLOP/s of 157.3 (incl. the slice reductions)"
    if k == "conv_train_dgrad_kernel": extra = f"  {(fl['fwd'][1] + fl['fwd'][2]) / (per * 1e-3) / 1e12:6.1f} TFLOP/s of 157.3 (incl. the mask/layout kernel)"
    print(f"  {k:28s} {c // 5:2d} launches per pass, {per * 1e3:8.1f} us per pass{extra}")
print(f"  algorithmic flops per forward+backward: {tot_all / 1e9:.1f} GFLOP")
