"""Training-side encoder kernels one by one (N = 2048, 3 x 84 x 84): HIP-event time and TFLOP/s of forward / backward-data /
backward-weight of every layer through the C ABI (tools/kernel_rooflines.py encoder is the bench.py form of the same figures).
python tools/conv_layer_time.py [N] [--fwd-lds] [--wgrad-lds] [--dgrad-lds]   (--fwd-lds: ALL forward passes with the images resident in LDS, csrc/conv_fwd_lds.hip;
without: none of them -- the library default is layer 2 only; --wgrad-lds: ALL weight gradients through csrc/conv_wgrad_lds.hip, without:
none -- the default is layer 1 only)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import torch
from etm import lib as etm_lib
from etm import ops
FWD_LDS = "--fwd-lds" in sys.argv
_args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(_args[0]) if _args else 2048
dev = torch.device("cuda", 0); torch.manual_seed(0)
lib = etm_lib.load()
etm_lib.check(lib.etm_conv_train_set_fwd_lds(7 if FWD_LDS else 0), "set_fwd_lds")
etm_lib.check(lib.etm_conv_train_set_dgrad_lds(1 if "--dgrad-lds" in sys.argv else 0), "set_dgrad_lds")
etm_lib.check(lib.etm_conv_train_set_wgrad_lds(7 if "--wgrad-lds" in sys.argv else 0), "set_wgrad_lds")
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
layers = [(3, 84, 84, 32, 8, 4), (32, 20, 20, 64, 4, 2), (64, 9, 9, 64, 3, 1)]
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = 0.0
for li, (c, h, w, cout, k, s) in enumerate(layers):
    ho, wo = (h - k) // s + 1, (w - k) // s + 1
    x = torch.rand((N, h, w, c), device=dev)
    wt = torch.randn((cout, c, k, k), device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    y = torch.empty((N, ho, wo, cout), device=dev)
    dy = torch.randn((N, ho, wo, cout), device=dev)
    packed = ops.conv_pack_weights(wt.permute(0, 2, 3, 1).reshape(cout, -1))
    fl = 2.0 * N * ho * wo * cout * k * k * c
    t = timed(lambda: etm_lib.check(lib.etm_conv_train_fwd(P(x), None, N, P(packed), P(b), P(y), N, c, h, w, cout, k, k, s, 0, st), "fwd"))
    print(f"conv{li + 1} forward   {t:8.1f} us  {fl / t / 1e6:6.1f} TFLOP/s ({fl / t / 1e6 / 157.3:.2f} of peak)"); tot += t
    if li > 0:
        pd = ops.conv_pack_dgrad_weights(wt, s)
        dx = torch.empty((N, h, w, c), device=dev)
        t = timed(lambda: etm_lib.check(lib.etm_conv_train_dgrad(P(dy), P(pd), P(x), P(dx), N, c, h, w, cout, k, k, s, st), "dgrad"))
        print(f"conv{li + 1} bwd-data  {t:8.1f} us  {fl / t / 1e6:6.1f} TFLOP/s ({fl / t / 1e6 / 157.3:.2f} of peak)"); tot += t
    K = k * k * c
    buf = torch.empty(K * cout + cout, device=dev)
    nbytes = lib.etm_conv_train_wgrad_workspace_bytes(N, c, h, w, cout, k, k, s)
    ws = torch.empty(max(nbytes, 8) // 4, device=dev)
    t = timed(lambda: etm_lib.check(lib.etm_conv_train_wgrad(P(x), None, P(dy), P(buf), P(ws), nbytes, N, c, h, w, cout, k, k, s, st), "wgrad"))
    print(f"conv{li + 1} bwd-weight {t:7.1f} us  {fl / t / 1e6:6.1f} TFLOP/s ({fl / t / 1e6 / 157.3:.2f} of peak)  (incl. slice reduction)"); tot += t
print(f"sum {tot:8.1f} us")
