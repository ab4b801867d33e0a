"""Encoder passes on the bf16 matrix pipe (csrc/conv_b3.hip) one by one: error against float64 next to the fp32-MFMA kernels' error,
and HIP-event time of both at N = 2048 (3 x 84 x 84), every call through the C ABI.
python tools/conv_b3_check.py [N] [--no-time]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import torch
import torch.nn.functional as F
from etm import lib as etm_lib
from etm import ops
_args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(_args[0]) if _args else 2048
NREF = 96                                   # images checked against float64 (CPU convolutions)
dev = torch.device("cuda", 0); torch.manual_seed(0)
lib = etm_lib.load()
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


def rel(a, ref):
    return ((a.double().cpu() - ref).norm() / ref.norm()).item()


def b3_pack(ws, dgrad, strides):
    outs = [torch.empty(3 * w.numel(), dtype=torch.int16, device=dev) for w in ws]
    n = len(ws)
    vp = lambda ts: (ctypes.c_void_p * n)(*[P(t) for t in ts])
    ia = lambda vs: (ctypes.c_int32 * n)(*vs)
    etm_lib.check(lib.etm_conv_b3_pack(vp(ws), vp(outs), ia(dgrad), ia([w.shape[0] for w in ws]), ia([w.shape[1] for w in ws]),
                                       ia([w.shape[2] for w in ws]), ia(strides), n, st), "etm_conv_b3_pack")
    return outs


tot_old = tot_new = 0.0
for li, (c, h, w, cout, k, s) in enumerate(layers):
    ho, wo = (h - k) // s + 1, (w - k) // s + 1
    x = torch.rand((N, h, w, c), device=dev) if li == 0 else torch.relu(torch.randn((N, h, w, c), device=dev))
    wt = torch.randn((cout, c, k, k), device=dev) * 0.05
    b = torch.randn(cout, device=dev) * 0.1
    dy = torch.randn((N, ho, wo, cout), device=dev) * (torch.rand((N, ho, wo, cout), device=dev) > 0.5)
    fl = 2.0 * N * ho * wo * cout * k * k * c
    # ---- forward
    packed = ops.conv_pack_weights(wt.permute(0, 2, 3, 1).reshape(cout, -1))
    wb3, = b3_pack([wt], [0], [s])
    y32 = torch.empty((N, ho, wo, cout), device=dev); yb3 = torch.full((N, ho, wo, cout), float("nan"), device=dev)
    f32 = lambda: etm_lib.check(lib.etm_conv_train_fwd(P(x), None, N, P(packed), P(b), P(y32), N, c, h, w, cout, k, k, s, 0, st), "fwd")
    ybits = torch.zeros((N, ho, wo, cout // 32), dtype=torch.int32, device=dev)
    fb3 = lambda: etm_lib.check(lib.etm_conv_b3_fwd(P(x), None, P(wb3), P(b), P(yb3), P(ybits), N, c, h, w, cout, k, k, s, st), "b3 fwd")
    f32(); fb3(); torch.cuda.synchronize()
    nr = min(N, NREF)
    sel = torch.cat([torch.arange(nr // 2), torch.arange(N - (nr - nr // 2), N)])          # first and last images
    ref = torch.relu(F.conv2d(x[sel].double().cpu().permute(0, 3, 1, 2), wt.double().cpu(), b.double().cpu(), stride=s)).permute(0, 2, 3, 1)
    line = f"conv{li + 1} forward   : err vs float64  fp32 MFMA {rel(y32[sel], ref):.2e}   bf16x3 {rel(yb3[sel], ref):.2e}   b3 vs fp32 (all N) {((yb3 - y32).norm() / y32.norm()).item():.2e}"
    if "--no-time" not in sys.argv:
        t0, t1 = timed(f32), timed(fb3); tot_old += t0; tot_new += t1
        line += f"   {t0:6.1f} -> {t1:6.1f} us  ({fl / t1 / 1e6:5.1f} fp32-equivalent TFLOP/s)"
    print(line, flush=True)
    yb = ((yb3 > 0).view(N, ho, wo, cout // 32, 32).to(torch.int64) << torch.arange(32, device=dev)).sum(-1)
    print(f"conv{li + 1} forward ReLU pattern words: identical to (y > 0) packed on the host: {bool((torch.where(yb >= 2 ** 31, yb - 2 ** 32, yb).to(torch.int32) == ybits).all().item())}")
    # ---- forward through the minibatch index (layer 1)
    if li == 0:
        idx = torch.randperm(N, device=dev)
        yi = torch.full((N, ho, wo, cout), float("nan"), device=dev)
        etm_lib.check(lib.etm_conv_b3_fwd(P(x), P(idx), P(wb3), P(b), P(yi), None, N, c, h, w, cout, k, k, s, st), "b3 fwd idx")
        print(f"conv1 forward through x_index: identical to the gathered run: {bool((yi == yb3[idx]).all().item())}")
    # ---- backward-data
    if li > 0:
        pd = ops.conv_pack_dgrad_weights(wt, s)
        wd3, = b3_pack([wt], [1], [s])
        dx32 = torch.empty((N, h, w, c), device=dev); dxb3 = torch.full((N, h, w, c), float("nan"), device=dev)
        d32 = lambda: etm_lib.check(lib.etm_conv_train_dgrad(P(dy), P(pd), P(x), P(dx32), N, c, h, w, cout, k, k, s, st), "dgrad")
        # the ReLU pattern of the layer below as bits (what its forward pass writes): here built from x itself
        xb = ((x > 0).view(N, h, w, c // 32, 32).to(torch.int64) << torch.arange(32, device=dev)).sum(-1)
        xbits = torch.where(xb >= 2 ** 31, xb - 2 ** 32, xb).to(torch.int32).contiguous()
        db3 = lambda: etm_lib.check(lib.etm_conv_b3_dgrad(P(dy), None, P(wd3), None, P(xbits), P(dxb3), N, c, h, w, cout, k, k, s, st), "b3 dgrad")
        dxv = torch.full((N, h, w, c), float("nan"), device=dev)
        etm_lib.check(lib.etm_conv_b3_dgrad(P(dy), None, P(wd3), P(x), None, P(dxv), N, c, h, w, cout, k, k, s, st), "b3 dgrad by values")
        d32(); db3(); torch.cuda.synchronize()
        ref = F.conv_transpose2d(dy[sel].double().cpu().permute(0, 3, 1, 2), wt.double().cpu(), stride=s).permute(0, 2, 3, 1) * (x[sel].double().cpu() > 0)
        line = f"conv{li + 1} bwd-data  : err vs float64  fp32 MFMA {rel(dx32[sel], ref):.2e}   bf16x3 {rel(dxb3[sel], ref):.2e}   b3 vs fp32 (all N) {((dxb3 - dx32).norm() / dx32.norm()).item():.2e}"
        if "--no-time" not in sys.argv:
            t0, t1 = timed(d32), timed(db3); tot_old += t0; tot_new += t1
            line += f"   {t0:6.1f} -> {t1:6.1f} us  ({fl / t1 / 1e6:5.1f} fp32-equivalent TFLOP/s)"
        print(line, flush=True)
        dxn = torch.full((N, h, w, c), float("nan"), device=dev)
        etm_lib.check(lib.etm_conv_b3_dgrad(P(dy), None, P(wd3), None, None, P(dxn), N, c, h, w, cout, k, k, s, st), "b3 dgrad nomask")
        refn = F.conv_transpose2d(dy[sel].double().cpu().permute(0, 3, 1, 2), wt.double().cpu(), stride=s).permute(0, 2, 3, 1)
        print(f"conv{li + 1} bwd-data without mask: err vs float64 {rel(dxn[sel], refn):.2e}; pattern from y_below's values identical to the bits form: {bool((dxv == dxb3).all().item())}", flush=True)
    # ---- backward-weight (slices + the grouped reduction, both paths)
    K = k * k * c
    buf32 = torch.empty(K * cout + cout, device=dev)
    nbytes = lib.etm_conv_train_wgrad_workspace_bytes(N, c, h, w, cout, k, k, s)
    ws32 = torch.empty(max(nbytes, 8) // 4, device=dev)
    w32 = lambda: etm_lib.check(lib.etm_conv_train_wgrad(P(x), None, P(dy), P(buf32), P(ws32), nbytes, N, c, h, w, cout, k, k, s, st), "wgrad")
    slices = lib.etm_conv_b3_wgrad_slices(N, c, h, w, cout, k, k, s)
    ws3 = torch.full((slices * (K * cout + cout),), float("nan"), device=dev)
    dw3 = torch.full((cout, c, k, k), float("nan"), device=dev); db3 = torch.full((cout,), float("nan"), device=dev)
    one = lambda ct, v: (ct * 1)(v)
    def wb3(xi=None):
        etm_lib.check(lib.etm_conv_b3_wgrad(P(x), xi, P(dy), None, P(ws3), ws3.numel() * 4, N, c, h, w, cout, k, k, s, st), "b3 wgrad")
        etm_lib.check(lib.etm_conv_wgrad_reduce_grouped(one(ctypes.c_void_p, P(ws3)), one(ctypes.c_int32, slices), one(ctypes.c_void_p, P(dw3)),
                                                        one(ctypes.c_void_p, P(db3)), one(ctypes.c_int32, cout), one(ctypes.c_int32, c),
                                                        one(ctypes.c_int32, k), one(ctypes.c_int32, k), 1, st), "reduce")
    w32(); wb3(); torch.cuda.synchronize()
    cols = F.unfold(x.permute(0, 3, 1, 2).double(), k, stride=s)                       # [N, c k k, pixels], float64 on the device
    ref_dw = torch.einsum("nkp,npo->ok", cols, dy.double().reshape(N, ho * wo, cout)).reshape(cout, c, k, k).cpu()
    ref_db = dy.double().sum((0, 1, 2)).cpu()
    del cols
    dw32 = buf32[: K * cout].view(cout, c, k, k)
    line = (f"conv{li + 1} bwd-weight: err vs float64  fp32 MFMA {rel(dw32, ref_dw):.2e}   bf16x3 {rel(dw3, ref_dw):.2e}   bias gradient fp32 {rel(buf32[K * cout:], ref_db):.2e}"
            f"   bf16x3 path {rel(db3, ref_db):.2e}")
    if "--no-time" not in sys.argv:
        t0, t1 = timed(w32), timed(wb3); tot_old += t0; tot_new += t1
        line += f"   {t0:6.1f} -> {t1:6.1f} us  ({fl / t1 / 1e6:5.1f} fp32-equivalent TFLOP/s, reductions included)"
    print(line, flush=True)
    if li == 0:
        idx = torch.arange(N, device=dev)
        keep = dw3.clone()
        wb3(P(idx)); torch.cuda.synchronize()
        print(f"conv1 bwd-weight through x_index (identity): identical {bool((dw3 == keep).all().item())}")
if "--no-time" not in sys.argv:
    print(f"sum of the passes above: fp32 MFMA {tot_old:.1f} us -> bf16x3 {tot_new:.1f} us")
