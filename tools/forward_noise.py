"""Stage-by-stage forward precision of the HIP path at the config-3 model size, against float64 (round-4 parity decomposition).

Every stage gets THE SAME fp32 inputs on the device and in float64; printed: ||fp32 result - float64 result|| / ||float64 result||,
for the device kernels / library products and for the CPU's fp32 evaluation of the same stage (what the reference runs).

    python tools/forward_noise.py [N]
"""
import os
import sys

os.environ.setdefault("ETM_TUNABLE_GEMM", "0")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "episodic-transformer-memory-ppo_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from etm import ops  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    conv1, conv2, conv3 = torch.nn.Conv2d(3, 32, 8, 4), torch.nn.Conv2d(32, 64, 4, 2), torch.nn.Conv2d(64, 64, 3, 1)
    lin = torch.nn.Linear(3136, 384)
    obs = torch.rand(n, 3, 84, 84)
    # ---- encoder: float64 on the CPU, fp32 on the CPU (library convolutions = the reference), fp32 hand-written kernels
    with torch.no_grad():
        x64 = obs.double()
        acts64 = []
        for c in (conv1, conv2, conv3):
            x64 = F.relu(F.conv2d(x64, c.weight.double(), c.bias.double(), c.stride))
            acts64.append(x64)
        x32 = obs
        acts32 = []
        for c in (conv1, conv2, conv3):
            x32 = F.relu(c(x32))
            acts32.append(x32)
        print("encoder, CPU fp32 (library convolutions) vs float64, layer 1 / 2 / 3:", " ".join(f"{rel(a, b):.2e}" for a, b in zip(acts32, acts64)))
        c1, c2, c3 = (torch.nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride).to(dev) for c in (conv1, conv2, conv3))
        for d, s in zip((c1, c2, c3), (conv1, conv2, conv3)):
            d.load_state_dict(s.state_dict())
        feats = ops.encoder_train(obs.to(dev).permute(0, 2, 3, 1).contiguous(), c1, c2, c3)         # [N, 7*7*64] NHWC-flattened
        ref = acts64[2].permute(0, 2, 3, 1).reshape(n, -1)
        print(f"encoder, hand-written fp32-MFMA kernels (3 layers) vs float64: {rel(feats, ref):.2e}")
        # the same through layer-by-layer float64 inputs is not possible (the kernels chain internally); library convolutions on the device:
        xg = obs.to(dev)
        for i, c in enumerate((c1, c2, c3)):
            xg = F.relu(c(xg))
            print(f"encoder, device library convolution chain up to layer {i + 1} vs float64: {rel(xg, acts64[i]):.2e}")
        # ---- lin_hidden: [N, 3136] x [3136, 384]
        a = torch.randn(n, 3136) * 0.3
        for name, K, M in (("lin_hidden [N,3136]x[3136,384]", 3136, 384), ("block product [N,384]x[384,384]", 384, 384)):
            a = torch.randn(n, K) * 0.3
            w = torch.randn(M, K) / K ** 0.5
            r64 = a.double() @ w.double().t()
            print(f"{name}: CPU fp32 {rel(a @ w.t(), r64):.2e}   device library GEMM {rel(a.to(dev) @ w.to(dev).t(), r64):.2e}")
        # dW-shaped product: [384, N] x [N, 384]
        for nn_ in (n, 2560):
            a, b = torch.randn(nn_, 384), torch.randn(nn_, 384)
            r64 = a.double().t() @ b.double()
            print(f"weight-gradient product [384,{nn_}]x[{nn_},384]: CPU fp32 {rel(a.t() @ b, r64):.2e}   device library GEMM {rel(a.to(dev).t() @ b.to(dev), r64):.2e}")
        # ---- LayerNorm
        x = torch.randn(n, 384)
        ln = torch.nn.LayerNorm(384)
        r64 = F.layer_norm(x.double(), (384,), ln.weight.double(), ln.bias.double())
        print(f"LayerNorm [N,384]: CPU fp32 {rel(ln(x), r64):.2e}   device library {rel(F.layer_norm(x.to(dev), (384,), ln.weight.to(dev), ln.bias.to(dev)), r64):.2e}")


if __name__ == "__main__":
    main()
