"""GPU probe: error of MIOpen conv fwd/bwd vs CPU fp32 under different backend flags (run on the MI355X box)."""
import itertools, os, sys, torch
torch.manual_seed(0)
x = torch.rand(64, 3, 84, 84)
convs = [torch.nn.Conv2d(3, 32, 8, 4), torch.nn.Conv2d(32, 64, 4, 2), torch.nn.Conv2d(64, 64, 3, 1)]
def run(dev):
    h = x.to(dev).requires_grad_(True)
    mods = [torch.nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride).to(dev) for c in convs]
    for m, c in zip(mods, convs):
        m.load_state_dict(c.state_dict())
    y = h
    for m in mods:
        y = torch.relu(m(y))
    g = torch.sin(torch.arange(y.numel(), dtype=torch.float32).reshape(y.shape)).to(dev)
    (y * g).sum().backward()
    return [y.detach().cpu()] + [p.grad.detach().cpu() for m in mods for p in m.parameters()]
ref = run("cpu")
for tf32, det, bench in itertools.product([True, False], [False, True], [False, True]):
    torch.backends.cudnn.allow_tf32 = tf32
    torch.backends.cudnn.deterministic = det
    torch.backends.cudnn.benchmark = bench
    got = run("cuda")
    errs = [float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(got, ref)]
    print(f"allow_tf32={tf32} deterministic={det} benchmark={bench}: max rel err per tensor " + " ".join(f"{e:.1e}" for e in errs))
