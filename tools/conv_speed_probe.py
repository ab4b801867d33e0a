"""GPU probe: encoder (3 convs) time per call for MIOpen default / benchmark mode / unfold+GEMM, N=32 fwd and N=2048 fwd+bwd."""
import time, torch, torch.nn.functional as F
dev = torch.device("cuda")
torch.manual_seed(0)
convs = [torch.nn.Conv2d(3, 32, 8, 4).to(dev), torch.nn.Conv2d(32, 64, 4, 2).to(dev), torch.nn.Conv2d(64, 64, 3, 1).to(dev)]

def enc_miopen(x):
    for c in convs:
        x = torch.relu(c(x))
    return x

def enc_unfold(x):
    for c in convs:
        n, _, h, w = x.shape
        k, s = c.kernel_size[0], c.stride[0]
        ho, wo = (h - k) // s + 1, (w - k) // s + 1
        cols = F.unfold(x, k, stride=s)                                   # [N, C*k*k, ho*wo]
        y = torch.matmul(c.weight.reshape(c.out_channels, -1), cols) + c.bias.view(1, -1, 1)
        x = torch.relu(y.view(n, c.out_channels, ho, wo))
    return x

def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

x32 = torch.rand(32, 3, 84, 84, device=dev)
x2k = torch.rand(2048, 3, 84, 84, device=dev)
def fwd32(enc):
    with torch.no_grad():
        return enc(x32)
def fb2k(enc):
    y = enc(x2k)
    y.sum().backward()
ref = enc_miopen(x32)
print("unfold vs miopen max abs diff", float((enc_unfold(x32) - ref).abs().max()))
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    print(f"miopen benchmark={bench}: N=32 fwd {timeit(lambda: fwd32(enc_miopen), 50):.3f} ms   N=2048 fwd+bwd {timeit(lambda: fb2k(enc_miopen), 5):.2f} ms")
print(f"unfold+gemm          : N=32 fwd {timeit(lambda: fwd32(enc_unfold), 50):.3f} ms   N=2048 fwd+bwd {timeit(lambda: fb2k(enc_unfold), 5):.2f} ms")
x32cl = x32.contiguous(memory_format=torch.channels_last)
for c in convs:
    c.to(memory_format=torch.channels_last)
x2kcl = x2k.contiguous(memory_format=torch.channels_last)
def fwd32cl():
    with torch.no_grad():
        return enc_miopen(x32cl)
def fb2kcl():
    enc_miopen(x2kcl).sum().backward()
print(f"miopen channels_last (benchmark on): N=32 fwd {timeit(fwd32cl, 50):.3f} ms   N=2048 fwd+bwd {timeit(fb2kcl, 5):.2f} ms")
