"""Tuned plain GEMM vs the bias / activation epilogue variants (torch.cuda.tunable on), rollout and training shapes."""
import torch
import torch.cuda.tunable as tunable
tunable.enable(True); tunable.tuning_enable(True); tunable.set_filename("/tmp/probe_tunable.csv", True)
dev = torch.device("cuda")
def timeit(fn, n=100):
    for _ in range(10): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n // 10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    for M, K, N in ((32, 384, 384), (32, 3136, 384), (2048, 384, 384), (2048, 3136, 384), (2048, 384, 1024)):
        x = torch.randn((M, K), device=dev); lin = torch.nn.Linear(K, N).to(dev)
        wt = lin.weight.t()
        g = torch.randn((M, N), device=dev)
        t_mm = timeit(lambda: torch.nn.functional.linear(x, lin.weight))
        t_addmm = timeit(lambda: torch.addmm(lin.bias, x, wt))
        t_2 = timeit(lambda: torch.nn.functional.linear(x, lin.weight).add_(lin.bias))
        t_dw = timeit(lambda: g.t() @ x)          # weight gradient shape
        t_dx = timeit(lambda: g @ lin.weight)     # input gradient shape
        print(f"M={M} K={K} N={N}: mm {t_mm:.1f} us | addmm {t_addmm:.1f} | mm + add_ {t_2:.1f} | dW {t_dw:.1f} | dX {t_dx:.1f}")
