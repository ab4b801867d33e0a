"""A column sum inside a captured HIP graph, replayed 2,000 times: torch's own reduction against the library's (round 5).

Why this exists: the optimisation step is one captured graph (trainer.py:_train_step_graph).  Its only torch reduction -- the
fallback `partial.sum(dim=0)` of the norm_kv gradient pass when the grouped column-sum launch was full -- produced garbage
gradients for ONE tensor in SOME replays (cfg2 teacher-forced fixture, second update), depending on the node count of the graph and
on the caching allocator's state at capture time.  torch sums a tall [rows, C] matrix in two stages: a 4-byte memset of a semaphore
(a MEMSET node in the graph) and a reduce kernel whose last-arriving workgroup adds the staged partial sums.  This script holds
nothing of the framework: a [1024, 256] matrix is refilled from one of 64 random sources, summed over its rows inside a captured
graph (between two matrix products), and every replay is compared with a float64 sum.  On ROCm 7.2 / gfx950 with the runtime's
default AQL packet capture, a large share of the replays returns wrong sums; with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 none does, and
the library's fixed-order reduction (etm_colsum_reduce_grouped: no semaphore, no memset node) is exact in both modes -- also behind
a device-to-device MEMCPY node.  A bare MEMSET node in front of a counter increment replays correctly 4,000 times: the memset is the
visible mark of such a reduction in a dumped graph (DEBUG_HIP_GRAPH_DOT_PRINT=1), not the failing element; that is the reduce
kernel's own cross-workgroup hand-over (semaphore + staged partial sums) under the runtime's packet replay.

    python tools/graph_reduce_hazard.py                                  # both reductions, the runtime's default graph launch
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/graph_reduce_hazard.py # the same with packet capture off
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "episodic-transformer-memory-ppo_amd"))
from etm import ops  # noqa: E402


def run(kind, replays=2000, rows=1024, C=256):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    src = torch.randn(64, rows, C, device=dev)
    partial = torch.empty(rows, C, device=dev)
    idx = torch.zeros((), dtype=torch.long, device=dev)
    out = torch.empty(C, device=dev)
    pre = torch.randn(512, 512, device=dev)
    staged = torch.empty_like(partial)

    def through_copy_node(p):               # a device-to-device MEMCPY node (contiguous copy_) between the producer and the library's reduction
        staged.copy_(p)
        return ops.colsum_rows(staged, rows, C)

    reduce = {"torch": lambda p: p.sum(dim=0), "library": lambda p: ops.colsum_rows(p, rows, C), "memcpy+library": through_copy_node}[kind]

    def body():
        y = pre @ pre
        partial.copy_(src.index_select(0, idx.view(1))[0])
        out.copy_(reduce(partial))
        return y @ pre

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    bad = []
    for it in range(replays):
        idx.fill_(it % 64)
        g.replay()
        err = float((out.double() - src[it % 64].double().sum(dim=0)).abs().max())
        if err > 1e-3:                      # (fp32 sums of 1,024 unit normals: ~1e-5)
            bad.append(it)
    return bad


def run_memset(replays=4000):
    """A bare MEMSET node (hipMemsetAsync through ctypes, captured) followed by `cnt += 1`, `acc += cnt`: after every replay cnt must be
    1 -- 0 would be a memset ordered after its consumer, k > 1 a memset that did not run."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    dev = torch.device("cuda:0")
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    big = torch.randn(1024, 1024, device=dev)

    def body():
        y = big @ big
        rc = hip.hipMemsetAsync(cnt.data_ptr(), 0, 4, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        cnt.add_(1)
        return y @ big

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    seen = {}
    for it in range(replays):
        g.replay()
        v = int(cnt.item())
        seen[v] = seen.get(v, 0) + 1
    return seen


if __name__ == "__main__":
    mode = "DEBUG_CLR_GRAPH_PACKET_CAPTURE=" + os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "<unset>")
    print(f"torch {torch.__version__}, hip {torch.version.hip}, {torch.cuda.get_device_name(0)}, {mode}")
    for kind in ("torch", "library", "memcpy+library"):
        bad = run(kind, replays=4000 if kind != "torch" else 2000)
        print(f"{kind:14s} column sum inside a captured graph: {len(bad)} of {4000 if kind != 'torch' else 2000} replays wrong; first {bad[:6]}, last {bad[-3:]}")
    print("bare MEMSET node + increment, value after each of 4000 replays (1 = correct):", run_memset())
