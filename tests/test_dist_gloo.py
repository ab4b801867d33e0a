"""Data-parallel plumbing (etm/dist.py) on CPU: world_size 2 over gloo, 127.0.0.1 rendezvous."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "episodic-transformer-memory-ppo_amd")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, PKG)
    from etm.dist import DataParallel
    dp = DataParallel(device=torch.device("cpu"), backend="gloo")
    assert dp.active and dp.rank == rank and dp.world == world
    first, count = dp.shard(64)
    assert (first, count) == (rank * 32, 32)

    # replicas start identical after the broadcast
    torch.manual_seed(100 + rank)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    dp.broadcast_parameters(model)
    params = list(model.parameters())
    flat = dp.attach_flat_grads(params)
    assert flat.numel() == sum(p.numel() for p in params)

    # shard the same global batch; averaged shard gradients == gradient of the global mean loss
    g = torch.Generator().manual_seed(7)
    x = torch.randn((64, 5), generator=g)
    y = torch.randn((64, 3), generator=g)
    flat.zero_()
    loss = ((model(x[first:first + count]) - y[first:first + count]) ** 2).mean()
    loss.backward()
    assert all(p.grad.data_ptr() >= flat.data_ptr() for p in params)   # grads still alias the bucket
    dp.all_reduce_grads()
    ref = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    ref.load_state_dict(model.state_dict())
    ((ref(x) - y) ** 2).mean().backward()
    ref_flat = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    assert torch.allclose(flat, ref_flat, atol=1e-6), (flat - ref_flat).abs().max()

    # the overlapped step sums the bucket in two slices (trainer.py: dp_overlap): slice sums == the whole bucket's sum, bit for bit
    dp.flat = torch.arange(1000, dtype=torch.float32) * (rank + 1) / 7.0
    whole = dp.flat.clone()
    dp.all_reduce_slice(37, 1000, None)
    dp.all_reduce_slice(0, 37, None)
    parts = dp.flat.clone()
    dp.flat = whole
    dp.all_reduce_grads(average=False)
    assert torch.equal(parts, dp.flat)
    dp.flat = flat

    # advantage statistics: merged (count, mean, M2) == statistics of the concatenated minibatch
    adv = torch.randn((100,), generator=g) * 3 + 1
    mine = adv[rank * 50:(rank + 1) * 50]
    local = torch.stack([torch.tensor(50.0), mine.mean(), ((mine - mine.mean()) ** 2).sum()])
    merged = dp.merge_adv_stats(local)
    assert torch.allclose(merged, torch.stack([torch.tensor(100.0), adv.mean(), ((adv - adv.mean()) ** 2).sum()]), rtol=1e-5)
    assert abs(float(torch.sqrt(merged[2] / (merged[0] - 1))) - float(adv.std())) < 1e-5

    assert dp.max_over_ranks(float(rank + 1)) == float(world)
    # round 6: the agreements around the one-graph data-parallel step -- a logical AND over the ranks (one "no" anywhere = "no" everywhere);
    # over gloo the library collective is not in use, so the collective can never be captured inside the step's graph
    assert dp.agree(True) is True and dp.agree(rank != 1) is False and dp.agree(False) is False
    assert dp.collective == "torch" and dp.graph_collective_ok() is False
    dp.barrier()
    dp.close()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


def test_data_parallel_world_size_2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def _worker4(rank, world, port, out_dir):
    """World size 4 with UNEVEN per-rank sample counts: the merged advantage statistics and the summed-then-scaled gradient (the
    trainer's all_reduce_grads(average=False) + grad_scale in the optimiser step) against the global quantities."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, PKG)
    from etm.dist import DataParallel
    dp = DataParallel(device=torch.device("cpu"), backend="gloo")      # collective 'etm' requested by default: must agree on torch
    assert dp.active and dp.world == 4 and dp.collective == "torch" and abs(dp.grad_scale - 0.25) < 1e-12
    g = torch.Generator().manual_seed(11)
    counts = [10, 25, 40, 25]                                          # e.g. ranks whose minibatches hold different sample counts
    adv = torch.randn((sum(counts),), generator=g) * 2 - 0.5
    lo = sum(counts[:rank])
    mine = adv[lo: lo + counts[rank]]
    # k = 3 independent minibatches merged row-wise with ONE all-gather (what the trainer does per epoch)
    rows = []
    for k in range(3):
        m = mine * (k + 1)
        rows.append(torch.stack([torch.tensor(float(m.numel())), m.mean(), ((m - m.mean()) ** 2).sum()]))
    merged = dp.merge_adv_stats(torch.stack(rows))
    for k in range(3):
        a = adv * (k + 1)
        want = torch.stack([torch.tensor(float(a.numel())), a.mean(), ((a - a.mean()) ** 2).sum()])
        assert torch.allclose(merged[k], want, rtol=1e-5, atol=1e-5), (k, merged[k], want)
        assert abs(float(torch.sqrt(merged[k][2] / (merged[k][0] - 1))) - float(a.std())) < 1e-4      # torch.std: unbiased
    # gradient exchange: sum over ranks, 1 / world applied by the consumer
    dp.flat = torch.full((1000,), float(rank + 1))
    dp.all_reduce_grads(average=False)
    assert torch.equal(dp.flat, torch.full((1000,), 10.0))
    assert torch.allclose(dp.flat * dp.grad_scale, torch.full((1000,), 2.5))
    dp.flat = torch.full((7,), float(rank))
    dp.all_reduce_grads()                                              # average=True: the stand-alone form
    assert torch.allclose(dp.flat, torch.full((7,), 1.5))
    assert dp.max_over_ranks(float(10 - rank)) == 10.0
    dp.barrier()
    dp.close()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


def test_data_parallel_world_size_4_uneven_counts(tmp_path):
    world, port = 4, _free_port()
    mp.spawn(_worker4, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1", "ok2", "ok3"]


def test_single_process_is_passthrough():
    sys.path.insert(0, PKG)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    from etm.dist import DataParallel
    dp = DataParallel(device=torch.device("cpu"))
    assert not dp.active and dp.shard(32) == (0, 32)
    s = torch.tensor([4.0, 1.0, 2.0])
    assert dp.merge_adv_stats(s) is s and dp.max_over_ranks(3.5) == 3.5


def test_bench_multi_process_plumbing_on_cpu():
    """The driver's multi-GPU invocation of bench.py (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...):
    rank / world parsing, process group, sharding, barriers, flat-bucket all-reduce, max over ranks and the rank-0 JSON line,
    with CPU tensors over gloo (--plumbing-check: the trainer itself needs the GPU)."""
    import json
    import subprocess
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--plumbing-check"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # exactly one JSON line, from rank 0
    assert [l for l in out.stdout.splitlines() if l.strip()] == lines, out.stdout      # ... and nothing else on stdout (bench.py contract)
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and abs(rec["allreduce_mean"] - 1.5) < 1e-6
