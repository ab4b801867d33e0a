#!/usr/bin/env python3
"""Throughput bench of the MI355X PPO + TransformerXL path (driver contract: see the repo prompt / DESIGN.md).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full PPO update of BASELINE config (3)/(4): a rollout of n_workers x worker_steps = 32 x 512
environment steps per GPU on synthetic 3x84x84 observations, GAE, and epochs x n_mini_batch = 5 x 8 optimiser steps
on minibatches of 2048 samples (configs/synthetic_minigrid.yaml).  Weak scaling: every rank owns its own 32
environments; gradients are all-reduced with RCCL.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     the dominant kernel's achieved rate from HIP events recorded around its launches inside the timed
               region (libetm_hip.so's etm_profile_* facility): the folded window-attention pass against the HBM peak,
               bytes_per_launch = N * L * D * 4 -- or, with --attention dense, the fp32-MFMA
               contraction against the fp32 MFMA peak of gfx950;
  rooflines    kernel micro-benchmarks run AFTER the timed region (tools/kernel_rooflines.py): window pass with every byte
               from HBM and with the training access pattern, the dense north-star attention kernels against the fp32
               MFMA peak at config 3 and config 5 dims, GAE and PPO loss against the HBM peak at config and scaled sizes;
  cpu_baseline the CPU oracle (oracle/ref_algo.OracleTrainer, a validated restatement of the reference trainer)
               timed on this host's cores on a bounded sample of the same workload (rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

# the host driver shares device memory between the ranks of a node through dmabuf handles only (RCCL, multi-process runs);
# must be in the environment before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# eight hardware queues: four rollout worker groups (two streams each) run concurrently (trainer.py, rollout_groups: auto);
# read by the HIP runtime when it starts, like the variable above
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("ETM_HW_QUEUES_SET_EARLY", "1")   # marker for trainer.py: the line above ran before the HIP runtime started

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "episodic-transformer-memory-ppo_amd")
for _p in (REPO, PKG, os.path.join(REPO, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
HBM_PEAK_GBS = 8000.0


CONFIG_NAME = "synthetic_minigrid"      # --config: another YAML of configs/ (synthetic_mortar_gtrxl = BASELINE config 5's shape, synthetic_cartpole = config 2's)


def load_config():
    from yaml_parser import YamlParser
    return YamlParser(os.path.join(PKG, "configs", CONFIG_NAME + ".yaml")).get_config()


def kernel_work(name, N, L, D, H):
    """Algorithmic work of one launch at the training shape (DESIGN.md section 'Kernels'): ("hbm", bytes) or ("mfma", flops)."""
    if name.startswith("conv_") and name[-1] in "123" and "layer" in name:
        # encoder pass of one layer: 2 * N * Ho * Wo * k * k * C * Cout flop (forward, backward-data and backward-weight alike; the
        # slice reduction of the weight gradients is a launch of its own with its own profile id since round 3)
        import kernel_rooflines
        return "mfma", kernel_rooflines.encoder_flops(N)[int(name[-1]) - 1]
    if name == "grouped_dw_kernel":
        # every dense-layer weight gradient of the step in one launch: the flops of the problems the collector really handed over
        # (etm/ops.py:DeferredDw.flush; config 3: 3 blocks x 5 + embedding + 2 hidden heads = 18 products of 2 * N * D * D)
        from etm import ops
        return "mfma", ops.DeferredDw.last_flops or 2.0 * N * D * D * 18
    if name in ("window_fwd_kernel", "window_bwd_kernel"):
        # folded attention pass: the read of the gathered window, L*D*4 B per sample and block (SURVEY.md 8d's algorithmic
        # figure; the H folded vectors in / out and the attention weights are reported separately as extra_bytes_per_launch)
        return "hbm", N * 4.0 * L * D
    if name == "mha_fwd_kernel":      # dense variant: K and V projections of the window + QK^T + att.V
        return "mfma", N * 2.0 * (2 * L * D * D + 2 * L * D)
    if name == "bwd_dw_kernel":       # dense variant: dWk and dWv: [2D, N*L] x [N*L, D]
        return "mfma", N * 2.0 * (2 * L * D * D)
    return None


# profile name of this build -> kernel symbol (prefix) in the rocprofv3 counter files
PMC_SYMBOL = {"window_fwd_kernel": "window_pass_kernel", "window_bwd_kernel": "window_pass_kernel",
              # (round 6: the encoder passes on the bf16 matrix pipe, csrc/conv_b3.hip / conv_b3_wgrad.hip; B3Geo<DGRAD, C, HW, ...>)
              "conv_fwd_layer1": "conv_b3_kernel<B3Geo<false, 3, 84", "conv_fwd_layer2": "conv_b3_kernel<B3Geo<false, 32, 20",
              "conv_fwd_layer3": "conv_b3_kernel<B3Geo<false, 64, 9", "conv_dgrad_layer2": "conv_b3_kernel<B3Geo<true, 32, 20",
              "conv_dgrad_layer3": "conv_b3_kernel<B3Geo<true, 64, 9", "conv_wgrad_layer1": "conv_b3_wgrad_kernel<W3Geo<3, 84",
              "conv_wgrad_layer2": "conv_b3_wgrad_kernel<W3Geo<32, 20", "conv_wgrad_layer3": "conv_b3_wgrad_kernel<W3Geo<64, 9"}


def pmc_traffic(kernel):
    """HBM-side bytes per launch of ``kernel`` at the training shape from the committed rocprofv3 PMC passes
    (the newest profiles/rNN_pmc_summary.json that holds the kernel; produced by tools/run_pmc.sh in separate --pmc passes):
    FETCH_SIZE KiB x 2 (gfx950 reports half of a wide coalesced read stream, MI355X_MICROARCH.md section HBM) + WRITE_SIZE KiB.
    None if no profile is present."""
    symbol = PMC_SYMBOL.get(kernel, kernel)
    try:
        doc = path = None
        # newest round that has the kernel; the group step kernel has its own passes at config 5's shape (that file also holds the
        # per-worker kernel at config 5's shape -- not the config-3 instantiation the other lines are about)
        own = (os.path.join("r05", "pmc_rollout_step_config5_fold.json"), os.path.join("r05", "pmc_rollout_step_config5.json")) \
            if kernel == "rollout_group_kernel" else ()         # (_fold: after fc_out was folded into the gate products; the other: before)
        for cand in own + ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json"):
            path = os.path.join(REPO, "profiles", cand)
            if not os.path.exists(path):
                continue
            with open(path) as f:
                d = json.load(f)
            if symbol in d or any(isinstance(v, dict) and any(n.startswith(symbol) for n in v) for v in d.values()):
                doc = d
                break
        if doc is None:
            return None
        target = ""
        if symbol not in doc:         # round-2 layout: {target: {kernel symbol incl. template arguments: counters}}
            # the timed region hands the window pass SORTED minibatches: that target if it was collected
            target = (("window_sorted" if "window_sorted" in doc else "window_train") if symbol == "window_pass_kernel"
                      else next(t for t in doc if any(n.startswith(symbol) for n in doc[t])))
            doc = doc[target]
            symbol = next(n for n in doc if n.startswith(symbol))
        k = doc[symbol]
        return {"bytes_per_launch": 2 * 1024 * k["FETCH_SIZE"]["mean"] + 1024 * k["WRITE_SIZE"]["mean"],
                "source": f"profiles/{os.path.relpath(path, os.path.join(REPO, 'profiles'))}, {('target ' + target + ', ') if target else ''}kernel symbol {symbol} "
                          f"(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at the {'config-5' if own else 'config-3'} shape"
                          + ("; forward and backward launches of the window pass averaged)" if symbol.startswith("window") else ")"),
                "mfma_busy_fraction_pmc": k.get("derived", {}).get("mfma_busy_fraction")}
    except Exception:
        return None


def cpu_baseline(cfg, seed=0):
    """Oracle trainer on host cores, bounded sample: 32 workers x 64 steps, 5 epochs x 1 minibatch of 2048."""
    from environments.vec_env import make_vec_env
    from oracle.ref_algo import OracleTrainer
    cores = len(os.sched_getaffinity(0)) or 1
    threads = min(cores, 64)
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        c = json.loads(json.dumps(cfg))
        c["worker_steps"] = 64
        c["n_mini_batch"] = 1
        env = make_vec_env(c["environment"], c["n_workers"])
        torch.manual_seed(seed)
        tr = OracleTrainer(c, env, seed=seed)
        # the rollout is 64 x (a few dozen tiny ops on 32 samples): more threads only add hand-over cost there, the optimisation
        # phase (convolutions on 2048 images) uses all of them
        roll_threads = min(threads, 8)
        sample = tr.sample

        def sample_few_threads(forced_actions=None):
            torch.set_num_threads(roll_threads)
            try:
                return sample(forced_actions)
            finally:
                torch.set_num_threads(threads)

        tr.sample = sample_few_threads
        t0 = time.perf_counter()
        _, _, split = tr.update(0)
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(prev)
    steps = c["n_workers"] * c["worker_steps"]
    quota = None
    try:      # the CPU time the container may really use (cgroup v2): "max" or "<quota us> <period us>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    return {"value": steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port", "cgroup_cpu_quota": quota,
            "sample_short": f"oracle trainer, 1 update of {c['n_workers']}x{c['worker_steps']} env steps, {c['epochs']} epochs x 1 minibatch of {steps}; {dt:.1f} s "
                            f"on {threads} torch threads of {cores} host cores",
            "sample": f"1 update of {c['n_workers']} workers x {c['worker_steps']} steps (= {steps} env steps) with {c['epochs']} epochs x 1 "
                      f"minibatch of {steps} samples: same per-env-step work as the full config (minibatch 2048, 5 epochs); "
                      f"{dt:.1f} s (rollout {split['rollout_s']:.1f} s on {roll_threads} torch threads, train {split['train_s']:.1f} s on "
                      f"{threads}) of {cores} host cores"}

LINE_LIMIT = 8000      # bytes of the ONE stdout line (the driver's parser dropped round 5's 20,013-character line); the full record goes to --full-json


def _num(x, sig=6):
    """Floats to ``sig`` significant digits (the line is a summary: the full-precision record is the side file)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _num(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, sig) for v in x]
    return float(x) if hasattr(x, "__float__") else str(x)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_unit", "avg_launch_ms", "launches", "bytes_per_launch",
                 "flops_per_launch", "unique_bytes_per_launch", "frac_unique", "hbm_side_frac", "mfma_busy_fraction_pmc", "est_region_ms",
                 "us_per_dependent_phase", "dtype", "pipe_achieved", "pipe_peak", "pipe_frac")


def _flat_rooflines(r, prefix="", out=None):
    """Micro-benchmark tree -> {path: [avg_launch_ms, frac]} (one entry per kernel line)."""
    out = {} if out is None else out
    if not isinstance(r, dict):
        return out
    if "frac" in r and ("avg_launch_ms" in r or "ms" in r):
        out[prefix.rstrip(".")] = [r.get("avg_launch_ms", r.get("ms")), r["frac"]] + ([r["pipe_frac"]] if "pipe_frac" in r else [])      # (third: of the bf16 pipe's peak, kernels of csrc/conv_b3*.hip)
    for k, v in r.items():
        if isinstance(v, dict) and k not in ("shape", "model"):
            _flat_rooflines(v, prefix + k + ".", out)
    return out


def worker_processes_run(updates=3, warmup=1):
    """The same update with the environments in worker PROCESSES over shared memory + the native rollout driver (the reference's
    workers step concurrently, worker.py:36-48), same fresh-draw environments.  Run as a fresh process of this script: the worker
    groups' streams of a second trainer in THIS process would share hardware queues with the first one's (torch hands out pooled
    streams; measured: rollout 0.161 s instead of 0.091 s per update)."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(updates), "--warmup", str(warmup), "--worker-processes", "on",
               "--config", CONFIG_NAME, "--no-rooflines", "--no-cpu-baseline", "--no-worker-processes-run", "--no-profile",
               "--full-json", os.path.join(tmp, "wp.json")]
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            raise RuntimeError(f"worker-process run failed (rc {r.returncode}): {r.stderr[-400:]}")
        rec = json.loads(lines[-1])
    return {"value": rec["value"], "unit": rec["unit"], "updates": updates, "warmup": warmup, "phase_s_per_step": rec.get("phase_s_per_step"),
            "envs_per_process": rec["config"].get("envs_per_process"), "env": "pool 0 (fresh U[0,1) draw per observation), worker processes, own process"}


def compact_line(full, full_path):
    """The driver's line: the contract's keys, the roofline / cpu_baseline objects with numbers and one short note each, and
    ONE {kernel: [avg_ms, frac]} summary of the micro-benchmarks.  Everything else (per-kernel tables, models, prose) is in
    the side file ``full_path``."""
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "dtype_note", "data") if k in full}
    cfgf = full["config"]
    line["config"] = dict(_pick(cfgf, ("env_pool", "observation_rows", "minibatch", "parallelism", "attention", "encoder_products", "dp_collective", "dp_step", "rollout_groups", "worker_processes",
                                       "envs_per_process", "cgroup_cpu_quota")), workload=cfgf["workload"][:300])
    for k in ("value_worker_processes", "phase_s_per_step", "speedup_vs_cpu_baseline"):
        if full.get(k) is not None:
            line[k] = full[k]
    for k in ("roofline", "roofline_train"):
        r = full.get(k)
        if r:
            line[k] = _pick(r, ROOFLINE_KEYS)
            line[k].setdefault("traffic", None)
            if r.get("note_short"):
                line[k]["note"] = r["note_short"][:120]
    if full.get("allreduce"):
        line["allreduce"] = _pick(full["allreduce"], ("bytes", "avg_ms", "bus_gbs", "exposed_ms", "per_update", "overlap", "error"))
    if isinstance(full.get("rooflines"), dict):
        line["rooflines"] = {"error": full["rooflines"]["error"][:200]} if "error" in full["rooflines"] else _flat_rooflines(full["rooflines"])
    if full.get("cpu_baseline"):
        c = full["cpu_baseline"]
        line["cpu_baseline"] = dict(_pick(c, ("value", "unit", "cores", "kind", "cgroup_cpu_quota")), sample=c.get("sample_short", c.get("sample", ""))[:200])
    line["full_json"] = full_path
    line = _num(line)
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) > LINE_LIMIT:         # never lose the headline: shed the optional summaries, largest first
        for k in ("rooflines", "roofline_train", "allreduce"):
            line.pop(k, None)
            text = json.dumps(line, allow_nan=False, separators=(",", ":"))
            if len(text) <= LINE_LIMIT:
                break
    return text


def main():
    # the contract is ONE JSON line on stdout: whatever the libraries / the trainer print on the way goes to stderr
    # (at the descriptor level: RCCL / gloo / the HIP runtime print from C)
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--no-rooflines", action="store_true", help="skip the kernel micro-benchmarks after the timed region")
    ap.add_argument("--dist-backend", default=None, help="testing only: torch.distributed backend instead of nccl (= RCCL), e.g. gloo")
    ap.add_argument("--dp-collective", choices=("torch", "etm"), default=None,
                    help="gradient all-reduce through the library's own RCCL communicator (etm, the default: created and self-tested "
                         "at start-up, torch.distributed takes over if that fails) or through torch.distributed (RCCL backend)")
    ap.add_argument("--all-ranks-on-device", type=int, default=None,
                    help="testing only: every rank uses this device index (exercises the multi-process path on a 1-GPU box)")
    ap.add_argument("--plumbing-check", action="store_true",
                    help="testing only, no GPU needed: run the multi-process skeleton of this script (rank / world parsing, process "
                         "group, sharding, barrier, flat-bucket all-reduce, max over ranks, rank-0 JSON line) on CPU tensors over gloo")
    ap.add_argument("--numa-pin", choices=("auto", "on", "off"), default="auto",
                    help="restrict the rank (trainer thread, observation-copier helpers, torch pool) to the CPUs of its GPU's NUMA node "
                         "before any pinned buffer is allocated; auto = on for multi-GPU runs (8 ranks share the host), off for one GPU")
    ap.add_argument("--attention", choices=("folded", "dense"), default="folded",
                    help="kernel family of the window attention (etm.ops.set_attention_impl); folded is the product default")
    ap.add_argument("--dp-overlap", choices=("on", "off"), default="off",
                    help="data-parallel runs: all-reduce of the head / transformer / lin_hidden gradient slice on a side stream under the "
                         "encoder's backward pass (bit-identical parameters; default off until a multi-GPU run has A/B-ed it)")
    ap.add_argument("--dp-graph-collective", choices=("on", "off"), default="on",
                    help="data-parallel runs with the library collective: capture the gradient all-reduce INSIDE the optimisation step's graph "
                         "(one replay per minibatch; self-tested on every rank at start-up, else the three-call step) or keep it a host call")
    ap.add_argument("--worker-processes", choices=("config", "on", "off"), default="config",
                    help="environments in worker processes over shared memory + the native rollout driver (config: what the YAML says)")
    ap.add_argument("--envs-per-process", type=int, default=None, help="environments per worker process (with worker processes)")
    ap.add_argument("--rollout-groups", default=None, help="override rollout_groups (auto, 1, 2, 4, 8)")
    ap.add_argument("--config", default="synthetic_minigrid",
                    help="YAML of configs/ to time (default: BASELINE config 3 = the driver's line; synthetic_mortar_gtrxl / synthetic_cartpole: the "
                         "shapes of BASELINE configs 5 / 2 -- gated pre-LN blocks, the group form of the rollout step kernel)")
    ap.add_argument("--env-pool", type=int, default=None,
                    help="default: the YAML's, 0 = SURVEY 8d to the letter: every observation is a fresh default_rng(seed + worker).random([3, 84, 84]) "
                         "draw inside the timed region; n > 0: every worker replays a ring of n frames drawn once (rounds 1 - 5 timed 64)")
    ap.add_argument("--full-json", default=os.path.join(REPO, "bench_full.json"),
                    help="side file for the FULL record (per-kernel tables, roofline models, micro-benchmark trees, prose): the stdout line "
                         "is a < 8 KB summary of it")
    ap.add_argument("--no-worker-processes-run", "--no-fresh-obs", dest="no_worker_processes_run", action="store_true",
                    help="skip the second measurement after the timed region (N == 1 only): the same update with the environments in worker "
                         "processes over shared memory (the reference's form), reported as value_worker_processes")
    ap.add_argument("--direct-rows", choices=("on", "off"), default="on",
                    help="in-process environments write the observation rows of a step straight into the staging array in device memory "
                         "(large BAR; round 6) instead of pinned memory + a copy-engine transfer per worker group and step")
    ap.add_argument("--gen-threads", type=int, default=None,
                    help="pool 0, in-process environments: host threads that draw a step's observation rows (libetm_envgen.so's pool)")
    args = ap.parse_args()
    global CONFIG_NAME
    CONFIG_NAME = args.config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs one process per GPU: launch with python -m torch.distributed.run "
                             f"--nproc-per-node {args.gpus} (WORLD_SIZE is {world})")
    if args.plumbing_check:
        from etm.dist import DataParallel
        dp = DataParallel(torch.device("cpu"), backend="gloo", collective="torch") if world > 1 else None
        first, count = (dp.shard(32 * world) if dp is not None else (0, 32))
        assert (first, count) == (rank * 32, 32)
        flat = torch.full((1000,), float(rank + 1))
        if dp is not None:
            dp.flat = flat
            dp.barrier()
        t0 = time.perf_counter()
        if dp is not None:
            dp.all_reduce_grads()
            dp.barrier()
        elapsed = time.perf_counter() - t0
        if dp is not None:
            elapsed = dp.max_over_ranks(elapsed)
        want = sum(range(1, world + 1)) / world
        assert abs(float(flat[0]) - want) < 1e-6, (float(flat[0]), want)
        if rank == 0:
            print(json.dumps({"metric": "plumbing-check", "value": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "first_worker_of_last_rank": (world - 1) * 32, "allreduce_mean": float(flat[0]), "elapsed_s": elapsed}),
                  file=json_out, flush=True)
        if dp is not None:
            dp.barrier()
            dp.close()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py measures the MI355X path; no HIP device is visible (there is no CPU fallback)")
    device = torch.device("cuda", local_rank if args.all_ranks_on_device is None else args.all_ranks_on_device)
    torch.cuda.set_device(device)
    numa_cpus = None
    if args.numa_pin == "on" or (args.numa_pin == "auto" and world > 1):
        from etm.dist import pin_to_gpu_numa_node
        numa_cpus = pin_to_gpu_numa_node(device.index)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)) // max(world, 1))))  # host side is a single python loop per rank

    from etm import lib as etm_lib
    from etm.dist import DataParallel
    from trainer import PPOTrainer
    lib = etm_lib.load()
    from etm import ops as etm_ops
    etm_ops.set_attention_impl(args.attention)

    cfg = load_config()
    cfg["dp_overlap"] = args.dp_overlap == "on"
    cfg["direct_observation_rows"] = args.direct_rows == "on"
    cfg["dp_graph_collective"] = args.dp_graph_collective == "on"
    if args.worker_processes != "config":
        cfg["worker_processes"] = args.worker_processes == "on"
    if args.envs_per_process is not None:
        cfg["envs_per_process"] = args.envs_per_process
    if args.env_pool is not None:
        cfg["environment"]["pool"] = args.env_pool
    if args.gen_threads is not None:
        cfg["environment"]["gen_threads"] = args.gen_threads
    if args.rollout_groups is not None:
        cfg["rollout_groups"] = args.rollout_groups if args.rollout_groups == "auto" else int(args.rollout_groups)
        cfg["rollout_min_group_size"] = min(int(cfg.get("rollout_min_group_size", 8)), 4)
    dp = DataParallel(device, backend=args.dist_backend, collective=args.dp_collective) if world > 1 else None
    torch.manual_seed(0)
    np.random.seed(0)
    trainer = PPOTrainer(cfg, run_id="bench", device=device, dp=dp, first_worker_id=rank * cfg["n_workers"], tensorboard=False)

    def one_update(i):
        lr, beta, clip = trainer.schedules(i)
        t0 = time.perf_counter()
        lib.etm_profile_set_tag(0)
        trainer._sample_training_data()
        trainer.buffer.prepare_batch_dict()
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        lib.etm_profile_set_tag(1)
        trainer._train_epochs(lr, clip, beta)
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1

    for i in range(args.warmup):
        one_update(i)
    if not args.no_profile:
        etm_lib.profile_collect()
        lib.etm_profile_enable(1)
        # the optimisation step is a captured graph (no per-kernel events inside a replay): ONE minibatch of every update of the timed
        # region runs the identical step eagerly so that the dominant kernel is still timed live, inside the timed region (rounds 2 - 5
        # sampled every 8th minibatch: an eager step costs ~0.9 ms more than a replay at config 3, five of them 3 % of the update --
        # and 16 % at the launch-bound config 2; the per-kernel averages then come from `steps` launches instead of 5 x `steps`)
        if getattr(trainer, "_use_train_graph", False):
            trainer.profile_sample_every = max(8, cfg["epochs"] * cfg["n_mini_batch"])
    if dp is not None:
        dp.barrier()
    torch.cuda.synchronize(device)
    t_start = time.perf_counter()
    phase = np.zeros(2)
    env_s = 0.0
    for i in range(args.steps):
        r, tr_s = one_update(args.warmup + i)
        phase += (r, tr_s)
        env_s += trainer.last_update_timing.get("env_s", 0.0)
    torch.cuda.synchronize(device)
    if dp is not None:
        dp.barrier()
    elapsed = time.perf_counter() - t_start
    lib.etm_profile_enable(0)
    prof = etm_lib.profile_collect() if not args.no_profile else {}
    if dp is not None:
        elapsed = dp.max_over_ranks(elapsed)

    # gradient all-reduce on its own (message = 4 * params bytes, bus bandwidth against the xGMI link rate);
    # measured AFTER the timed region, on every rank, with the trainer's own flat bucket and collective call
    allreduce = None
    if dp is not None:
        try:
            bucket = trainer.flat_grads
            for _ in range(3):
                dp.all_reduce_grads(average=False)
            torch.cuda.synchronize(device)
            dp.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n_ar = 20
            e0.record()
            for _ in range(n_ar):
                dp.all_reduce_grads(average=False)
            e1.record()
            torch.cuda.synchronize(device)
            ar_ms = dp.max_over_ranks(e0.elapsed_time(e1) / n_ar)
            nbytes = bucket.numel() * 4
            allreduce = {"bytes": nbytes, "avg_ms": ar_ms, "includes": "sum all-reduce of the flat fp32 gradient arena (the division by the world size rides in the optimiser kernel)",
                         "bus_gbs": nbytes * 2 * (world - 1) / world / (ar_ms * 1e-3) / 1e9, "xgmi_link_peak_gbs": 153.0,
                         "per_update": cfg["epochs"] * cfg["n_mini_batch"], "overlap": bool(cfg.get("dp_overlap", False))}
            # what the optimisation step really waits for the collective: without overlap all of it; with --dp-overlap on the main
            # stream's wait for the side stream after the encoder's backward pass, measured on one extra (eager) minibatch step
            exposed = ar_ms
            if cfg.get("dp_overlap", False) and getattr(trainer, "_train_graph_a2", None) is not None:
                trainer._ar_probe = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                trainer._train_graph[0].replay()
                trainer._allreduce_rest_async()
                trainer._train_graph_a2.replay()
                trainer._allreduce_conv_and_join()
                torch.cuda.synchronize(device)
                exposed = dp.max_over_ranks(trainer._ar_probe[0].elapsed_time(trainer._ar_probe[1]))
                trainer._ar_probe = None
            allreduce["exposed_ms"] = exposed
        except Exception as exc:       # reporting only: never lose the throughput line over it
            allreduce = {"error": repr(exc)}

    W, S = cfg["n_workers"], cfg["worker_steps"]
    t = cfg["transformer"]
    L, D, H, nb = t["memory_length"], t["embed_dim"], t["num_heads"], t["num_blocks"]
    N = W * S // cfg["n_mini_batch"]
    total_env_steps = world * W * S * args.steps

    if rank == 0:
        # ---- roofline of the dominant kernel (training-tag launches, minibatch shape)
        kernels = {}
        for (tag, name), (ms, cnt) in prof.items():
            if tag != 1:
                continue
            entry = {"launches": cnt, "avg_ms": ms / cnt, "total_ms": ms}
            work = kernel_work(name, N, L, D, H)
            if work:
                entry["bound"] = work[0]
                if work[0] == "mfma":
                    entry["tflops"] = work[1] / (ms / cnt * 1e-3) / 1e12
                else:
                    entry["gbs"] = work[1] / (ms / cnt * 1e-3) / 1e9
            kernels[name] = entry
        # ---- the kernels of a rollout step, timed eagerly AFTER the timed region (inside it they run from captured graphs, which
        # take no per-kernel events): average launch time x the launches the timed region made = their share of GPU time
        rollout_step = None
        if not args.no_profile and trainer is not None:
            try:
                import kernel_rooflines
                rollout_step = kernel_rooflines.rollout_step(trainer)
            except Exception as exc:       # reporting only: never lose the throughput line over it
                rollout_step = {"error": repr(exc)}
        # estimated GPU time of every modelled kernel over the whole timed region: the optimisation phase is sampled (every
        # profile_sample_every-th minibatch runs eagerly with events), the rollout step kernels are launched S x groups times per update
        n_mb = cfg["epochs"] * cfg["n_mini_batch"] * args.steps
        sampled = max(1, n_mb // max(1, getattr(trainer, "profile_sample_every", 1) or 1))
        for k in kernels.values():
            k["est_region_ms"] = k["total_ms"] * n_mb / sampled
        est = {k: v["est_region_ms"] for k, v in kernels.items() if "bound" in v}
        if rollout_step and "rollout_trxl_kernel" in rollout_step:
            est["rollout_trxl_kernel"] = rollout_step["rollout_trxl_kernel"]["avg_launch_ms"] * S * len(trainer._groups) * args.steps
        roofline = None
        cand = [k for k in kernels if "bound" in kernels[k]]
        dom_all = max(est, key=est.get) if est else None
        if dom_all == "rollout_trxl_kernel":
            # the kernel with the largest share of GPU time is the rollout step kernel: a dependent chain of matrix-vector products
            # (latency-bound); its rate is reported as streamed bytes / time against the HBM peak, with what that stream really is
            rs = rollout_step["rollout_trxl_kernel"]
            roofline = {"kernel": "rollout_trxl_kernel", "bound": "hbm", "achieved": rs["achieved"], "peak": rs["peak"], "unit": "GB/s",
                        "frac": rs["frac"], "traffic": None, "traffic_unit": "bytes per launch", "avg_launch_ms": rs["avg_launch_ms"],
                        "launches": rs["launches"], "bytes_per_launch": rs["bytes_per_launch"], "dtype": "f32",
                        "unique_bytes_per_launch": rs.get("unique_bytes_per_launch"), "frac_unique": rs.get("frac_unique"),
                        "est_region_ms": est[dom_all], "us_per_dependent_phase": rs["us_per_dependent_phase"], "model": rs["model"],
                        "note_short": "rollout step kernel: latency chain, bytes = streamed (L2-served) weights + K|V; frac_unique = once-only bytes",
                        "note": "dominant kernel of ALL GPU time (rollout: one launch per worker group and step).  It is a dependency "
                                "chain (products -> exchange -> LayerNorm ...), bound by round-trip latency, not by a roofline: "
                                "bytes_per_launch = workers x every matrix of the chain (each team streams them once for ONE worker) + the "
                                "K | V window columns, served by L2 / the Infinity Cache (unique bytes: model.unique_weight_bytes); "
                                "timed eagerly after the timed region (graph replays take no per-kernel events).  The dominant kernel of "
                                "the optimisation phase is in roofline_train."}
            grouped = bool(getattr(trainer._groups[0], "group_kernel", False))
            if grouped:
                roofline["kernel"] = "rollout_group_kernel"
                roofline["note"] = ("dominant kernel of ALL GPU time: the GROUP form of the rollout step kernel (csrc/rollout_group.hip, gated layouts): the "
                                    "workers of a group are the rows of every product, 32 workgroups own its columns, every matrix is read once per group "
                                    "and step; bytes_per_launch = every matrix of the chain once + per worker the tail's K | V projection and the K | V "
                                    "window columns.  A dependency chain (82 phases at config 5: products -> all-gather of tagged packets -> ...), bound by "
                                    "the exchange latency, not by a roofline; timed eagerly after the timed region.")
            # (PMC passes exist for the config-3 step kernel and for the group kernel at config 5's shape only)
            tr_pmc = (pmc_traffic("rollout_group_kernel") if CONFIG_NAME == "synthetic_mortar_gtrxl" else None) if grouped else \
                (pmc_traffic("rollout_trxl_kernel") if CONFIG_NAME == "synthetic_minigrid" else None)
            if tr_pmc:
                roofline["traffic"], roofline["traffic_source"] = tr_pmc["bytes_per_launch"], tr_pmc["source"]
                roofline["hbm_side_frac"] = tr_pmc["bytes_per_launch"] / (rs["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        roofline_train = None
        if cand:
            dom = max(cand, key=lambda k: kernels[k]["est_region_ms"])
            kind, work = kernel_work(dom, N, L, D, H)
            if kind == "mfma":
                ach, peak, unit, extra = kernels[dom]["tflops"], FP32_MFMA_PEAK_TFLOPS, "TFLOP/s", {"flops_per_launch": work, "dtype": "f32 (v_mfma_f32_32x32x2_f32)"}
                if dom.startswith("conv_") and trainer.model.encoder_products == "bf16x3":      # its products run on the bf16 pipe: both views
                    import kernel_rooflines
                    extra.update(kernel_rooflines.b3_fields(work, kernels[dom]["avg_ms"]))
            else:
                ach, peak, unit, extra = kernels[dom]["gbs"], HBM_PEAK_GBS, "GB/s", {"bytes_per_launch": work, "dtype": "f32"}
            tr_pmc = pmc_traffic(dom)
            roofline_train = {"kernel": dom, "bound": kind, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                              "traffic": tr_pmc["bytes_per_launch"] if tr_pmc else None,       # HBM-side bytes per launch (PMC)
                              "traffic_unit": "bytes per launch", "traffic_source": tr_pmc["source"] if tr_pmc else None,
                              "avg_launch_ms": kernels[dom]["avg_ms"], "launches": kernels[dom]["launches"],
                              "est_region_ms": kernels[dom]["est_region_ms"], "shape": {"N": N, "L": L, "D": D, "H": H}}
            if tr_pmc:
                # what crosses the fabric per launch / time / peak: next to the algorithmic `frac`
                roofline_train["hbm_side_frac"] = tr_pmc["bytes_per_launch"] / (kernels[dom]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            if tr_pmc and kind == "mfma":
                roofline_train["mfma_busy_fraction_pmc"] = tr_pmc.get("mfma_busy_fraction_pmc")
            roofline_train.update(extra)
            roofline_train["note_short"] = ("dominant kernel of the optimisation phase, HIP events inside the timed region (1 eager minibatch per update)")
            if kind == "hbm":
                roofline_train["extra_bytes_per_launch"] = N * 4.0 * (2 * H * D + H * L * (2 if dom.endswith("bwd_kernel") else 1))
                roofline_train["note"] = ("bytes_per_launch = N*L*D*4, the un-deduplicated window read of SURVEY 8d; minibatches are sorted by "
                                    "(worker, step) and every XCD takes a contiguous chunk of the samples, so most of that stream is "
                                    "served by the XCD's L2 (unique window rows per block <= 61 MB; `traffic` = the PMC passes of "
                                    "the sorted pattern: about a third of bytes_per_launch crosses the fabric): see "
                                    "rooflines.window.cold_hbm for the same kernel with every byte coming from HBM")
        rollout_k = {name: {"launches": cnt, "avg_ms": ms / cnt} for (tag, name), (ms, cnt) in prof.items() if tag == 0}
        out = {
            # (BASELINE.json's metric is quoted on configs (3)/(4): the default line; --config lines name their own shape)
            "metric": ("env-steps/sec (whole node) MinigridMemory 3x84x84" if CONFIG_NAME == "synthetic_minigrid" else
                       "env-steps/sec (whole node), shape of BASELINE config (5) MortarMayhem-Grid GTrXL 3x84x84" if CONFIG_NAME == "synthetic_mortar_gtrxl" else
                       "env-steps/sec (whole node), shape of BASELINE config (2) CartPole-masked-velocity GTrXL" if CONFIG_NAME == "synthetic_cartpole" else
                       f"env-steps/sec (whole node), configs/{CONFIG_NAME}.yaml"),
            "value": total_env_steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": ("f32 tensors and results throughout; the encoder's products of the optimisation phase are taken as six bf16 MFMA products of "
                           "exactly split f32 operands (error vs float64 below the f32 MFMA kernels'; config.encoder_products)"
                           if getattr(trainer.model, "encoder_products", "fp32") == "bf16x3" else "f32 throughout"),
            "data": "synthetic",
            "config": {"workload": (("BASELINE config (3)/(4)" if CONFIG_NAME == "synthetic_minigrid" else
                                     "shape of BASELINE config (5)" if CONFIG_NAME == "synthetic_mortar_gtrxl" else
                                     "shape of BASELINE config (2)" if CONFIG_NAME == "synthetic_cartpole" else "config")
                                    + f" configs/{CONFIG_NAME}.yaml: per GPU {W} workers x {S} steps, {cfg['epochs']} epochs x {cfg['n_mini_batch']} minibatches of {N}, "
                                    f"{'GTrXL' if cfg['transformer'].get('gtrxl') else 'TrXL'} {cfg['transformer'].get('layer_norm') or 'no'}-LN "
                                    f"{cfg['transformer']['num_blocks']} blocks D={cfg['transformer']['embed_dim']} H={cfg['transformer']['num_heads']} "
                                    f"L={cfg['transformer']['memory_length']}, synthetic {'x'.join(str(x) for x in cfg['environment']['obs_shape'])} obs: "
                                    + ("fresh default_rng draw per observation (SURVEY 8d to the letter)" if cfg["environment"].get("pool", 64) == 0 else
                                       f"{cfg['environment'].get('pool', 64)}-frame ring per worker drawn once from U[0,1)")
                                    + (", envs in worker processes" if cfg.get("worker_processes", False) else ", in-process envs")
                                    + " inside the timed region, random-init weights"),
                       "workload_detail": "rewards Bernoulli(0.05), done at 96 steps or Bernoulli(0.02); observations = numpy default_rng(seed + worker id)"
                                          ".random([3, 84, 84], float32), drawn by libetm_envgen.so (the same stream bit for bit); value_worker_processes: "
                                          "the same environments stepped in worker processes over shared memory",
                       "env_pool": cfg["environment"].get("pool", 64), "env_gen_threads": cfg["environment"].get("gen_threads", 1),
                       "host_threads_busy": trainer._host_plan["busy_threads"], "host_cpu_plan": trainer._host_plan["reason"],
                       "host_cpus_per_rank": trainer._host_plan["budget"]["per_rank"], "cgroup_cpu_quota": trainer._host_plan["budget"]["cgroup_quota"],
                       "copy_threads": trainer._host_plan["copy_threads"],
                       "env_steps_per_update_per_gpu": W * S, "minibatch": N, "parallelism": f"dp{world}",
                       "attention": args.attention, "dp_collective": dp.collective if dp is not None else None,
                       "dp_step": (None if dp is None else "one_graph" if getattr(trainer, "_dp_one_graph", False) else "graph_a+allreduce+graph_b"),
                       "numa_pinned_cpus": len(numa_cpus) if numa_cpus else None,
                       "rollout_groups": len(getattr(trainer, "_groups", None) or []) or 1, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "encoder_products": trainer.model.encoder_products + (" (f32 products as 6 bf16 MFMA products of exactly split operands; error vs float64 below the fp32 MFMA kernels')" if trainer.model.encoder_products == "bf16x3" else ""),
                       "observation_rows": ("direct (host writes into device memory" + (", HDP flush register written)" if etm_ops._direct_mode.get(device.index) == 2 else ")")
                                            if getattr(trainer, "_direct_rows", False) else "pinned + upload"),
                       "worker_processes": bool(cfg.get("worker_processes", False)), "envs_per_process": trainer._host_plan["envs_per_process"] if cfg.get("worker_processes", False) else None,
                       "native_rollout_driver": bool(getattr(trainer, "_native_rollout", False))},
            "phase_s_per_step": {"rollout": phase[0] / args.steps, "train": phase[1] / args.steps, "env_host": env_s / args.steps},
            "roofline": roofline if roofline is not None else roofline_train,
            "roofline_train": roofline_train,
            "rollout_step": rollout_step,
            "allreduce": allreduce,
            "kernels_train": kernels,
            "kernels_rollout": rollout_k,
        }
        if not args.no_rooflines:
            try:
                import kernel_rooflines
                trainer.close()
                del trainer
                torch.cuda.empty_cache()
                out["rooflines"] = kernel_rooflines.all_rooflines(device)
                out["rooflines"]["rollout_step"] = rollout_step
            except Exception as exc:       # reporting only: never lose the throughput line over it
                out["rooflines"] = {"error": repr(exc)}
            trainer = None
        if world == 1 and not args.no_worker_processes_run and CONFIG_NAME == "synthetic_minigrid" and not cfg.get("worker_processes", False):
            # the other environment form, after the timed region: the same fresh-draw environments stepped in worker processes
            try:
                if trainer is not None:
                    trainer.close()
                    trainer = None
                    torch.cuda.empty_cache()
                out["worker_processes_run"] = worker_processes_run()
                out["value_worker_processes"] = out["worker_processes_run"]["value"]
            except Exception as exc:       # reporting only: never lose the throughput line over it
                out["worker_processes_run"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        full_path = args.full_json
        try:
            os.makedirs(os.path.dirname(os.path.abspath(full_path)), exist_ok=True)
            with open(full_path, "w") as f:
                json.dump(out, f, indent=1, default=float)
            full_path = os.path.relpath(full_path, REPO)
        except OSError as exc:
            full_path = f"not written: {exc!r}"[:100]
        print(compact_line(out, full_path), file=json_out, flush=True)
    if dp is not None:
        dp.barrier()            # rank 0 ran the kernel micro-benchmarks: leave together
    if trainer is not None:
        trainer.close()
    if dp is not None:
        dp.close()


if __name__ == "__main__":
    main()
