"""Command line entry, same flags as upstream train.py (--config, --run-id, --cpu).

    python train.py --config ./configs/synthetic_minigrid.yaml --run-id demo

One process drives one MI355X.  For data-parallel training over the GPUs of a node launch one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py --config ...

(``n_workers`` in the YAML is the number of environments PER PROCESS; gradients are all-reduced with RCCL -- through
the library's own communicator by default, through torch.distributed with ETM_DP_COLLECTIVE=torch.)
"""
import argparse
import os

# the host driver shares device memory between the ranks of a node through dmabuf handles only (RCCL, multi-process runs); the HIP
# runtime reads this when it starts, i.e. at the first device call below -- so it is set here, before anything touches the device
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# eight hardware queues: four rollout worker groups (two streams each) run concurrently (trainer.py, rollout_groups: auto);
# read by the HIP runtime when it starts, like the variable above
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("ETM_HW_QUEUES_SET_EARLY", "1")   # marker for trainer.py: the line above ran before the HIP runtime started

import torch  # noqa: E402

from trainer import PPOTrainer  # noqa: E402
from yaml_parser import YamlParser  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description="PPO + TransformerXL episodic memory on MI355X")
    ap.add_argument("--config", default="./configs/poc_memory_env.yaml", help="Path to the yaml config file")
    ap.add_argument("--run-id", default="run", help="Tag for the tensorboard summary and the saved model")
    ap.add_argument("--cpu", action="store_true", help="(upstream flag) not available in this build: raises")
    args = ap.parse_args()
    config = YamlParser(args.config).get_config()
    if args.cpu:
        raise SystemExit("--cpu: this build is the MI355X-native path and has no CPU trainer "
                         "(the CPU restatement used for parity lives under oracle/)")
    if not torch.cuda.is_available():
        raise SystemExit("no HIP device visible: the MI355X path cannot run (there is no CPU fallback)")
    dp = None
    first_worker = 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from etm.dist import DataParallel, pin_to_gpu_numa_node
        pin_to_gpu_numa_node(local_rank)            # before the pinned staging buffers are allocated; no-op if the platform does not say
        dp = DataParallel(device)
        first_worker = dp.rank * config["n_workers"]
    trainer = PPOTrainer(config, run_id=args.run_id, device=device, dp=dp, first_worker_id=first_worker,
                         tensorboard=(dp is None or dp.rank == 0))
    trainer.run_training()
    trainer.close()
    if dp is not None:
        dp.close()


if __name__ == "__main__":
    main()
