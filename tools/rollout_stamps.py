"""Phase timeline of the fused rollout-step kernel (workgroup 0), from the diagnostic build `make -C .../csrc diag-stamps`.

    ETM_DIAG_LIB=$PWD/tools/scratch/libetm_hip_stamps.so python tools/rollout_stamps.py
"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "episodic-transformer-memory-ppo_amd"))
import torch
from etm import lib as etm_lib
from yaml_parser import YamlParser
from trainer import PPOTrainer

cfg = YamlParser(os.path.join(REPO, "episodic-transformer-memory-ppo_amd", "configs", "synthetic_minigrid.yaml")).get_config()
tr = PPOTrainer(cfg, run_id="stamps", device=torch.device("cuda", 0), tensorboard=False)
tr._sample_training_data()
tr._sample_training_data()
torch.cuda.synchronize()
out = (ctypes.c_longlong * 64)()
assert etm_lib.load().etm_diag_rollout_trxl_stamps(out) == 0
nb = cfg["transformer"]["num_blocks"]
names = {0: "start", 1: "embedding product", 2: "E0 exchange"}
for b in range(nb):
    for k, n in enumerate(["q product", "energies", "softmax + context", "fc_out product", "Ea exchange + sum", "LayerNorm1",
                           "fc product", "Eb exchange + LayerNorm2"]):
        names[3 + 8 * b + k] = f"block {b}: {n}"
names.update({40: "hidden heads + output dots", 41: "Ez exchange", 42: "sampling"})
prev = out[0]
for k in sorted(names):
    if out[k]:
        print(f"{names[k]:40s} +{(out[k] - prev) / 100.0:7.2f} us   (t = {(out[k] - out[0]) / 100.0:7.2f})")
        prev = out[k]
