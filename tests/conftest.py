import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "episodic-transformer-memory-ppo_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, PKG, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)

# Per-shape GEMM tuning (trainer.py, `tunable_gemm`) is a speed feature of long runs; the parity tests build dozens of small
# trainers and would spend most of their time tuning shapes they use once.  (bench.py / train.py keep it on.)
os.environ.setdefault("ETM_TUNABLE_GEMM", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
