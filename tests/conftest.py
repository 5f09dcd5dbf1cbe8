import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# PyTorch ships its own copy of the HIP runtime; libfastp_gpu.so links the system one.  Whichever is loaded first
# initialises the GPU, and a test process that loaded ours first and imports torch later finds "No HIP GPUs"
# from torch's copy.  Tests that need torch (device-resident batches, torch.distributed) therefore get it loaded
# before any engine is created; bench.py imports torch first for the same reason.
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
