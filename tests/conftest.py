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
    config.addinivalue_line("markers", "twin(name): the CPU-suite test that runs this GPU test's parametrisations on the emulator")


# The stream inflates a non-bgzip ".gz" input with several host threads (fastp_amd/csrc/fq_pgunzip.h), by default in chunks of
# 2 MiB of compressed data - more than any test file has.  The stream and binding tests run it with three threads and chunks
# of a few KiB (they grow until a deflate block fits), so that their small inputs cross many chunk boundaries;
# tests/test_gunzip.py covers the production geometry.  (Read when an input is opened; inherited by the patched reference.)
os.environ.setdefault("FASTP_GPU_STREAM_GUNZIP_THREADS", "3")
os.environ.setdefault("FASTP_GPU_STREAM_GUNZIP_CHUNK_KB", "6")


# No budget skipping: a `-m gpu` case that does not run is an untested case, whatever the summary line says.  The suite is
# kept inside the driver's 1200 s by its sizes (tests/test_gpu_parity.py), not by leaving cases out.


def pytest_collection_modifyitems(session, config, items):
    """The twin rule.  A `-m gpu` test marked `@pytest.mark.twin("test_function_name")` names the CPU-suite test that runs the
    same parametrisation on the emulator (tests/hostsim) at a small size.  Whenever both are collected, every parameter id of
    the GPU test must exist for its twin - a GPU case whose command line never ran anywhere before it reaches the box is how
    round 4's driver run went red.  (Runs before `-m` deselects: trylast=False, and pytest's own mark filter is a later hook.)"""
    by_func = {}
    for it in items:
        by_func.setdefault(it.originalname if hasattr(it, "originalname") else it.name, set()).add(
            it.callspec.id if hasattr(it, "callspec") else "")
    missing = []
    for it in items:
        m = it.get_closest_marker("twin")
        if m is None:
            continue
        twin = m.args[0]
        if twin not in by_func:       # the twin's file is not part of this collection (a single file was named)
            continue
        pid = it.callspec.id if hasattr(it, "callspec") else ""
        if pid not in by_func[twin]:
            missing.append(f"{it.nodeid}: no emulator twin {twin}[{pid}]")
    if missing:
        raise pytest.UsageError("GPU cases without an emulator twin:\n  " + "\n  ".join(missing))
