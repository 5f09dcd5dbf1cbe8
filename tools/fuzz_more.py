"""more seeds of the differential option fuzz (tests/test_option_fuzz.py) than the suite runs, on the GPU or (FUZZ_ENGINE=sim)
on the SIMT emulator: python tools/fuzz_more.py FIRST LAST   -> one line per failing seed, a summary at the end"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import torch  # noqa: F401  (before the engine: one HIP runtime per process)
import pytest
import engines
import test_option_fuzz as tf

first, last = int(sys.argv[1]), int(sys.argv[2])
mk = engines.sim_engine if os.environ.get("FUZZ_ENGINE") == "sim" else engines.gpu_engine
exotic = 0
t0 = time.time()
ok = skipped = failed = 0
for seed in range(first, last):
    try:
        tf._check(mk, seed)
        ok += 1
    except pytest.skip.Exception:
        skipped += 1
    except BaseException as e:   # noqa: BLE001
        failed += 1
        print(f"seed {seed}: {type(e).__name__}: {str(e)[:300]}", flush=True)
print(f"seeds {first}..{last - 1}: {ok} equal to the oracle, {skipped} refused combinations, {failed} FAILED, {time.time() - t0:.0f}s")
