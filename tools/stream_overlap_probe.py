"""Why do the configurations whose kernels run beside each other on the engine's second stream lose 5 - 13 % inside the full bench
process?  Runs bench.other_configs' lines (a) alone in a fresh process, (b) again in the same process,
(c) after eight other engines and a few torch streams have been created and closed, each with 4 and with 16 timed steps.
  python tools/stream_overlap_probe.py [only]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench
from fastp_amd import abi, engine

only = sys.argv[1] if len(sys.argv) > 1 else "soft-masked|-c --cut_right|--merge"
dev = torch.device("cuda", 0)


def lines(tag):
    for r in bench.other_configs(dev, only=only):
        print(json.dumps({"pass": tag, "config": r["config"][:60], "ms_per_step": r["ms_per_step"], "plan": r["plan"]}), flush=True)


lines("fresh process, first")
lines("same process, second")
streams = [torch.cuda.Stream(device=dev) for _ in range(6)]
for k in range(8):
    p = abi.default_params(True, 150)
    p.cut_right = 1
    e = engine.GpuEngine(p, device=0)
    e.synchronize()
    e.close()
lines("after 8 engines + 6 torch streams")
time.sleep(20)
lines("after 20 s of idle GPU")
