"""one line of bench.other_configs (for a traced / ablated run of that configuration alone)
  python tools/one_config.py "<text the configuration's name contains>" """
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench
for r in bench.other_configs(torch.device("cuda", 0), only=sys.argv[1]):
    print(json.dumps(r), flush=True)
