"""The reference's graph configuration leg of bench.py on its own (for rocprofv3 --kernel-trace --stats)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
r = bench.graph_config_leg(torch.device('cuda', 0), reps=int(sys.argv[1]) if len(sys.argv) > 1 else 20)
print(json.dumps(r['stage_ms']), r['ms_per_batch'])
