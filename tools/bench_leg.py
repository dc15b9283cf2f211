"""run one extra leg of bench.py by name (dev helper): python tools/bench_leg.py float_formats_leg"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for name in sys.argv[1:]:
    print(json.dumps({name: getattr(bench, name)(dev)}))
