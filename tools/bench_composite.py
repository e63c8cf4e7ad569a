"""Composite (paste + warp) throughput on MI355X: BASELINE config 3 geometry (128x128 lip in a
500x500 face).  HBM roofline: algorithmic bytes per frame = 8,000,000 + 12*h*w.
    python tools/bench_composite.py [frames]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import benchlib
print(json.dumps(benchlib.bench_composite(torch.device("cuda:0"), int(sys.argv[1]) if len(sys.argv) > 1 else 256)))
