"""The opt-in split-half speed mode of the lip renderer next to the exact kernel (BASELINE config 2's workload): frames/s, ratio,
accuracy of both against the CPU oracle.   python tools/bench_render_split.py [frames=1000]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import benchlib
print(json.dumps(benchlib.bench_render_split(torch.device("cuda:0"), int(sys.argv[1]) if len(sys.argv) > 1 else 1000)))
