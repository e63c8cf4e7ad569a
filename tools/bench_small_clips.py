"""One frame per call (the reference's operating mode, inference.py:129,140-159) and 16-frame clips: python tools/bench_small_clips.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import benchlib
print(json.dumps(benchlib.bench_small_clips(torch.device("cuda:0"))))
