"""BASELINE config 3 end to end on one MI355X: 128x128 lip crop rendered for N frames
(audio encoder -> frame vectors -> fused MLP) + paste/head-pose-warp composite into 500x500 faces.
Inputs are synthetic and resident in HBM (SURVEY.md §8d recipe); frames are processed in batches so
that the pose grids (2 MB/frame) and observed frames (3 MB/frame) fit comfortably.
    python tools/bench_config3.py [frames=5000] [batch=500] [--unet] [--split]      (--split: the U-Net's split-bf16 operand mode)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import benchlib
a = [x for x in sys.argv[1:] if not x.startswith("--")]
print(json.dumps(benchlib.bench_config3(torch.device("cuda:0"), int(a[0]) if a else 5000, int(a[1]) if len(a) > 1 else 500,
                                        unet="--unet" in sys.argv or "--split" in sys.argv,
                                        unet_precision="split" if "--split" in sys.argv else "fp32")))
