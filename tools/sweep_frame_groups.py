"""Config-5 step with train-mode BatchNorm: frames per launch set of the frozen U-Net (the 40 window frames in groups of g).
    python tools/sweep_frame_groups.py [g ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools import benchlib
dev = torch.device("cuda:0")
for g in [int(a) for a in sys.argv[1:]] or [5, 8, 10, 14, 20, 25, 40]:
    r = benchlib.bench_train_sync(dev, 64, 8, "bf16", unet_train_mode=True, frames_per_group=g)
    print(f"frames per group {g:3d}: {r['ms_per_step']:.2f} ms per step, peak {r['peak_mem_gb']} GB", flush=True)
    torch.cuda.empty_cache()
