#!/usr/bin/env python3
"""Time the T3 sync loss (SyncNet_color forward x2 + face-encoder dgrad) on one MI355X.
    python tools/bench_syncnet.py [BATCH]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import speech2lip_amd as s2l
from speech2lip_amd import weights as W
from speech2lip_amd.syncnet import sync_window


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"      # "split": hi + lo bf16 operands (csrc/conv_gemm.h)
    dev = torch.device("cuda:0")
    net = s2l.SyncNet_color().to(dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_syncnet_state_dict(0).items()})
    mel, pos, neg = (torch.from_numpy(x).to(dev) for x in W.synthetic_sync_batch(B, seed=1))
    net.conv_precision = precision
    sl = s2l.SyncLoss(net)
    face = sync_window(pos)
    t_f = timed(lambda: net.embed_nhwc(mel, face))
    t_l = timed(lambda: sl.get_sync_contrastive_loss(mel, pos, neg))
    t_g = timed(lambda: sl.get_sync_contrastive_loss(mel, pos, neg, want_grad=True))
    gmac = 1.21 * B      # 1.21 GMAC per forward of both encoders
    print(f"batch {B} ({precision}): SyncNet forward {t_f * 1e3:.3f} ms ({2 * gmac / t_f / 1e3:.2f} TFLOP/s); "
          f"contrastive loss {t_l * 1e3:.3f} ms; loss + d/d window {t_g * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
