"""Is s2l_out_grad_bf16 (and the rest of the bf16 step's gradients) the same bits call after call on the same inputs?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
from tools import benchlib
dev = torch.device("cuda:0")
for (h, w, B) in ((16, 24, 1), (16, 24, 6), (96, 96, 4)):
    m = benchlib.make_model(dev, h, w)
    step = s2l.LipTrainStep(m, h, w, "bf16")
    audio = torch.from_numpy(W.synthetic_audio(B, 1).astype(np.float32)).to(dev)
    tgt = torch.rand(B, h * w, 3, device=dev)
    ref = None
    for rep in range(6):
        loss, g, _ = step.loss_and_grads(audio, list(range(B)), tgt, [0.5] * B)
        torch.cuda.synchronize()
        cur = {k: v.clone() for k, v in g.items()}
        if ref is None:
            ref = cur
        else:
            bad = [(k, float((cur[k] - ref[k]).abs().max())) for k in ref if not torch.equal(cur[k], ref[k])]
            print(h, w, B, "rep", rep, "differs:", bad)
