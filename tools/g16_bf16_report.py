"""How close the bf16 step (half-width U-Net chain / fp32 tensors) is to the REFERENCE's own G16 numbers (tests/golden/g16_stage1_trainbn.npz)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import speech2lip_amd as s2l
from tests.test_gpu_training_chain import _g11_device, full_model
dev = torch.device("cuda:0")
golden = lambda name: dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', name)))
from speech2lip_amd import weights as W
net = s2l.SyncNet_color().to(dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_syncnet_state_dict(0).items()})
_, data, eps, sync, face = _g11_device(golden, dev)
g = golden("g16_stage1_trainbn.npz")
for prec, half in (("fp32", True), ("bf16", False), ("bf16", True)):
    m = full_model(dev, 16, 24).train()
    for p in m.post_fusion_unet.parameters():
        p.requires_grad = False
    m.post_fusion_unet.half_width_tensors = half
    step = s2l.StageOneStep(m, 16, 24, syncnet=net, precision=prec, face_loss=True)
    a, idx, tgt = data["audio"].to(dev), [data["index"]], data["rgb"].reshape(1, -1, 3).to(dev)
    loss, grads, aux = step.loss_and_grads(a, idx, tgt, [eps[0]], sync=sync, face=face)
    worst, wrel = 1.0, 0.0
    for key in g:
        if key.startswith("g_") and key != "g_pts5_cols":
            x, y = grads[key[2:]].double().flatten().cpu(), torch.from_numpy(np.asarray(g[key])).double().flatten()
            worst = min(worst, float((x @ y) / (x.norm() * y.norm() + 1e-30)))
            wrel = max(wrel, float((x - y).norm() / y.norm()))
    print(f"{prec:5s} half={half}: loss {float(loss):.6f} (ref {float(g['loss']):.6f}, rel {abs(float(loss)-float(g['loss']))/float(g['loss']):.2e})  "
          f"loss_sync rel {abs(float(aux['loss_sync'])-float(g['loss_sync']))/float(g['loss_sync']):.2e}  worst gradient cos {worst:.5f} rel {wrel:.3f}")
