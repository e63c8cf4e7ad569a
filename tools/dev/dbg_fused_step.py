import random, sys, torch
sys.path.insert(0, "/root/repo")
import speech2lip_amd as s2l
from tests.test_gpu_training_chain import _g11_device, full_model
from tests.test_gpu_callers import _late_model, _cfg
from speech2lip_amd import weights as W
import numpy as np
dev = torch.device("cuda:0")
def golden(name):
    return dict(np.load(f"/root/repo/tests/golden/{name}"))
net = s2l.SyncNet_color().to(dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_syncnet_state_dict(0).items()}, strict=True)
_, data, _, _, _ = _g11_device(golden, dev)
for it in (50000, 100001):
    res = []
    for fused in (True, False, False):
        m = _late_model(dev) if it > 100000 else full_model(dev, 16, 24).train()
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
        tr = s2l.Trainer(m, opt, cfg=_cfg(m), syncnet=net, use_syncloss=True, precision="bf16", hole_noise="device", fused_step=fused)
        torch.manual_seed(7); random.seed(7)
        out = tr.train_step(data, it=it) if fused else tr.train_steps([data], it=it)
        res.append(({k: v.detach().clone() for k, v in m.state_dict().items()}, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    for a, b, tag in ((res[0], res[1], "fused_step vs train_steps"), (res[1], res[2], "train_steps vs train_steps")):
        bad = [k for k in a[0] if not torch.equal(a[0][k], b[0][k])]
        badg = [k for k in a[1] if not torch.equal(a[1][k], b[1][k])]
        print(it, tag, "state differs:", bad[:6], len(bad), "grads differ:", badg[:6], len(badg))
        for k in badg[:3]:
            print("   ", k, float((a[1][k] - b[1][k]).abs().max()), float(a[1][k].abs().max()))
