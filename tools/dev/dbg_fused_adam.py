import random, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import speech2lip_amd as s2l
from speech2lip_amd import weights as W
from tests.test_gpu_callers import _late_model, _cfg
from tests.test_gpu_training_chain import _g11_device
from tests.conftest import *      # noqa
dev = torch.device("cuda:0")
import tests.conftest as C
golden = lambda name: dict(np.load(os.path.join("tests", "golden", name), allow_pickle=True))
net = s2l.SyncNet_color().to(dev)
net.load_state_dict({k: torch.from_numpy(v) for k, v in W.make_syncnet_state_dict(0).items()})
_, data, _, _, _ = _g11_device(golden, dev)
frames = [dict(data, index=int(data["index"]) + k) for k in range(2)]
res = {}
for name, cls in (("torch", torch.optim.Adam), ("torch2", torch.optim.Adam), ("fused", s2l.FusedAdam)):
    m = _late_model(dev)
    p0 = {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}
    opt = cls([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    tr = s2l.Trainer(m, opt, cfg=_cfg(m), syncnet=net, use_syncloss=True, precision="bf16", hole_noise="device")
    torch.manual_seed(3); random.seed(3)
    tr.train_steps(frames, it=100001)
    g = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    res[name] = ({n: p.detach().clone() - p0[n] for n, p in m.named_parameters() if p.requires_grad}, g)
for other in ("torch2", "fused"):
    print("==", other)
    for n in res["torch"][0]:
        a, b = res["torch"][0][n], res[other][0][n]
        ga, gb = res["torch"][1].get(n), res[other][1].get(n)
        frac = float(((a - b).abs() > 1e-5).float().mean())
        gd = float((ga - gb).abs().max() / (ga.abs().max() + 1e-30)) if ga is not None else -1
        print(f"{n:34s} upd max {float(a.abs().max()):.2e} diff-frac {frac:.3f} grad rel diff {gd:.2e} grad absmax {float(ga.abs().max()) if ga is not None else 0:.2e}")
