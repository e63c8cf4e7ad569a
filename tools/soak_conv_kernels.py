"""Race screen for the persistent bf16-operand convolution kernel: random frame shapes, split-bf16 forward and the plain-bf16 training
chain, persistent form vs the one-tile-per-workgroup kernels, bit for bit, repeated.   python tools/soak_conv_kernels.py [rounds=150]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import speech2lip_amd as s2l
from speech2lip_amd import weights as W, _abi
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
u = s2l.SimpleUnetLight().to(dev).eval()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
lib = _abi.load()
rng = np.random.default_rng(0)
bad = 0
for it in range(rounds):
    big = it % 3 == 0      # a third: few large frames; the rest: many small ones (both reach >= 4 tiles per workgroup, the plain form's threshold)
    F = int(rng.integers(1, 4)) if big else int(rng.integers(1, 64))
    H = int(rng.integers(200, 520)) if big else int(rng.integers(4, 140))
    Wd = int(rng.integers(200, 520)) if big else int(rng.integers(4, 140))
    x = torch.rand(F, H, Wd, 3, device=dev)
    d = torch.randn(F, H, Wd, 3, device=dev)
    res = []
    for kind in (1, 0, 0):
        _abi.check(lib.s2l_set_unet_split_kernel(kind), "kind")
        a = u.forward_nhwc(x, precision="split").clone()
        o, ctx = u.forward_saved_nhwc(x, precision="bf16")
        res.append((a, o.clone(), u.backward_input(ctx, d).clone()))
    ok = all(torch.equal(res[0][j], res[k][j]) for j in range(3) for k in (1, 2))
    if not ok:
        bad += 1
        print("MISMATCH", F, H, Wd, [[torch.equal(res[0][j], res[k][j]) for j in range(3)] for k in (1, 2)])
lib.s2l_set_unet_split_kernel(0)
print(f"{rounds} shapes, {bad} mismatches")
