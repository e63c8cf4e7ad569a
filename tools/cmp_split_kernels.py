"""The persistent bf16-operand convolution kernel (s2l_set_unet_split_kernel(0)) against the one-tile-per-workgroup forms (1): same
bits, in the split-bf16 inference mode and in the plain-bf16 training chain (forward with saved state, input gradient, train-mode
BatchNorm pair).    python tools/cmp_split_kernels.py [frames=3] [H=500] [W=500]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import speech2lip_amd as s2l
from speech2lip_amd import weights as W, _abi
dev = torch.device("cuda:0")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H = int(sys.argv[2]) if len(sys.argv) > 2 else 500
Wd = int(sys.argv[3]) if len(sys.argv) > 3 else 500
u = s2l.SimpleUnetLight().to(dev).eval()
u.load_state_dict({k[len("post_fusion_unet."):]: torch.from_numpy(v) for k, v in W.make_unet_state_dict(0).items()})
x = torch.rand(F, H, Wd, 3, device=dev)
d = torch.randn(F, H, Wd, 3, device=dev)
lib = _abi.load()


def run():
    res = {"split": u.forward_nhwc(x, precision="split").clone()}
    o, ctx = u.forward_saved_nhwc(x, precision="bf16")
    res["bf16 saved forward"], res["bf16 input gradient"] = o.clone(), u.backward_input(ctx, d).clone()
    u.train()
    try:
        o, ctx = u.forward_train_nhwc(x[:1], update_running=False, precision="bf16")
        dx, _ = u.backward_train(ctx, d[:1], want_param_grads=False)
    finally:
        u.eval()
    res["bf16 train-mode forward"], res["bf16 train-mode input gradient"] = o.clone(), dx.clone()
    return res


outs = []
for kind in (1, 0, 0):
    _abi.check(lib.s2l_set_unet_split_kernel(kind), "s2l_set_unet_split_kernel")
    outs.append(run())
torch.cuda.synchronize()
lib.s2l_set_unet_split_kernel(0)
ref = u.forward_nhwc(x)
for k in outs[0]:
    print(f"{k:32s} persistent == one-tile form: {torch.equal(outs[0][k], outs[1][k])}  run-to-run: {torch.equal(outs[1][k], outs[2][k])}"
          f"  max |diff|: {float((outs[0][k] - outs[1][k]).abs().max()):.3g}")
print("split rmse vs fp32:", float(((outs[1]["split"] - ref) ** 2).mean().sqrt()))
